#!/usr/bin/env python3
"""Command line of the MI355X SinDDM hot-path build: same flags as the reference's main.py
(reference main.py:13-58) for the modes that run on the hot path: `train`, `sample`, `style_transfer`,
`harmonization` and `roi`.

    python main.py --scope balloons --mode train  --dataset_folder ./datasets/balloons/ --image_name balloons.png
    python main.py --scope balloons --mode sample --dataset_folder ./datasets/balloons/ --image_name balloons.png \
                   --load_milestone 12 --sample_batch_size 16 [--scale_mul 2 4]

Multi-GPU sampling: launch one process per GPU with torchrun; the sample batch is sharded over the
ranks as independent chains and all-gathered (RCCL over xGMI):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 main.py --mode sample ...

`style_transfer` / `harmonization` (reference main.py:296-322) drive `MultiscaleTrainer.image2image`; `roi`
(main.py:257-294) drives `roi_guided_sampling` -- the reference picks the boxes with a cv2 GUI, here they come from
`--roi_target y x h w` and `--roi_bbs y x h w [y x h w ...]` (finest-scale pixel coordinates).
The CLIP-guided modes (clip_content, clip_style_*, clip_roi; main.py:153-255) are not wired to the command line: CLIP
itself is outside this build.  Their drivers exist (`MultiscaleTrainer.clip_sampling` / `clip_roi_sampling`, the guidance
branch of `p_mean_variance`) and take any scorer with the reference's ClipExtractor interface.
"""
import argparse
import os

import torch

from sinddm_amd.functions import create_img_scales
from sinddm_amd.models import MultiScaleGaussianDiffusion, SinDDMNet
from sinddm_amd.trainer import MultiscaleTrainer


# (flag, default, type, nargs) -- names, defaults and types follow reference main.py:13-58 so existing
# command lines keep working; flags that only feed the un-built guided modes are accepted and ignored.
_FLAGS = [
    ("scope", "forest", str, None), ("mode", None, str, None),
    ("input_image", "seascape_composite_dragon.png", str, None), ("start_t_harm", 5, int, None),
    ("start_t_style", 15, int, None), ("harm_mask", "seascape_mask_dragon.png", str, None),
    ("clip_text", "Fire in the Forest", str, None), ("fill_factor", None, float, None),
    ("strength", None, float, None), ("roi_n_tar", 1, int, None),
    ("dataset_folder", "./datasets/forest/", str, None), ("image_name", "forest.jpeg", str, None),
    ("results_folder", "./results/", str, None),
    ("dim", 160, int, None), ("scale_factor", 1.411, float, None), ("timesteps", 100, int, None),
    ("train_batch_size", 32, int, None), ("grad_accumulate", 1, int, None), ("train_num_steps", 120001, int, None),
    ("save_and_sample_every", 10000, int, None), ("avg_window", 100, int, None), ("train_lr", 1e-3, float, None),
    ("sched_k_milestones", [20, 40, 70, 80, 90, 110], int, "+"), ("load_milestone", 0, int, None),
    ("sample_batch_size", 16, int, None), ("scale_mul", [1, 1], float, "+"), ("sample_t_list", None, int, "+"),
    ("device_num", 0, int, None), ("omega", 0, float, None), ("loss_factor", 1, float, None),
    # non-interactive stand-ins for the cv2.selectROI dialogs of the reference's `roi` mode
    ("roi_target", None, int, "+"), ("roi_bbs", None, int, "+"),
]


def build_parser():
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    for name, dflt, typ, nargs in _FLAGS:
        kw = dict(default=dflt, type=typ)
        if nargs:
            kw["nargs"] = nargs
        p.add_argument("--" + name, **kw)
    p.add_argument("--sample_limited_t", action="store_true")
    return p


def main():
    args = build_parser().parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    td = None
    if world > 1:
        import torch.distributed as td
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        td.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        args.device_num = local_rank
        torch.manual_seed(1234 + td.get_rank())      # every rank draws its own chains
    print('num devices: ' + str(torch.cuda.device_count()))
    device = f"cuda:{args.device_num}"
    if world == 1 and torch.cuda.is_available():
        torch.cuda.set_device(args.device_num)      # the HIP library launches on the CURRENT device's stream
    scale_mul = (args.scale_mul[0], args.scale_mul[1])
    sched_milestones = [val * 1000 for val in args.sched_k_milestones]
    results_folder = args.results_folder + '/' + args.scope

    # reference main.py:71-75.  Under torch.distributed only rank 0 writes the scale_i/ PNG pyramid; the others wait
    # and then only derive the sizes / losses from the files rank 0 wrote (create=False), so no rank ever opens a
    # half-written PNG.
    rank = 0
    if world > 1:
        rank = td.get_rank()
    if rank == 0:
        sizes, rescale_losses, scale_factor, n_scales = create_img_scales(
            args.dataset_folder, args.image_name, scale_factor=args.scale_factor, create=True, auto_scale=50000)
    if world > 1:
        td.barrier()
        if rank != 0:
            sizes, rescale_losses, scale_factor, n_scales = create_img_scales(
                args.dataset_folder, args.image_name, scale_factor=args.scale_factor, create=False, auto_scale=50000)
        td.barrier()

    model = SinDDMNet(dim=args.dim, multiscale=True, device=device)
    model.to(device)
    ms_diffusion = MultiScaleGaussianDiffusion(
        denoise_fn=model, save_interm=False, results_folder=results_folder, n_scales=n_scales,
        scale_factor=scale_factor, image_sizes=sizes, scale_mul=scale_mul, channels=3, timesteps=args.timesteps,
        train_full_t=True, scale_losses=rescale_losses, loss_factor=args.loss_factor, loss_type='l1', betas=None,
        device=device, reblurring=True, sample_limited_t=args.sample_limited_t, omega=args.omega).to(device)

    sample_t_list = ms_diffusion.num_timesteps_ideal[1:] if args.sample_t_list is None else args.sample_t_list

    trainer = MultiscaleTrainer(
        ms_diffusion, folder=args.dataset_folder, n_scales=n_scales, scale_factor=scale_factor, image_sizes=sizes,
        train_batch_size=args.train_batch_size, train_lr=args.train_lr, train_num_steps=args.train_num_steps,
        gradient_accumulate_every=args.grad_accumulate, ema_decay=0.995, fp16=False,
        save_and_sample_every=args.save_and_sample_every, avg_window=args.avg_window,
        sched_milestones=sched_milestones, results_folder=results_folder, device=device)

    if args.load_milestone > 0:
        trainer.load(milestone=args.load_milestone)
    if args.mode == 'train':
        trainer.train()
        trainer.sample_scales(scale_mul=(1, 1), custom_sample=True, image_name=args.image_name,
                              batch_size=args.sample_batch_size, custom_t_list=sample_t_list)
    elif args.mode == 'sample':
        trainer.sample_scales(scale_mul=scale_mul, custom_sample=True, image_name=args.image_name,
                              batch_size=args.sample_batch_size, custom_t_list=sample_t_list, save_unbatched=True)
    elif args.mode in ('style_transfer', 'harmonization'):                 # reference main.py:296-322
        i2i_folder = os.path.join(args.dataset_folder, 'i2i')
        start_s = n_scales - 1                                             # start the diffusion at the last scale
        start_t = args.start_t_style if args.mode == 'style_transfer' else args.start_t_harm
        use_hist = args.mode == 'style_transfer'
        custom_t = [0] * (n_scales - 1) + [start_t]
        trainer.ema_model.reblurring = True
        trainer.image2image(input_folder=i2i_folder, input_file=args.input_image, mask=args.harm_mask,
                            hist_ref_path=f'{args.dataset_folder}scale_{start_s}/', batch_size=args.sample_batch_size,
                            image_name=args.image_name, start_s=start_s, custom_t=custom_t, scale_mul=(1, 1),
                            device=device, use_hist=use_hist, save_unbatched=True, auto_scale=50000, mode=args.mode)
    elif args.mode == 'roi':                                               # reference main.py:257-294
        if not args.roi_target or len(args.roi_target) != 4 or not args.roi_bbs or len(args.roi_bbs) % 4:
            raise SystemExit("--mode roi needs --roi_target y x h w and --roi_bbs y x h w [y x h w ...]")
        bbs = [list(args.roi_bbs[i:i + 4]) for i in range(0, len(args.roi_bbs), 4)]
        trainer.roi_guided_sampling(custom_t_list=sample_t_list, target_roi=list(args.roi_target), roi_bb_list=bbs,
                                    save_unbatched=True, batch_size=args.sample_batch_size, scale_mul=scale_mul)
    else:
        raise NotImplementedError(
            f"mode {args.mode!r}: train, sample, style_transfer, harmonization and roi are built for MI355X; the CLIP-guided "
            "modes of the reference need CLIP autograd and are out of scope (SURVEY.md section 8)")
    if world > 1:
        td.destroy_process_group()


if __name__ == '__main__':
    main()
    quit()
