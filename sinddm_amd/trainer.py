"""MultiscaleTrainer: training loop, multi-scale sampling driver, checkpoint I/O.

Mirror of reference SinDDM/trainer.py:35-285 (the hot-path part: __init__, train, sample_scales,
save, load, step_ema, reset_parameters, Dataset) and of its application drivers (image2image, roi_guided_sampling,
clip_sampling, clip_roi_sampling -- trainer.py:287-488; the two CLIP ones against any external scorer with the
reference's ClipExtractor interface: CLIP itself is not part of this build).
"""
from __future__ import annotations

import copy
import datetime
from pathlib import Path
from typing import List, Optional

import numpy as np
import torch
from PIL import Image
from torch.optim.lr_scheduler import MultiStepLR

from . import dist as sdist
from .functions import num_to_groups
from .models import EMA, SinDDMNet
from .optim import FusedAdam, ema_update_


def image_to_tensor(img: Image.Image) -> torch.Tensor:
    """PIL RGB uint8 -> float32 CHW in [-1, 1]  (ToTensor + Lambda(t*2-1), trainer.py:46-50)."""
    a = np.asarray(img.convert('RGB'), dtype=np.uint8)
    return torch.from_numpy(a.transpose(2, 0, 1).copy()).to(torch.float32).div(255).mul(2).sub(1)


def image_to_tensor_01(img: Image.Image) -> torch.Tensor:
    """PIL RGB uint8 -> float32 CHW in [0, 1]  (torchvision ToTensor)."""
    a = np.asarray(img.convert('RGB'), dtype=np.uint8)
    return torch.from_numpy(a.transpose(2, 0, 1).copy()).to(torch.float32).div(255)


def save_image(tensor: torch.Tensor, path: str, nrow: int = 8, padding: int = 2) -> None:
    """Minimal stand-in for torchvision.utils.save_image (grid of [0,1] images -> PNG). Host side,
    off the timed path."""
    t = tensor.detach().float().cpu()
    if t.dim() == 3:
        t = t[None]
    t = t.clamp(0, 1)
    if t.shape[1] == 1:                      # (torchvision's make_grid shows a single-channel image as three equal channels)
        t = t.repeat(1, 3, 1, 1)
    B, C, H, W = t.shape
    if B == 1:
        # torchvision.utils.make_grid returns a single image as it is: no grid, no 2-pixel frame
        arr = t[0].mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
        Image.fromarray(arr).save(path)
        return
    ncol = min(nrow, B)
    nrows = (B + ncol - 1) // ncol
    grid = torch.zeros(C, nrows * (H + padding) + padding, ncol * (W + padding) + padding)
    for i in range(B):
        r, c = divmod(i, ncol)
        y, x = r * (H + padding) + padding, c * (W + padding) + padding
        grid[:, y:y + H, x:x + W] = t[i]
    arr = (grid.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy())
    Image.fromarray(arr).save(path)


class Dataset(torch.utils.data.Dataset):
    """One training image (and its blurry re-upsampled version) per scale folder (trainer.py:35-63)."""

    def __init__(self, folder, image_size, blurry_img=False, exts=('jpg', 'jpeg', 'png')):
        super().__init__()
        self.folder = folder
        self.image_size = image_size
        self.blurry_img = blurry_img
        self.paths = [p for ext in exts for p in Path(f'{folder}').glob(f'**/*.{ext}')]
        if blurry_img:
            self.folder_recon = folder + '_recon/'
            self.paths_recon = [p for ext in exts for p in Path(f'{self.folder_recon}').glob(f'**/*.{ext}')]

    def __len__(self):
        return len(self.paths) * 128

    def __getitem__(self, index):
        img = image_to_tensor(Image.open(self.paths[0]))
        if self.blurry_img:
            return img, image_to_tensor(Image.open(self.paths_recon[0]))
        return img


class MultiscaleTrainer(object):

    def __init__(self, ms_diffusion_model, folder, *, ema_decay=0.995, n_scales=None, scale_factor=1,
                 image_sizes=None, train_batch_size=32, train_lr=2e-5, train_num_steps=100000,
                 gradient_accumulate_every=2, fp16=False, step_start_ema=2000, update_ema_every=10,
                 save_and_sample_every=25000, avg_window=100, sched_milestones=None, results_folder='./results',
                 device=None):
        super().__init__()
        self.device = device
        self.sched_milestones = [10000, 30000, 60000, 80000, 90000] if sched_milestones is None else sched_milestones
        if image_sizes is None:
            image_sizes = []
        self.model = ms_diffusion_model
        self.ema = EMA(ema_decay)
        self.ema_decay = ema_decay
        self.ema_model = copy.deepcopy(self.model)
        self.update_ema_every = update_ema_every
        self.step_start_ema = step_start_ema
        self.save_and_sample_every = save_and_sample_every
        self.avg_window = avg_window
        self.batch_size = train_batch_size
        self.n_scales = n_scales
        self.scale_factor = scale_factor
        self.gradient_accumulate_every = gradient_accumulate_every
        self.train_num_steps = train_num_steps

        # data-parallel training when launched under torch.distributed (one process per GPU): this rank trains on
        # its shard of the batch; gradients are summed with one all-reduce per optimizer step (sinddm_amd/dist.py)
        self.data_parallel = sdist.is_dist() and sdist.world_size() > 1
        self.local_batch_size = sdist.local_batch(train_batch_size) if self.data_parallel else train_batch_size
        if self.local_batch_size < 1:
            raise ValueError(f'train_batch_size={train_batch_size} leaves rank {sdist.rank()} without samples')
        self.loss_weight = self.local_batch_size / float(train_batch_size)
        self._scale_gen = None

        self.input_paths = []
        self.ds_list = []
        self.data_list = []
        self.results_folder = Path(results_folder)
        self.results_folder.mkdir(parents=True, exist_ok=True)

        # one batch of B identical copies per scale, parked on the device (trainer.py:113-132)
        for i in range(n_scales):
            self.input_paths.append(folder + 'scale_' + str(i))
            ds = Dataset(self.input_paths[i], image_sizes[i], blurry_img=i > 0)
            self.ds_list.append(ds)
            item = ds[0]
            rep = lambda t: t[None].repeat(self.local_batch_size, 1, 1, 1).contiguous().to(self.device)
            if i > 0:
                self.data_list.append((rep(item[0]), rep(item[1])))
            else:
                self.data_list.append((rep(item), rep(item)))

        if fp16:
            raise NotImplementedError('apex mixed precision is never enabled by main.py (fp16=False, main.py:122)')
        self.fp16 = fp16

        if isinstance(self.model.denoise_fn, SinDDMNet):
            self.opt = FusedAdam(self.model.denoise_fn, lr=train_lr)
        else:
            self.opt = torch.optim.Adam(ms_diffusion_model.parameters(), lr=train_lr)
        self.scheduler = MultiStepLR(self.opt, milestones=self.sched_milestones, gamma=0.5)

        if self.data_parallel:
            # ranks seed their generators differently (own noise streams), so the initial weights must be agreed on
            with torch.no_grad():
                if isinstance(self.model.denoise_fn, SinDDMNet):
                    sdist.broadcast_(self.model.denoise_fn.flat_params)      # one 4.4 MB message
                    self.model.denoise_fn.mark_dirty()
                else:
                    for prm in self.model.denoise_fn.parameters():
                        sdist.broadcast_(prm.data)

        self.step = 0
        self.running_loss = []
        self.running_scale = []
        self.avg_t = []
        # optional injection hooks for parity tests: scale_fn(step) -> int
        self.scale_fn = None
        self.reset_parameters()

    def reset_parameters(self):
        """EMA weights := model weights (reference trainer.py:152-153).  The network's 52 tensors are one flat buffer: ONE
        launch (sinddm_adam_ema_step mode 2); the diffusion object's 13 schedule buffers are constants equal in both
        modules by construction, load_state_dict stays the path for foreign denoisers and CPU modules."""
        src, dst = self.model.denoise_fn, self.ema_model.denoise_fn
        if (isinstance(src, SinDDMNet) and isinstance(dst, SinDDMNet) and src.flat_params.is_cuda
                and dst.flat_params.is_cuda and src.flat_params.numel() == dst.flat_params.numel()):
            from .optim import ema_copy_
            ema_copy_(dst, src)
            return
        self.ema_model.load_state_dict(self.model.state_dict())

    def step_ema(self):
        """Copy until step_start_ema, then ema = decay*ema + (1-decay)*p  (trainer.py:155-159)."""
        if self.step < self.step_start_ema:
            self.reset_parameters()
            return
        src, dst = self.model.denoise_fn, self.ema_model.denoise_fn
        if isinstance(src, SinDDMNet) and isinstance(dst, SinDDMNet):
            ema_update_(dst, src, self.ema_decay)
        else:
            self.ema.update_model_average(self.ema_model, self.model)

    # ---- checkpoints: same file layout as the reference (trainer.py:161-187) so its model-N.pt files interoperate ----
    _CKPT_KEYS = ('step', 'model', 'ema', 'sched', 'running_loss', 'running_scale')

    def _ckpt_path(self, milestone) -> str:
        return str(self.results_folder / f'model-{milestone}.pt')

    def _checkpoint_state(self) -> dict:
        state = dict(zip(self._CKPT_KEYS, (self.step, self.model.state_dict(), self.ema_model.state_dict(),
                                           self.scheduler.state_dict(), self.running_loss, self.running_scale)))
        if isinstance(self.opt, FusedAdam):
            state['opt'] = self.opt.state_dict()       # extra key: the reference drops the Adam moments on save
        return state

    def _plot_running_loss(self) -> None:
        """running_loss.png next to the checkpoints (the reference plots the same curve); best effort, headless."""
        try:
            import matplotlib
            matplotlib.use('Agg')
            from matplotlib import pyplot as plt
        except Exception:
            return
        fig, ax = plt.subplots(figsize=(16, 8))
        ax.plot(self.running_loss)
        ax.set_ylim(0, 0.2)
        ax.grid(True)
        fig.savefig(str(self.results_folder / 'running_loss'))
        plt.close(fig)

    def save(self, milestone):
        torch.save(self._checkpoint_state(), self._ckpt_path(milestone))
        self._plot_running_loss()

    def load(self, milestone):
        state = torch.load(self._ckpt_path(milestone), map_location=self.device, weights_only=False)
        self.model.load_state_dict(state['model'])
        self.ema_model.load_state_dict(state['ema'])
        self.scheduler.load_state_dict(state['sched'])
        self.step, self.running_loss = state['step'], state['running_loss']
        if isinstance(self.opt, FusedAdam) and 'opt' in state:
            self.opt.load_state_dict(state['opt'])

    def _pick_scale(self, weights: torch.Tensor) -> int:
        if self.scale_fn is not None:
            return int(self.scale_fn(self.step))
        # trainer.py:197 (host draw: no sync); under data parallelism every rank draws from an identically
        # seeded private generator so that all ranks train the same scale in the same step
        return int(torch.multinomial(input=weights, num_samples=1, generator=self._scale_gen))

    def train(self):
        loss_acc = None
        s_weights = torch.tensor(self.model.num_timesteps_trained, dtype=torch.float)
        dp = self.data_parallel
        if dp and self._scale_gen is None:
            self._scale_gen = torch.Generator()
            # rank 0's seed, agreed on by broadcast; torch.initial_seed() only READS the global generator, so the
            # documented per-rank noise streams (manual_seed(1234 + rank) in main.py) stay what the user set
            self._scale_gen.manual_seed(sdist.broadcast_int(int(torch.initial_seed() % (2 ** 31))))
        net = self.model.denoise_fn
        while self.step < self.train_num_steps:
            s = self._pick_scale(s_weights)
            for _ in range(self.gradient_accumulate_every):
                data = self.data_list[s]
                loss = self.model(data, s)
                if dp:
                    loss = loss * self.loss_weight       # shard mean -> this shard's share of the global-batch mean
                d = loss.detach()
                loss_acc = d if loss_acc is None else loss_acc + d
                (loss / self.gradient_accumulate_every).backward()
            if dp:
                # one collective per optimizer step on the flat gradient buffer (4.4 MB at dim=160)
                if isinstance(net, SinDDMNet):
                    sdist.allreduce_sum_(net.flat_grads)
                else:
                    for prm in self.model.parameters():
                        if prm.grad is not None:
                            sdist.allreduce_sum_(prm.grad)
            if self.step % self.avg_window == 0:
                if dp:
                    loss_acc = sdist.allreduce_sum_(loss_acc.reshape(1).clone())
                avg = float(loss_acc) / self.avg_window          # the only device->host sync of the loop
                if sdist.rank() == 0:
                    print(f'step:{self.step} loss:{avg}')
                self.running_loss.append(avg)
                loss_acc = None
            self.opt.step()
            self.opt.zero_grad()
            if self.step % self.update_ema_every == 0:
                self.step_ema()
            self.scheduler.step()
            self.step += 1
            if self.step % self.save_and_sample_every == 0 and sdist.rank() == 0:
                milestone = self.step // self.save_and_sample_every
                batches = num_to_groups(16, self.batch_size)
                all_images = torch.cat([self.ema_model.sample(batch_size=n) for n in batches], dim=0)
                all_images = (all_images + 1) * 0.5
                save_image(all_images, str(self.results_folder / f'sample-{milestone}.png'), nrow=4)
                self.save(milestone)
        if sdist.rank() == 0:
            print('training completed')

    @torch.no_grad()
    def sample_scales(self, scale_mul=None, batch_size=16, custom_sample=False, custom_image_size_idxs=None,
                      custom_scales=None, image_name='', start_noise=True, custom_t_list=None, desc=None,
                      save_unbatched=True, save_images=True) -> List[torch.Tensor]:
        """Drive the sampler over all scales (behaviour of reference trainer.py:226-285).  Under torch.distributed the
        batch is sharded over ranks as independent chains and gathered with one all-gather per scale (RCCL over xGMI
        on MI355X); returns the list of per-scale (global) sample batches."""
        em = self.ema_model
        # --- what to run: one (scale, size index, start timestep) triple per stage ---
        scales = list(range(self.n_scales)) if custom_scales is None else list(custom_scales)
        size_idx = list(range(self.n_scales)) if custom_image_size_idxs is None else list(custom_image_size_idxs)
        t_starts = list(em.num_timesteps_ideal[1:]) if custom_t_list is None else list(custom_t_list)
        stretch = (1, 1) if scale_mul is None else scale_mul
        first_size = None
        if scale_mul is not None:
            h0, w0 = self.model.image_sizes[size_idx[0]]
            first_size = (int(h0 * scale_mul[0]), int(w0 * scale_mul[1]))
        # --- where the PNGs go (names as the reference writes them) ---
        tag = desc if desc is not None else f'sample_{str(datetime.datetime.now()).replace(":", "_")}'
        tag += '_rblr' if em.reblurring else ''
        tag += '_t_lmtd' if em.sample_limited_t else ''
        t_tag = '_'.join(str(e) for e in [em.num_timesteps_trained[0]] + t_starts)
        out_dir = Path(str(self.results_folder / 'final_samples'))
        writer = sdist.rank() == 0 and save_images
        if writer:
            out_dir.mkdir(parents=True, exist_ok=True)
        # --- this rank's chains ---
        if batch_size < sdist.world_size():
            raise ValueError(f'sample_scales: batch_size={batch_size} < world_size={sdist.world_size()} would leave ranks '
                             'without chains (every rank must join the all-gather)')
        mine = sdist.local_batch(batch_size)
        per_scale, cur, shown = [], None, None
        for stage, s in enumerate(scales):
            if stage > 0:
                cur = em.sample_via_scale(mine, cur, s=s, scale_mul=stretch, custom_sample=custom_sample,
                                          custom_img_size_idx=size_idx[stage], custom_t=t_starts[int(s) - 1])
            elif start_noise:
                cur = em.sample(batch_size=mine, scale_0_size=first_size, s=s)
            else:                                   # start from the training image of that scale instead of noise
                seed_img = Image.open(self.input_paths[s] + '/' + image_name).convert('RGB')
                cur = image_to_tensor(seed_img).repeat(mine, 1, 1, 1).to(self.device)
            whole = sdist.gather_batch(cur, batch_size)          # identity on a single process
            per_scale.append(whole)
            if writer:
                shown = (whole + 1) * 0.5
                save_image(shown, str(out_dir / t_tag) + f'_out_s{stage}_{tag}_sm_{stretch[0]}_{stretch[1]}.png', nrow=4)
        if writer and save_unbatched and shown is not None:
            single_dir = Path(str(self.results_folder / f'final_samples_unbatched_{tag}'))
            single_dir.mkdir(parents=True, exist_ok=True)
            for b, one in enumerate(shown):
                save_image(one, str(single_dir / t_tag) + f'_out_b{b}.png')
        return per_scale

    # ---- application drivers of the reference that run entirely on the hot path ----
    def _i2i_source(self, input_folder, input_file, mask, hist_ref_path, image_name, use_hist, auto_scale, mode, device):
        """Host-side preparation of harmonization / style transfer: the (optionally shrunk, optionally histogram
        matched) source image as a [-1,1] tensor, and the blend mask (1 = keep the sample everywhere)."""
        import os
        from .functions import dilate_mask, match_histograms
        src = Image.open(os.path.join(input_folder, input_file)).convert("RGB")
        size = src.size
        if auto_scale is not None:
            shrink = np.sqrt((size[0] * size[1]) / auto_scale)
            if shrink > 1:
                size = (int(size[0] / shrink), int(size[1] / shrink))
                src = src.resize(size, Image.LANCZOS)
        keep = 1
        if mode == 'harmonization':
            m = Image.open(os.path.join(input_folder, mask)).convert("RGB").resize(size, Image.LANCZOS)
            keep = torch.from_numpy(dilate_mask(image_to_tensor_01(m), mode=mode)).to(device=device, dtype=torch.float32)
        if use_hist:
            ref = Image.open(hist_ref_path + image_name.rsplit(".", 1)[0] + '.png').convert("RGB")
            src = Image.fromarray(match_histograms(image=np.array(src), reference=np.array(ref), channel_axis=2))
        return image_to_tensor(src), keep

    @torch.no_grad()
    def image2image(self, input_folder='', input_file='', mask='', hist_ref_path='', image_name='', start_s=1,
                    custom_t=None, batch_size=16, scale_mul=(1, 1), device=None, use_hist=False, save_unbatched=True,
                    auto_scale=None, mode=None, save_images=True):
        """Harmonization / style transfer (behaviour of reference trainer.py:287-362): the input image is re-noised at
        scale `start_s` to `custom_t[start_s]` and denoised through the remaining scales by the trained model; in
        harmonization mode the result is pasted into the input through the dilated mask.  Returns the per-scale sample
        batches (the reference only writes PNGs)."""
        import os
        device = self.device if device is None else device
        em = self.ema_model
        t_starts = em.num_timesteps_ideal if custom_t is None else custom_t
        src, keep = self._i2i_source(input_folder, input_file, mask, hist_ref_path, image_name, use_hist, auto_scale,
                                     mode, device)
        src_hw = torch.tensor(src.shape[1:])
        batch = src.repeat(batch_size, 1, 1, 1).to(device)
        if start_s > 0:
            # no blur mixing at the scale the chain starts from: its gamma row is zeroed in place, as the reference does
            em.gammas[start_s - 1].clamp_(0, 0)
        stamp = str(datetime.datetime.now()).replace(":", "_")
        t_tag = '_'.join(str(e) for e in t_starts)
        out_dir = Path(str(self.results_folder / 'i2i_final_samples'))
        if save_images:
            out_dir.mkdir(parents=True, exist_ok=True)
        last = self.n_scales - 1
        outs, shown = [], None
        for s in range(start_s, self.n_scales):
            # target size of this scale: the (possibly shrunk) input divided by scale_factor^(scales still to go)
            hw = src_hw / (self.scale_factor ** (last - s))
            target = (int(hw[0].item()), int(hw[1].item()))
            outs.append(em.sample_via_scale(batch_size, batch if not outs else outs[-1], s=s, custom_t=t_starts[s],
                                            scale_mul=scale_mul, custom_image_size=target))
            shown = (outs[-1] + 1) * 0.5
            if s == last:
                shown = keep * shown + (1 - keep) * ((batch + 1) * 0.5).clamp_(0.0, 1.0)
            if save_images:
                stem = input_file.rsplit(".", 1)[0]
                save_image(shown, str(out_dir / f'{stem}_i2i_s_{s}_t_{t_tag}_hist_{"on" if use_hist else "off"}_{stamp}.png'),
                           nrow=4)
        if save_images and save_unbatched:
            single_dir = Path(str(self.results_folder / f'unbatched_i2i_s{start_s}_t_{t_tag}_{stamp}'))
            single_dir.mkdir(parents=True, exist_ok=True)
            for b in range(batch_size):
                save_image(shown[b], os.path.join(single_dir, input_file + f'_out_b{b}_i2i.png'))
        self.last_i2i_image = shown
        return outs

    @torch.no_grad()
    def roi_guided_sampling(self, custom_t_list=None, target_roi=None, roi_bb_list=None, save_unbatched=False,
                            batch_size=4, scale_mul=(1, 1), save_images=True):
        """ROI guided generation (trainer.py:436-454): at every scale but the finest the predicted clean image is
        pulled (eta = 0.8) towards a patch of the training image inside the given boxes; the blend runs inside the
        fused reverse-step kernel (`sinddm_reverse_step_edit`)."""
        from .functions import extract_patch
        em = self.ema_model
        em.roi_guided_sampling = True
        em.roi_bbs = roi_bb_list
        em.roi_target_patch = []       # (the reference appends on every call; a fresh list per call is what it means)
        for scale in range(self.n_scales):
            bb = [int(bb_i / np.power(self.scale_factor, self.n_scales - scale - 1)) for bb_i in target_roi]
            em.roi_target_patch.append(extract_patch(self.data_list[scale][0][0][None, :, :, :], bb))
        try:
            return self.sample_scales(scale_mul=scale_mul, custom_sample=False, image_name='', batch_size=batch_size,
                                      custom_t_list=custom_t_list,
                                      desc=f'roi_{str(datetime.datetime.now()).replace(":", "_")}',
                                      save_unbatched=save_unbatched, start_noise=True, save_images=save_images)
        finally:
            em.roi_guided_sampling = False

    # ---- CLIP-driven modes of the reference.  CLIP itself (clip/, text2live_util/) is not part of this build; the
    # driver takes any `clip_model` with the interface the reference uses: get_text_embedding(text, template=...),
    # zero_grad(), calculate_clip_loss(image in [0,1], embedding) (differentiable), cfg["n_aug"] ----
    def clip_sampling(self, clip_model, text_input, strength, sample_batch_size, custom_t_list=None,
                      guidance_sub_iters=None, quantile=0.8, stop_guidance=None, save_unbatched=False, scale_mul=(1, 1),
                      llambda=0, start_noise=True, image_name='', templates=('hr', 'lr')):   # trainer.py:363-410
        em = self.ema_model
        if guidance_sub_iters is None:
            guidance_sub_iters = [*reversed(range(self.n_scales))]
        em.clip_strength = strength
        em.clip_text = text_input
        em.text_embedds_hr = clip_model.get_text_embedding(text_input, template=templates[0])
        em.text_embedds_lr = clip_model.get_text_embedding(text_input, template=templates[1])
        em.clip_guided_sampling = True
        em.guidance_sub_iters = guidance_sub_iters
        em.quantile = quantile
        em.stop_guidance = stop_guidance
        em.clip_model = clip_model
        em.clip_score = []
        em.llambda = llambda
        em.clip_mask = None
        em.x_recon_prev = None
        n_aug = getattr(clip_model, "cfg", {}).get("n_aug", 0)
        desc = (f"clip_{text_input.replace(' ', '_')}_n_aug{n_aug}_str_{strength}_gsi_" + '_'.join(str(e) for e in guidance_sub_iters)
                + f'_ff{1 - quantile}' + f'_{str(datetime.datetime.now()).replace(":", "_")}')
        try:
            if not start_noise:                                             # clip_style_trans: start from the last two scales
                return self.sample_scales(scale_mul=scale_mul, custom_sample=True,
                                          custom_scales=[self.n_scales - 2, self.n_scales - 1],
                                          custom_image_size_idxs=[self.n_scales - 2, self.n_scales - 1],
                                          image_name=image_name, batch_size=sample_batch_size, custom_t_list=custom_t_list,
                                          desc=desc, save_unbatched=save_unbatched, start_noise=start_noise)
            return self.sample_scales(scale_mul=scale_mul, custom_sample=False, image_name='', batch_size=sample_batch_size,
                                      custom_t_list=custom_t_list, desc=desc, save_unbatched=save_unbatched,
                                      start_noise=start_noise)
        finally:
            em.clip_guided_sampling = False

    def clip_roi_sampling(self, clip_model, text_input, strength, sample_batch_size, num_clip_iters=100,
                          num_denoising_steps=2, clip_roi_bb=None, save_unbatched=False, template='lr',
                          save_images=True):                                      # trainer.py:412-468
        """Text-guided edit of a region of the training image: `num_clip_iters` steps of normalised gradient ascent of
        the external score on the ROI alone (step = strength * |roi| / |grad| per sample, clamped to [-1, 1] -- plain
        autograd on the ROI tensor, no network involved), the patch pasted back into the image, then the finest
        scale's sampler for `num_denoising_steps` reverse steps (q_sample to t = num_denoising_steps + the HIP chain) to
        blend it in.  Returns the final batch in [-1, 1]."""
        em = self.ema_model
        y0, x0, h, w = clip_roi_bb
        top = self.n_scales - 1
        emb = clip_model.get_text_embedding(text_input, template=template)
        image = self.data_list[top][0][0][None].repeat(sample_batch_size, 1, 1, 1).clone()
        roi = image[:, :, y0:y0 + h, x0:x0 + w].clone()
        trace_dir = Path(str(em.results_folder / 'interm_samples_clip_roi'))
        if em.save_interm:
            trace_dir.mkdir(parents=True, exist_ok=True)
        norm = lambda v: torch.linalg.vector_norm(v, dim=(1, 2, 3), keepdim=True)
        for it in range(num_clip_iters):
            roi = roi.detach().requires_grad_(True)
            clip_model.zero_grad()
            with torch.enable_grad():
                score = -clip_model.calculate_clip_loss((roi + 1) * 0.5, emb)
                grad = torch.autograd.grad(score, roi, create_graph=False)[0]
            if em.save_interm:
                save_image((roi.detach().clamp(-1., 1.) + 1) * 0.5, str(trace_dir / f'iter_{it}.png'), nrow=4)
            with torch.no_grad():
                roi = (roi + strength * (norm(roi) / norm(grad)) * grad).clamp_(-1., 1.)
        image[:, :, y0:y0 + h, x0:x0 + w] = roi.detach()
        final = em.sample_via_scale(sample_batch_size, image, s=top, custom_t=num_denoising_steps, scale_mul=(1, 1))
        if save_images:
            n_aug = getattr(clip_model, "cfg", {}).get("n_aug", 0)
            tag = (f"clip_roi_{text_input.replace(' ', '_')}_n_aug{n_aug}_str_{strength}_n_iters_{num_clip_iters}"
                   f'_{str(datetime.datetime.now()).replace(":", "_")}')
            shown = (final + 1) * 0.5
            out_dir = Path(str(em.results_folder / 'final_samples'))
            out_dir.mkdir(parents=True, exist_ok=True)
            save_image(shown, str(out_dir / (tag + '.png')), nrow=4)
            if save_unbatched:
                single_dir = Path(str(self.results_folder / f'final_samples_unbatched_{tag}'))
                single_dir.mkdir(parents=True, exist_ok=True)
                for b in range(sample_batch_size):
                    save_image(shown[b], str(single_dir / f'{tag}_out_b{b}.png'))
        return final
