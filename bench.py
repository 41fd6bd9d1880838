#!/usr/bin/env python3
"""Benchmark of the SinDDM multi-scale diffusion hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Headline metric (BASELINE.json): diffusion steps/sec at the finest scale, on config C2 (balloons,
5-scale pyramid, T=1000, batch 16 per GPU, finest scale 186x248, dim=160, fp32, synthetic closed-form
weights, torch.randn noise).  One "step" = one reverse diffusion step (p_sample: SinDDMNet forward +
fused reverse-step kernel + noise draw) for the whole per-GPU batch.  `value` = sample-steps/s
summed over all ranks (weak scaling: every rank runs its own 16 independent chains, no data-path
collective).  The second half of the metric, images/s of a FULL multi-scale sample (all 5 scales,
2478 network evaluations per image, RCCL all-gather of the results at the end), is measured once
after the timed region and reported as `full_sample` in the same JSON line.

Also in the line: `roofline` for the dominant kernel (the fp32-MFMA implicit-GEMM conv, measured with
HIP events around every conv launch of the timed region, on the stream they are launched on) and
`cpu_baseline` (the oracle's CPU restatement of the same step, timed on the host cores, rank 0, N=1).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FP32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md (v_mfma_f32_16x16x4_f32 rate)
NET_FLOP_PER_PIXEL = 2_150_230     # one SinDDMNet forward, per pixel per sample


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C2")
    ap.add_argument("--batch", type=int, default=None, help="chains per GPU (default: the config's batch)")
    ap.add_argument("--no-full", action="store_true", help="skip the full multi-scale sample leg")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--seed", type=int, default=1234)
    return ap.parse_args()


def main():
    args = parse()
    import torch.distributed as td
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback of the hot path exists)")
    # test hooks (a 1-GPU box cannot run RCCL with 2 ranks): SINDDM_BENCH_BACKEND=gloo stages the tiny timing
    # collectives through the host and SINDDM_BENCH_ONE_DEVICE=1 puts every rank on cuda:0; production = nccl/RCCL
    backend = os.environ.get("SINDDM_BENCH_BACKEND", "nccl")
    if os.environ.get("SINDDM_BENCH_ONE_DEVICE", "0") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            td.init_process_group(backend="nccl", device_id=dev)
        else:
            td.init_process_group(backend=backend)

    def allreduce_max(t):
        if world > 1:
            if backend == "nccl":
                td.all_reduce(t, op=td.ReduceOp.MAX)
            else:
                h = t.cpu()
                td.all_reduce(h, op=td.ReduceOp.MAX)
                t.copy_(h)
        return t
    assert world == args.gpus or world == 1, f"launched with WORLD_SIZE={world} but --gpus {args.gpus}"

    from sinddm_amd import _lib
    from sinddm_amd.configs import CONFIGS, build_diffusion
    lib = _lib.load()
    cfg = CONFIGS[args.config]
    B = args.batch or (cfg["batch"] if args.config in ("C1", "C2", "C3") else max(1, cfg["batch"] // 8))
    torch.manual_seed(args.seed + rank)
    net, d = build_diffusion(args.config, dim=160, device=dev)
    n_scales = len(cfg["sizes"])
    s = n_scales - 1
    mul = cfg.get("scale_mul", (1, 1))
    H, W = d.target_size(s, mul, True, s)
    total_t = d.num_timesteps_ideal[s]

    # state of a chain that has just arrived at the finest scale
    x_tilde = torch.randn(B, 3, H, W, device=dev).clamp_(-1, 1)
    d.img_prev_upsample = x_tilde
    img = d._q_sample_impl(x_tilde, None, total_t, torch.randn_like(x_tilde))
    t_seq = [(total_t - 1 - i) % total_t for i in range(args.warmup + args.steps)]

    def barrier():
        if world > 1:
            td.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        img = d._p_sample_host_t(img, t_seq[i], s)
    barrier()
    lib.sinddm_prof_begin()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        img = d._p_sample_host_t(img, t_seq[i], s)
    barrier()
    dt = time.perf_counter() - t0
    def prof(kind, reset):
        ms, n, fl, ex = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
        _lib.check(lib.sinddm_prof_end3(kind, C.byref(ms), C.byref(n), C.byref(fl), C.byref(ex), reset), "sinddm_prof_end3")
        return ms.value, n.value, fl.value, ex.value

    wino = os.environ.get("SINDDM_CONV_WINO", "1") != "0"
    dom_ms, dom_n, dom_fl, dom_ex = prof(1 if wino else 3, 0)        # the dominant kernel family only
    all_ms, all_n, all_fl, all_ex = prof(0, 1)                      # every MFMA convolution of the step
    tt = allreduce_max(torch.tensor([dt], device=dev, dtype=torch.float64))
    dt = float(tt)
    assert torch.isfinite(img).all()

    ms_per_step = dt / args.steps * 1e3
    value = world * B * args.steps / dt

    # ---- roofline of the dominant kernel ----
    # achieved = ALGORITHMIC FLOPs per launch (direct-convolution count 2*9*Cin*Cout per pixel of the layers this
    # kernel runs; the Winograd kernel executes 16/36 of them) / its average launch duration, measured with HIP events
    # around every launch on the launch stream inside the timed region.  The same kernel's average duration in
    # profiles/*_bench_kernel_stats.txt (rocprofv3 --kernel-trace --stats) must agree with avg_launch_ms.
    px = B * H * W
    avg_launch_ms = dom_ms / max(1, dom_n)
    achieved = dom_fl / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
    executed = dom_ex / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
    traffic = None
    tpath = os.path.join(REPO, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("conv_bytes_per_launch")   # PMC passes, profiles/r01k_pmc_summary.txt
        except Exception:
            traffic = None
    roofline = {"bound": "mfma",
                "kernel": ("conv_wino_kernel<5,3,*> (Winograd F(2x2,3x3) 3x3 conv on fp32 v_mfma_f32_16x16x4_f32; 7 launches "
                           "per step)" if wino else
                           "conv_mfma_dma_kernel<5,2,0,8> (fp32 v_mfma_f32_16x16x4_f32 implicit-GEMM 3x3 conv)"),
                "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                "algorithmic_flops_per_launch": round(dom_fl / max(1, dom_n)),
                "executed_tflops": round(executed, 2), "executed_frac": round(executed / FP32_MFMA_PEAK_TFLOPS, 4),
                "avg_launch_ms": round(avg_launch_ms, 4), "launches": int(dom_n),
                "share_of_step": round(dom_ms / (dt * 1e3), 4),
                "all_mfma_convs": {"achieved": round(all_fl / (all_ms * 1e-3) / 1e12, 2) if all_ms > 0 else 0.0,
                                   "launches": int(all_n), "share_of_step": round(all_ms / (dt * 1e3), 4)},
                "net_tflops_whole_step": round(NET_FLOP_PER_PIXEL * px * args.steps / dt / 1e12, 2)}

    # ---- second half of the metric: one FULL multi-scale sample (all scales + all-gather) ----
    full = None
    if not args.no_full:
        barrier()
        t0 = time.perf_counter()
        cur = d.sample(batch_size=B, scale_0_size=d.target_size(0, mul, True, 0), s=0)
        for si in range(1, n_scales):
            cur = d.sample_via_scale(B, cur, s=si, scale_mul=mul, custom_sample=True, custom_img_size_idx=si,
                                     custom_t=d.num_timesteps_ideal[si])
        if world > 1:
            from sinddm_amd import dist as sdist
            if backend == "nccl":
                out = torch.empty((world * B,) + tuple(cur.shape[1:]), device=dev)
                td.all_gather_into_tensor(out, cur.contiguous())
                cur = out
            else:
                cur = sdist.gather_batch(cur.cpu(), world * B).to(dev)
        barrier()
        ft = float(allreduce_max(torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)))
        evals = sum(d.num_timesteps_ideal)
        pix_steps = 0
        for si in range(n_scales):
            h, w = d.target_size(si, mul, True, si)
            pix_steps += h * w * d.num_timesteps_ideal[si]
        full = {"imgs_per_sec": round(world * B / ft, 4), "seconds": round(ft, 3), "images": world * B,
                "net_evals_per_image": evals,
                "net_tflops": round(NET_FLOP_PER_PIXEL * pix_steps * B * world / ft / 1e12, 2),
                "finite": bool(torch.isfinite(cur).all())}

    # ---- CPU baseline: the oracle's restatement of the same step on the host cores ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle import sinddm_oracle as O
        from sinddm_amd.synth import closed_form_state_dict
        ncpu = os.cpu_count() or 1
        sd = closed_form_state_dict(160)
        sched = O.make_schedule(cfg["T"], n_scales, cfg["rescale_losses"], 1, train_full_t=True)
        # pick the thread count that is actually fastest for this op mix (oneDNN convs stop scaling long before
        # 256 threads): a short calibration on a small image, then the measurement with the winner
        cands = sorted({ncpu, min(ncpu, 64), min(ncpu, 32)})
        best, cores = None, ncpu
        xs = torch.randn(1, 3, 94, 126)
        for nthr in cands:
            torch.set_num_threads(nthr)
            with torch.no_grad():
                O.net_forward(sd, xs, torch.tensor([5]), 2)
                t0 = time.perf_counter()
                O.net_forward(sd, xs, torch.tensor([5]), 2)
                dtc = time.perf_counter() - t0
            if best is None or dtc < best:
                best, cores = dtc, nthr
        torch.set_num_threads(cores)
        cb = min(B, 4)
        xc = torch.randn(cb, 3, H, W)
        xt = torch.randn(cb, 3, H, W)
        zc = torch.randn(cb, 3, H, W)
        with torch.no_grad():
            O.p_sample(sched, sd, xc, total_t - 1, s, zc, xt)          # warm
            n_cpu, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < 10.0 and n_cpu < 50:
                xc = O.p_sample(sched, sd, xc, total_t - 2 - n_cpu, s, zc, xt)
                n_cpu += 1
            ct = time.perf_counter() - t0
        cpu = {"value": round(cb * n_cpu / ct, 3), "unit": "sample-steps/s", "cores": cores, "kind": "port",
               "sample": f"{n_cpu} finest-scale p_sample steps of batch {cb} at {H}x{W} "
                         f"(oracle/sinddm_oracle.py, torch CPU fp32, {cores} threads)"}

    if rank == 0:
        line = {
            "metric": "diffusion steps/sec (finest scale) + imgs/sec full multi-scale sample, 1/2/4/8 GPU",
            "value": round(value, 3), "unit": "sample-steps/s (finest scale, batch x steps/s, all GPUs)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "steps_per_sec_per_gpu": round(args.steps / dt, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (closed-form weights of the dim=160 architecture, torch.randn noise/images)",
            "config": {"workload": f"{args.config}: balloons 5-scale pyramid, T=1000, finest scale {H}x{W}, "
                                   f"batch {B} per GPU, dim=160" if args.config == "C2" else
                                   f"{args.config}: finest scale {H}x{W}, T={cfg['T']}, batch {B} per GPU, dim=160",
                       "batch_per_gpu": B, "global_batch": B * world, "finest_hw": [H, W], "scale": s,
                       "parallelism": f"independent chains x{world}"},
            "full_sample": full, "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        td.destroy_process_group()


if __name__ == "__main__":
    main()
