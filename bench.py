#!/usr/bin/env python3
"""bench.py -- decode throughput of the CDC hot path on MI355X.

One "step" = one pass of the hot path over one batch: a full `sample_steps`-iteration DDIM decode
(GaussianDiffusion.p_sample_loop) of `batch` synthetic 256x256 images per GPU, inputs (init noise +
context pyramid) resident in HBM when the timed region starts.  Default workload = BASELINE.json
configs[1]: x-param, batch 32, 256x256, 500 steps, 1 MI355X.  Multi-GPU: one process per GPU
(torchrun), the image batch is sharded (weak scaling: 32 images per GPU), the only collective is
the final all_gather of the decoded images over RCCL.

    python bench.py                      # N=1, 1 timed decode (~1 min) + CPU baseline sample
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 1 --warmup 0
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FULL = {
    "x": dict(kw=dict(dim=64, channels=3, context_channels=64, dim_mults=(1, 2, 3, 4, 5, 6),
                      context_dim_mults=(1, 2, 3, 4)), ctx=[64, 64, 128, 192], T=8193, vs="cosine",
              gflop_per_image_step=128.97, gb_per_image_step=0.714),
    "eps": dict(kw=dict(dim=64, channels=3, context_channels=3, dim_mults=(1, 2, 3, 4, 5, 6),
                        context_dim_mults=(1, 2, 3, 4)), ctx=[3, 64, 128, 192], T=20000, vs="linear",
                gflop_per_image_step=103.39, gb_per_image_step=0.682),
}
PEAK_F32_MFMA_TFLOPS = 157.3    # /opt/skills/guides/MI355X_MICROARCH.md: dense f32-input MFMA peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide: dense bf16 MFMA peak
# The 3x3 Block convolutions run fp32-exact products on the bf16 matrix cores: every fp32 operand is the
# exact sum of three bf16 numbers and six bf16 products reproduce the fp32 product (conv_split_kernel.h),
# so the matrix-core ceiling for one ALGORITHMIC fp32 flop is the bf16 peak / 6.
PEAK_SPLIT_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0


def cpu_baseline(param, size, sample_steps, n_iter=2):
    """Oracle (CPU restatement of the reference, kind="port") on the host cores: B=1, n_iter DDIM
    iterations, extrapolated linearly to `sample_steps` (time is iteration-linear)."""
    from cdc_compression_amd import synth
    from oracle import model as om
    from oracle import ops as oops
    cfgd = FULL[param]
    cfg = om.UnetConfig(**cfgd["kw"])
    sd = synth.unet_state_dict(om.unet_manifest(cfg), seed=0, final_gain=0.2 if param == "eps" else 1.0)
    O = oops.OrcOps("f32")
    rng = np.random.default_rng(0)
    ctx = [rng.standard_normal((1, c, size >> l, size >> l)).astype(np.float32) * 0.5
           for l, c in enumerate(cfgd["ctx"])]
    x = rng.standard_normal((1, 3, size, size)).astype(np.float32) * 0.8
    sched = om.Schedule(cfgd["T"], cfgd["vs"], param).set_sample_schedule(sample_steps)
    t0 = time.time()
    for i in range(n_iter):
        x = om.ddim_step(O, cfg, sd, sched, x, sample_steps - 1 - i, ctx, None,
                         True if param == "x" else "none")
    dt = (time.time() - t0) / n_iter
    return {"value": 1.0 / (dt * sample_steps), "unit": "images/s", "cores": os.cpu_count(),
            "kind": "port",
            "sample": f"oracle/ (C+OpenMP restatement), 1 image x {n_iter} of {sample_steps} DDIM "
                      f"iterations at {size}x{size}, {dt:.2f} s/iteration, extrapolated linearly"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1, help="timed batch decodes")
    ap.add_argument("--warmup", type=int, default=0, help="untimed batch decodes")
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--sample-steps", type=int, default=500)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--param", choices=["x", "eps"], default="x")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prof-every", type=int, default=50)
    a = ap.parse_args()

    import torch
    import cdc_compression_amd as cdc
    from cdc_compression_amd import _lib, synth

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or "RANK" in os.environ       # under torchrun always (exercises the RCCL path at N=1 too)
    if use_dist:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    cfgd = FULL[a.param]
    un = cdc.Unet(**cfgd["kw"], device=local)
    un.load_state_dict(synth.unet_state_dict(un.manifest(), seed=0,
                                             final_gain=0.2 if a.param == "eps" else 1.0))
    if a.param == "x":
        diff = cdc.GaussianDiffusionX(un, None, None, num_timesteps=cfgd["T"], pred_mode="x",
                                      var_schedule=cfgd["vs"])
    else:
        diff = cdc.GaussianDiffusionEps(un, None, num_timesteps=cfgd["T"], clip_noise="none",
                                        pred_mode="noise", var_schedule=cfgd["vs"])
    B, S = a.batch, a.size
    gen = torch.Generator(device=dev).manual_seed(1000 + rank)
    init = torch.randn((B, 3, S, S), generator=gen, device=dev) * 0.8          # gamma 0.8
    ctx = [torch.randn((B, c, S >> l, S >> l), generator=gen, device=dev) * 0.5
           for l, c in enumerate(cfgd["ctx"])]
    shape = (B, 3, S, S)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    diff.decompress(ctx, shape, sample_steps=2, init=init)      # builds the launch program, pages in code
    for _ in range(a.warmup):
        diff.decompress(ctx, shape, sample_steps=a.sample_steps, init=init)
    L, h = _lib.lib(), un._handle()
    L.cdc_prof_reset(h)
    L.cdc_prof_enable(h, max(2, a.prof_every))
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        rec = diff.decompress(ctx, shape, sample_steps=a.sample_steps, init=init)
        if use_dist:
            gathered = [torch.empty_like(rec) for _ in range(world)]
            dist.all_gather(gathered, rec)                      # the trivial result gather (RCCL)
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ok = bool(torch.isfinite(rec).all().item())

    import ctypes
    classes = {}
    for c in range(L.cdc_prof_num_classes()):
        ms, n, fl, by = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
        L.cdc_prof_get(h, c, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl), ctypes.byref(by))
        classes[L.cdc_prof_name(c).decode()] = dict(ms=ms.value, launches=n.value, flops=fl.value,
                                                   bytes=by.value)
    L.cdc_prof_enable(h, 0)

    if rank == 0:
        images = B * world * a.steps
        value = images / dt
        split = not os.environ.get("CDC_NO_SPLIT")
        peak = PEAK_SPLIT_TFLOPS if split else PEAK_F32_MFMA_TFLOPS
        dom = classes["conv3x3"]
        ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
        tot_ms = sum(c["ms"] for c in classes.values())
        out = {
            "metric": f"decoded images/sec at {S}x{S}, {a.sample_steps}-step {a.param}-param",
            "value": value, "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "dtype_note": "float32 tensors and accumulation; k x k convolution products as exact 3-way bf16 splits",
            "config": {"workload": f"{a.param}-param decode, batch={B}/GPU synthetic {S}x{S}, "
                                   f"{a.sample_steps} DDIM steps (BASELINE configs[1] shape)",
                       "batch_per_gpu": B, "global_batch": B * world, "sample_steps": a.sample_steps,
                       "parallelism": f"batch-shard x{world}", "finite": ok},
            "roofline": {
                "bound": "mfma",
                "kernel": ("conv_split2_kernel / conv_split_kernel (3x3 Block convolutions, fused LN epilogue; "
                           "3-way bf16 split operands, 6 bf16 MFMA products per fp32 product, f32 accumulate)"
                           if split else "conv_mfma_kernel (3x3 Block convolutions, v_mfma_f32_32x32x2_f32)"),
                "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                "peak_basis": ("2500 TFLOP/s dense bf16 MFMA / 6 products per algorithmic fp32 product"
                               if split else "157.3 TFLOP/s dense f32-input MFMA"),
                "frac_of_f32_mfma_peak": ach / PEAK_F32_MFMA_TFLOPS,
                "avg_launch_ms": dom["ms"] / max(dom["launches"], 1),
                "flops_per_launch": dom["flops"] / max(dom["launches"], 1),
                "traffic": None,
                # PMC pass of the dominant launch shape, collected separately (rocprofv3 --pmc FETCH_SIZE /
                # WRITE_SIZE, gfx950 x2 read correction): profiles/pmc_r01_v6_conv3x3_64_256_traffic.txt
                "traffic_sample": {"launch": "conv_split2_kernel<2,2,0>, 64->64 3x3 @256x256, batch 32",
                                   "hbm_bytes": 1.0998e9, "algorithmic_bytes": 1.0737e9},
                "whole_path_tflops": cfgd["gflop_per_image_step"] * 1e-3 * a.sample_steps * value,
                # both whole-path terms on the canonical (fused-minimum) work
                "whole_path_mfma_frac_of_split_peak": cfgd["gflop_per_image_step"] * 1e-3 * a.sample_steps * value / peak,
                "whole_path_hbm_frac": ((cfgd["gb_per_image_step"] + 0.160 / B) * 1e9 * a.sample_steps * value) / 8e12,
                "class_ms_share": {k: (v["ms"] / tot_ms if tot_ms else 0) for k, v in classes.items()},
                "class_tflops": {k: (v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0)
                                 for k, v in classes.items()},
            },
        }
        if world == 1 and a.param == "x" and S % 64 == 0:
            # informational (outside the timed region): the compressor on the GPU -- Compressor.forward (analysis
            # transform, hyper encoder/decoder, quantisers, rate estimate, synthesis transform) and decode alone
            comp = cdc.ResnetCompressor(dim=64, dim_mults=[1, 2, 3, 4], reverse_dim_mults=[4, 3, 2, 1],
                                        hyper_dims_mults=[4, 4, 4], channels=3, out_channels=64, device=local)
            man = comp.manifest() + comp.hyper_manifest() + comp.encoder_manifest()
            csd = synth.unet_state_dict(man, seed=5)
            C0 = comp.reversed_hyper_dims[0]
            pd = (1, 3, 3, 3, 1)
            for i in range(4):
                csd[f"prior.affine.{i}.weight"] = synth.normal(f"pw{i}", (C0, 1, 1, pd[i], pd[i + 1]), 5, 1.0)
                csd[f"prior.affine.{i}.bias"] = synth.normal(f"pb{i}", (C0, 1, 1, 1, pd[i + 1]), 5, 0.1)
                if i < 3:
                    csd[f"prior.a.{i}"] = synth.normal(f"pa{i}", (C0, 1, 1, 1, pd[i + 1]), 5, 0.5)
            comp.load_state_dict(csd)
            img = torch.rand((B, 3, S, S), generator=gen, device=dev) * 2 - 1
            q = torch.round(torch.randn((B, 256, S // 16, S // 16), generator=gen, device=dev) * 2.0)

            def timed(fn, n=5):
                fn()
                torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(n):
                    r = fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t) / n * 1e3, r
            ms_dec, pyr = timed(lambda: comp.decode(q))
            ms_fwd, fo = timed(lambda: comp(img))
            out["context_decode"] = {"ms_per_batch": ms_dec, "batch": B,
                                     "finite": bool(all(torch.isfinite(p).all().item() for p in pyr)),
                                     "note": "Compressor.decode (SURVEY 8f row 1), once per image, not in `value`"}
            out["compressor_forward"] = {"ms_per_batch": ms_fwd, "batch": B,
                                         "finite": bool(torch.isfinite(fo["bpp"]).all().item()),
                                         "note": "Compressor.forward = encode + bpp + decode (SURVEY 8f rows 1-3), "
                                                 "once per image, not in `value`"}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.param, S, a.sample_steps)
        out["roofline"]["class_ms_per_ddim_iter"] = {k: v["ms"] / max(1, len([i for i in range(a.sample_steps) if i % max(2, a.prof_every) == 0]) * a.steps) for k, v in classes.items()}
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
