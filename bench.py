#!/usr/bin/env python3
"""bench.py -- decode throughput of the CDC hot path on MI355X.

One "step" = one pass of the hot path over one batch: a full `sample_steps`-iteration DDIM decode
(GaussianDiffusion.p_sample_loop) of `batch` synthetic images per GPU, inputs (init noise + context
pyramid) resident in HBM when the timed region starts.  Default workload = BASELINE.json configs[1]:
x-param, batch 32, 256x256, 500 steps, 1 MI355X.  Multi-GPU: one process per GPU (torchrun), the image
batch is sharded (weak scaling: `batch` images per GPU) through cdc_compression_amd.parallel.sharded_decode;
the only collective is its final all_gather of the decoded images over RCCL.

    python bench.py                      # N=1, 1 timed decode + verification + CPU baseline sample
    python bench.py --gpus 8             # starts its 8 ranks itself (one per GPU, RCCL; self_launch below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 1 --warmup 0      # the same under an external launcher
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process GPU work on this platform needs dmabuf IPC (RCCL / CUDA-tensor sharing fail with "hipIpcGetMemHandle: invalid
# argument" otherwise); set before torch / HIP initialise, for every rank torchrun starts from this script
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

# canonical (reference op list, fused-minimum bytes) work per image and DDIM iteration at 256x256;
# both scale with the pixel count.
FULL = {
    "x": dict(kw=dict(dim=64, channels=3, context_channels=64, dim_mults=(1, 2, 3, 4, 5, 6),
                      context_dim_mults=(1, 2, 3, 4)), ctx=[64, 64, 128, 192], T=8193, vs="cosine",
              gflop_per_image_step=128.97, gb_per_image_step=0.714),
    "eps": dict(kw=dict(dim=64, channels=3, context_channels=3, dim_mults=(1, 2, 3, 4, 5, 6),
                        context_dim_mults=(1, 2, 3, 4)), ctx=[3, 64, 128, 192], T=20000, vs="linear",
                gflop_per_image_step=103.39, gb_per_image_step=0.682),
}
PEAK_F32_MFMA_TFLOPS = 157.3    # /opt/skills/guides/MI355X_MICROARCH.md: dense f32-input MFMA peak
PEAK_16BIT_MFMA_TFLOPS = 2500.0  # same guide: dense bf16 / fp16 MFMA peak
# The split convolutions form every fp32 product on the 16-bit matrix cores: three fp16 MFMA products (two-plane
# fp16 operands, CDC_ARITH_F16X2, default) or six bf16 products (exact three-plane bf16, CDC_ARITH_BF16X3) per
# ALGORITHMIC fp32 product -- the matrix-core ceiling for algorithmic flops is the 16-bit peak / 3 resp. / 6.
PRODUCTS = {1: 3, 0: 6}
# the true reference (PyTorch 2.10 CPU, oneDNN) probed in the build container
REFERENCE_PROBE = {"value": 0.003, "unit": "images/s", "cores": 8, "kind": "reference",
                   "sample": "BASELINE.md section 2: reference compress(), B=2, 256x256, 4 of 500 steps on 8 container "
                             "cores (0.7 s per image-step), extrapolated; NOT measured on the GPU box"}


def workload_name(param, B, S, steps, world):
    known = {("x", 32, 256, 500): "BASELINE configs[1]", ("eps", 32, 256, 1000): "BASELINE configs[2]",
             ("x", 16, 512, 500): "BASELINE configs[4]"}
    tag = known.get((param, B, S, steps))
    if param == "x" and B == 32 and S == 256 and steps == 500 and world == 8:
        tag = "BASELINE configs[3]"
    return (f"{param}-param decode, batch={B}/GPU synthetic {S}x{S}, {steps} DDIM steps"
            + (f" ({tag})" if tag else " (not a BASELINE configuration)"))


def cpu_baseline(param, size, sample_steps, n_iter=4):
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
    x = om.ddim_step(O, cfg, sd, sched, x, sample_steps - 1, ctx, None, True if param == "x" else "none")   # (untimed: pages, threads)
    t0 = time.time()
    x = om.ddim_step(O, cfg, sd, sched, x, sample_steps - 2, ctx, None, True if param == "x" else "none")
    n_iter = max(n_iter, min(40, int(15.0 / max(time.time() - t0, 1e-3))))      # about 15 s of CPU work
    t0 = time.time()
    for i in range(n_iter):
        x = om.ddim_step(O, cfg, sd, sched, x, sample_steps - 3 - i, ctx, None,
                         True if param == "x" else "none")
    dt = (time.time() - t0) / n_iter
    return {"value": 1.0 / (dt * sample_steps), "unit": "images/s", "cores": O.threads,
            "kind": "port",
            "sample": f"oracle/ (C + OpenMP restatement of the reference's ATen calls; the stride-1 convolutions -- 93 % of the "
                      f"multiply-adds -- as a register-tiled direct convolution, AVX2; {O.threads} OpenMP threads = the container's CPU quota "
                      f"of {os.cpu_count()} visible CPUs), 1 image x {n_iter} of {sample_steps} DDIM "
                      f"iterations at {size}x{size}, {dt:.2f} s/iteration = {FULL[param]['gflop_per_image_step'] * (size / 256.0) ** 2 / dt:.0f} "
                      f"GFLOP/s, extrapolated linearly",
            "reference_probe": REFERENCE_PROBE}


def measure_ceilings(L, local):
    """The part's own ceilings, measured on the spot: csrc/probe.hip through the C-ABI, ~50 ms in all."""
    tf_r, tf_c, gbs = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    rc = [L.cdc_probe_mfma_f16(local, 1, 20000, ctypes.byref(tf_r)), L.cdc_probe_mfma_f16(local, 0, 20000, ctypes.byref(tf_c)),
          L.cdc_probe_hbm_copy(local, 1 << 30, 5, ctypes.byref(gbs))]
    if any(rc):
        return None
    return {"mfma": {"tflops_random_operands": tf_r.value, "tflops_constant_operands": tf_c.value,
                     "frac_of_nominal_random": tf_r.value / PEAK_16BIT_MFMA_TFLOPS, "nominal_tflops": PEAK_16BIT_MFMA_TFLOPS,
                     "note": "register-only v_mfma_f32_32x32x16_f16 loop, two waves per SIMD, best of 4 timed launches of 20000 x 16 instructions "
                             "per wave; random = eight pseudo-random operand pairs cycled (inputs toggle as on real data)"},
            "hbm": {"gb_per_s": gbs.value, "frac_of_nominal": gbs.value / 8000.0, "nominal_gb_per_s": 8000.0,
                    "note": "float4 copy of 1 GiB, bytes read + bytes written, best of 5 launches"}}


def arith_name(arith):
    return "f16x2" if arith == 1 else "bf16x3"


def dtype_label(arith):
    """`dtype` of the JSON line: float32 tensors and accumulation, products formed on the 16-bit matrix cores."""
    return "f32 (f16x2 split products)" if arith == 1 else "f32 (bf16x3 split products)"


def build_model(param, local):
    import cdc_compression_amd as cdc
    from cdc_compression_amd import synth
    cfgd = FULL[param]
    un = cdc.Unet(**cfgd["kw"], device=local)
    un.load_state_dict(synth.unet_state_dict(un.manifest(), seed=0, final_gain=0.2 if param == "eps" else 1.0))
    if param == "x":
        diff = cdc.GaussianDiffusionX(un, None, None, num_timesteps=cfgd["T"], pred_mode="x", var_schedule=cfgd["vs"])
    else:
        diff = cdc.GaussianDiffusionEps(un, None, num_timesteps=cfgd["T"], clip_noise="none", pred_mode="noise",
                                        var_schedule=cfgd["vs"])
    return un, diff, cfgd


def make_inputs(cfgd, B, S, dev, seed):
    import torch
    gen = torch.Generator(device=dev).manual_seed(seed)
    init = torch.randn((B, 3, S, S), generator=gen, device=dev) * 0.8          # gamma 0.8
    ctx = [torch.randn((B, c, S >> l, S >> l), generator=gen, device=dev) * 0.5 for l, c in enumerate(cfgd["ctx"])]
    return init, ctx, gen


def read_prof(L, h):
    """Per-class and per-op hipEvent tables of the handle (sampled DDIM iterations inside the timed region)."""
    classes = {}
    for c in range(L.cdc_prof_num_classes()):
        ms, n, fl, by = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
        L.cdc_prof_get(h, c, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl), ctypes.byref(by))
        classes[L.cdc_prof_name(c).decode()] = dict(ms=ms.value, launches=n.value, flops=fl.value, bytes=by.value)
    ops = []
    for i in range(L.cdc_prof_num_ops(h)):
        lab, ms, n, fl = ctypes.c_char_p(), ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
        L.cdc_prof_op(h, i, ctypes.byref(lab), ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl))
        if n.value:
            ops.append(dict(label=lab.value.decode(), ms=ms.value / n.value, n=n.value, flops=fl.value))
    return classes, ops


FAMILY = {"PF3": "conv_pf3_kernel", "PF": "conv_pf_kernel", "PW": "conv_pw_kernel", "SPLIT2H": "conv_split2_kernel",
          "SPLIT2": "conv_split2_kernel", "SPLIT": "conv_split_kernel", "CONV": "conv_mfma_kernel", "WS": "conv_ws_kernel", "WS1": "conv_ws1_kernel"}


def kern_of(label):
    t = label.split()
    return next((k for k in ("PF3", "PF", "PW", "WS1", "WS", "SPLIT2H", "SPLIT2", "SPLIT") if k in t), "CONV")


def op_bytes(label, B):
    """input once + output once (+ the residual operand), 4 bytes each."""
    t = label.split()
    cin, cout = (int(v) for v in t[3].split("->"))
    ho, wo = (int(v) for v in t[t.index("out") + 1].split("x"))
    st = int(t[2][1:]) if t[2].startswith("s") else 1
    return 4.0 * B * (cin * ho * wo * st * st + cout * ho * wo * (2 if ("+res" in t or "+resP" in t) else 1))     # (+resP: the residual read from a PF tensor, same bytes)


def roofline_block(classes, ops, B, S, arith, value, sample_steps, steps, dt, cfgd, prof_every, full=True, ceilings=None):
    """The `roofline` object of the JSON line.  Dominant kernel = the (layer shape, kernel) pair with the largest total time among
    the 3x3 stride-1 Block convolutions (the rule of rounds 1-3; epilogue variants of one kernel on one layer shape together);
    `by_shape` lists the other pairs, `families` the totals per kernel function (what rocprofv3 --stats rows add up to)."""
    products = PRODUCTS[arith]
    peak = PEAK_16BIT_MFMA_TFLOPS / products
    conv_ops = [o for o in ops if o["label"].startswith("conv ") and o["flops"] > 0]
    fams, groups = {}, {}
    for o in conv_ops:
        kern = kern_of(o["label"])
        f = fams.setdefault(FAMILY[kern], dict(ms=0.0, n=0, flops=0.0, bytes=0.0))
        f["ms"] += o["ms"]; f["n"] += 1; f["flops"] += o["flops"]; f["bytes"] += op_bytes(o["label"], B)
        key = " ".join(o["label"].split()[:6]) + " " + kern
        g = groups.setdefault(key, dict(ms=0.0, n=0, flops=0.0, bytes=0.0, family=FAMILY[kern], labels=[], raw=[]))
        g["ms"] += o["ms"]; g["n"] += 1; g["flops"] += o["flops"]; g["bytes"] += op_bytes(o["label"], B)
        g["labels"].append(f'{o["ms"]:.4f} ms  {o["label"]}')
        g["raw"].append(o["label"])
    cand = {k: g for k, g in groups.items() if k.startswith("conv 3x3 s1")}
    domk, dom = max(cand.items(), key=lambda kv: kv[1]["ms"]) if cand else ("", dict(ms=0, n=1, flops=0, bytes=0, family="", labels=[], raw=[]))
    dom_ms = dom["ms"] / max(dom["n"], 1)
    ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
    alg_bytes = dom["bytes"] / max(dom["n"], 1)
    out = {"bound": "mfma", "kernel": dom["family"], "launch_key": domk, "launches": dom["labels"],
           "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak if peak else 0,
           "avg_launch_ms": dom_ms, "launches_per_iteration": dom["n"],
           "ms_per_ddim_iter": dt / steps / sample_steps * 1e3}
    if not full:
        return out
    cls3 = classes["conv3x3"]
    cls_ach = cls3["flops"] / (cls3["ms"] * 1e-3) / 1e12 if cls3["ms"] > 0 else 0.0
    tot_ms = sum(c["ms"] for c in classes.values())
    n_prof_iters = max(1, len([i for i in range(sample_steps) if i % max(2, prof_every) == 0]) * steps)
    scale = (S / 256.0) ** 2
    canon_tf = cfgd["gflop_per_image_step"] * scale * 1e-3 * sample_steps * value
    exec_gflop_iter = sum(c["flops"] for c in classes.values()) / n_prof_iters / 1e9     # per batch iteration
    exec_tf = exec_gflop_iter * 1e-3 / B * sample_steps * value
    # Counter-based HBM traffic of the dominant pair, if PMC passes of THIS round's build were committed for it
    # (tools/gpu_profiles_r06.sh -> profiles/pmc_r06_traffic.json: {op label: {...}}, per-dispatch FETCH_SIZE / WRITE_SIZE of the
    # whole-path passes matched to the launch program's op labels).  `traffic` is the mean over the SAME launches that
    # `algorithmic_bytes_per_launch` averages (every epilogue variant of the pair), so the two figures compare like with like.
    traffic, traffic_src, traffic_variants, traffic_hash = None, None, None, None
    import glob
    from cdc_compression_amd._lib import kernel_source_hash
    build_hash = kernel_source_hash()
    for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "pmc_r*_traffic.json")), reverse=True):      # newest round first
        if not dom["raw"]:
            break
        try:
            tj = json.load(open(tpath))
            if tj.get("kernel_source_hash") != build_hash:      # counters of other kernels say nothing about this build
                continue
            per = tj.get("ops", {}) if tj.get("batch") == B and tj.get("arith") == arith_name(arith) else {}
            rows = [per.get(lab) for lab in dom["raw"]]
            if rows and all(rows):
                traffic = sum(r["hbm_bytes_corrected"] for r in rows) / len(rows)
                traffic_src = "from_file: " + os.path.relpath(tpath, ROOT) + " -- " + str(tj.get("source"))
                traffic_hash = tj.get("kernel_source_hash")
                traffic_variants = [{"launch": lab, "hbm_bytes_corrected": r["hbm_bytes_corrected"], "hbm_bytes_raw": r["hbm_bytes_raw"],
                                     "algorithmic_bytes": op_bytes(lab, B)} for lab, r in zip(dom["raw"], rows)]
                break
        except Exception:
            pass
    by_shape = sorted(({"launch_key": k, "family": g["family"], "launches_per_iteration": g["n"], "avg_launch_ms": g["ms"] / g["n"],
                        "achieved": g["flops"] / (g["ms"] * 1e-3) / 1e12, "frac": g["flops"] / (g["ms"] * 1e-3) / 1e12 / peak,
                        "algorithmic_tb_s": g["bytes"] / (g["ms"] * 1e-3) / 1e12, "launches": g["labels"]}
                       for k, g in groups.items()), key=lambda r: -r["avg_launch_ms"] * r["launches_per_iteration"])
    families = {k: {"launches_per_iteration": f["n"], "ms_per_iteration": f["ms"], "achieved": f["flops"] / (f["ms"] * 1e-3) / 1e12,
                    "frac": f["flops"] / (f["ms"] * 1e-3) / 1e12 / peak} for k, f in fams.items() if f["ms"] > 0}
    out.update({
        "by_shape": by_shape[:8], "families": families,
        "kernel_note": "dominant (layer shape, kernel) pair: largest total time among the 3x3 stride-1 Block convolutions (epilogue "
                       "variants of the kernel on that layer shape together); achieved = its algorithmic flops / the hipEvent-timed "
                       "average duration of its launches (sampled inside the timed region, on the launch stream); by_shape = the "
                       "next pairs, families = totals per kernel function (the rows rocprofv3 --stats adds up)",
        "hbm_view": ({"algorithmic_bytes_per_launch": alg_bytes, "achieved_tb_s": alg_bytes / (dom_ms * 1e-3) / 1e12,
                      "frac_of_8tb_s": alg_bytes / (dom_ms * 1e-3) / 8e12} if alg_bytes and dom_ms > 0 else None),
        "peak_basis": f"2500 TFLOP/s dense 16-bit MFMA / {products} products per algorithmic fp32 product",
        "mfma_tflops_executed": ach * products,
        # what THIS box sustains, measured by this run right after the timed region (csrc/probe.hip through the C-ABI): a register-only
        # loop of v_mfma_f32_32x32x16_f16 on constant / on pseudo-random operands, and a 1 GiB float4 copy.  `peak` / `frac` stay on the
        # guide's nominal 2500 TFLOP/s; these say how far the part itself is from it under load.
        "mfma_sustained_measured": ({**ceilings["mfma"], "frac_of_random_operand_loop": (ach * products / ceilings["mfma"]["tflops_random_operands"]
                                                                                     if ceilings["mfma"].get("tflops_random_operands") else None)}
                                    if ceilings else None),
        "hbm_copy_measured": ceilings["hbm"] if ceilings else None,
        "frac_of_f32_mfma_peak": ach / PEAK_F32_MFMA_TFLOPS,
        "flops_per_launch": dom["flops"] / max(dom["n"], 1), "algorithmic_bytes_per_launch": alg_bytes,
        "traffic": traffic, "traffic_source": traffic_src, "traffic_by_variant": traffic_variants,
        "traffic_kernel_source_hash": traffic_hash, "kernel_source_hash": build_hash,
        "traffic_note": "from a committed counter file, not re-measured by this run: mean corrected counter bytes (FETCH_SIZE x2 + WRITE_SIZE) "
                        "over the same launches that algorithmic_bytes_per_launch averages; null unless a PMC pass of exactly these kernel "
                        "sources (kernel_source_hash), batch and arithmetic is committed under profiles/",
        "class_conv3x3": {"achieved": cls_ach, "frac": cls_ach / peak if peak else 0,
                          "avg_launch_ms": cls3["ms"] / max(cls3["launches"], 1),
                          "flops_per_launch": cls3["flops"] / max(cls3["launches"], 1)},
        # whole-path terms.  canonical = the reference's op list (what a user gets per image);
        # executed = what the launch program really multiplies (context hoisting removes ~25 %)
        "whole_path_tflops_canonical": canon_tf,
        "whole_path_tflops_executed": exec_tf,
        "whole_path_mfma_frac_canonical": canon_tf / peak, "whole_path_mfma_frac_executed": exec_tf / peak,
        "whole_path_hbm_frac": ((cfgd["gb_per_image_step"] * scale + 0.160 / B) * 1e9 * sample_steps * value) / 8e12,
        "whole_path_hbm_note": "north_star's >= 40 % of the HBM roofline is not reachable in fp32-class arithmetic "
                               "(AI ~180 flop/B vs a ridge of ~100-300): the path is matrix-bound (SURVEY section 7)",
        "class_ms_share": {k: (v["ms"] / tot_ms if tot_ms else 0) for k, v in classes.items()},
        "class_tflops": {k: (v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0) for k, v in classes.items()},
        # algorithmic bytes of the class (each op: its inputs once + its outputs once) / its time
        "class_tb_per_s": {k: (v["bytes"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0) for k, v in classes.items()},
        "class_ms_per_ddim_iter": {k: v["ms"] / n_prof_iters for k, v in classes.items()},
    })
    # first-class scalars: the executed matrix rate of the dominant kernel against what THIS box's matrix pipe
    # sustains on random operands, the measured ceilings themselves, the launches sampled
    rnd = ceilings["mfma"].get("tflops_random_operands") if ceilings else None
    out["frac_of_measured_sustained"] = (ach * products / rnd) if rnd else None
    out["mfma_sustained_tflops_measured"] = rnd
    out["hbm_copy_tb_s_measured"] = (ceilings["hbm"]["gb_per_s"] / 1e3) if ceilings else None
    out["achieved_tb_s_algorithmic"] = alg_bytes / (dom_ms * 1e-3) / 1e12 if alg_bytes and dom_ms > 0 else None
    out["sampled_iterations"] = n_prof_iters
    # The driver's record keeps the scalar fields of this object in order and drops what is nested or comes late: numbers first,
    # prose and tables last.
    first = ["bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "launch_key", "avg_launch_ms", "launches_per_iteration",
             "ms_per_ddim_iter", "frac_of_measured_sustained", "mfma_sustained_tflops_measured", "hbm_copy_tb_s_measured",
             "mfma_tflops_executed", "achieved_tb_s_algorithmic", "flops_per_launch", "algorithmic_bytes_per_launch",
             "whole_path_tflops_canonical", "whole_path_mfma_frac_canonical", "whole_path_tflops_executed", "whole_path_mfma_frac_executed",
             "whole_path_hbm_frac", "frac_of_f32_mfma_peak", "sampled_iterations", "kernel_source_hash", "traffic_kernel_source_hash"]
    return {**{k: out[k] for k in first if k in out}, **{k: v for k, v in out.items() if k not in first}}


def launches_per_iter(L, h):
    """Kernel launches of one DDIM iteration of the handle's current launch program: its ops without the hoisted (once per decode)
    context convolutions and without the 7-row combine, which the sampler kernel evaluates -- that kernel is the + 1."""
    n = 0
    for i in range(L.cdc_prof_num_ops(h)):
        lab, ms, cnt, fl = ctypes.c_char_p(), ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
        L.cdc_prof_op(h, i, ctypes.byref(lab), ctypes.byref(ms), ctypes.byref(cnt), ctypes.byref(fl))
        t = lab.value.decode()
        if " HOIST" in t or t == "combine" or t == "temb":
            continue
        # (the attention fold is two kernels -- ctx_r0, fold_r12_mfma -- at the model's folded widths, ctx_r0 + R1 + R2 otherwise: aux_kernels.hip ctx_fold_launch)
        n += (2 if any(("C=%d " % c) in t for c in (64, 128, 192)) else 3) if t.startswith("ctxf ") else 1
    return n + 2          # + the time-embedding row copy + the sampler update


def verify_rows(decode_fn, init, ctx, rec, B, sample_steps, count_ops=None):
    """Rows 0 and B-1 of a timed decode decoded again on their own (batch-1 launch plans: different kernels, K splits and
    summation orders) must agree; also the batch-1 latency (what the reference's test scripts run: one image per call)."""
    import torch
    rows, errs, t1 = (0, B - 1), [], 0.0
    for k in rows:
        torch.cuda.synchronize()
        ta = time.perf_counter()
        r1 = decode_fn(init[k:k + 1], [c[k:k + 1] for c in ctx])
        torch.cuda.synchronize()
        t1 += time.perf_counter() - ta
        den = max(1.0, float(rec[k].abs().max().item()))
        errs.append(float((r1[0] - rec[k]).abs().max().item()) / den)
    verify = {"rows": list(rows), "max_rel_err_vs_batch1_decode": max(errs), "tolerance": 1e-4, "ok": bool(max(errs) <= 1e-4)}
    batch1 = {"images_per_s": len(rows) / t1, "ms_per_ddim_iter": t1 / len(rows) / sample_steps * 1e3,
              # kernel launches of one DDIM iteration of the batch-1 launch program (the U-Net's ops + the sampler update)
              "launches_per_iter": count_ops() if count_ops else None,
              "note": "one image per call, same model and step count (the reference test scripts' mode)"}
    return verify, batch1


def other_config(param, B, S, sample_steps, local, dev, prof_every):
    """One timed decode of another BASELINE.json configuration (same code path as the headline workload, its own model and
    launch program), verified against batch-1 decodes of two rows.  Reported beside the headline, never inside `value`."""
    import torch
    from cdc_compression_amd import _lib
    un, diff, cfgd = build_model(param, local)
    init, ctx, _ = make_inputs(cfgd, B, S, dev, 2000)

    def decode_fn(i, c, steps=None):
        return diff.decompress(c, (c[0].shape[0], 3, S, S), sample_steps=steps or sample_steps, init=i)

    decode_fn(init, ctx, steps=2)
    L, h = _lib.lib(), un._handle()
    L.cdc_prof_reset(h)
    L.cdc_prof_enable(h, max(2, prof_every))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rec = decode_fn(init, ctx)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    classes, ops = read_prof(L, h)
    L.cdc_prof_enable(h, 0)
    arith = L.cdc_get_arith(h)
    value = B / dt
    rl = roofline_block(classes, ops, B, S, arith, value, sample_steps, 1, dt, cfgd, prof_every, full=False)
    verify, batch1 = verify_rows(decode_fn, init, ctx, rec, B, sample_steps)
    res = {"workload": workload_name(param, B, S, sample_steps, 1), "value": value, "unit": "images/s", "ms_per_step": dt * 1e3,
           "ms_per_ddim_iter": dt / sample_steps * 1e3, "dtype": dtype_label(arith),
           "finite": bool(torch.isfinite(rec).all().item()) and verify["ok"],
           "roofline": {k: rl[k] for k in ("kernel", "launch_key", "achieved", "peak", "unit", "frac", "avg_launch_ms")},
           "whole_path_tflops_canonical": cfgd["gflop_per_image_step"] * (S / 256.0) ** 2 * 1e-3 * sample_steps * value,
           "verify": verify, "batch1_ms_per_ddim_iter": batch1["ms_per_ddim_iter"], "range_guard": _lib.handle_status(h),
           "note": "one timed decode, inputs resident in HBM, outside `value`"}
    del diff, un
    return res


def self_launch(n):
    """`python bench.py --gpus N` without a launcher (the form the driver uses at N = 1): start the N ranks here -- one child
    process of this same command line per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment, rendezvous on
    127.0.0.1 at a free port -- and wait for them.  Rank 0's single JSON line goes to this process's stdout unchanged.  The first
    rank to fail ends the others (by their PIDs); the exit code is that rank's."""
    import socket
    import subprocess
    import tempfile
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    # The ranks rendezvous through a FILE store (CDC_BENCH_STORE_FILE -> init_process_group(init_method="file://...")): a free port found
    # by bind + close can be taken by another process before rank 0 binds it.  MASTER_ADDR / MASTER_PORT are still exported for
    # anything that reads them, but nothing of this script listens there.
    store_dir = tempfile.mkdtemp(prefix="cdc_bench_store_")
    store = os.path.join(store_dir, "store")
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CDC_BENCH_SELF_LAUNCHED="1", CDC_BENCH_STORE_FILE=store)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc, live = 0, list(procs)
    try:
        while live:
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in live:
                        q.terminate()
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        import shutil
        shutil.rmtree(store_dir, ignore_errors=True)
    return rc


def init_pg(dist, backend, **kw):
    """env:// under an external launcher; the launcher's file store when this script started its own ranks (self_launch)."""
    store = os.environ.get("CDC_BENCH_STORE_FILE")
    if store:
        dist.init_process_group(backend, init_method="file://" + store, rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]), **kw)
    else:
        dist.init_process_group(backend, **kw)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1, help="timed batch decodes")
    ap.add_argument("--warmup", type=int, default=0, help="untimed batch decodes")
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--sample-steps", type=int, default=None, help="DDIM iterations (default 500 x-param, 1000 eps-param)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--param", choices=["x", "eps"], default="x")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="process-group backend: nccl = RCCL over xGMI (one rank per GPU); gloo lets several ranks share one GPU "
                         "(dry run of the multi-rank path on a 1-GPU box: rank r uses cuda:(LOCAL_RANK mod device count))")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the B=1 re-decode of two output rows")
    ap.add_argument("--no-alt-arith", action="store_true", help="skip the extra decode in the exact bf16x3 arithmetic")
    ap.add_argument("--no-extras", action="store_true", help="skip the informational compressor / entropy coder legs (profiling runs)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the timed decodes of BASELINE configs[2] and configs[4]")
    ap.add_argument("--dump-ops", default=None, help="write the launch program's op labels (program order) to this file (profiling tools)")
    ap.add_argument("--prof-every", type=int, default=125,
                    help="hipEvent pairs around every launch of the DDIM iterations i %% N == 0 (inside the timed region: roofline.sampling)")
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous only: every rank joins the process group, one all_reduce counts the ranks, rank 0 prints "
                         "{launch_check, ranks_seen, launcher}; no GPU work (CPU test of the self-launch path, gloo)")
    a = ap.parse_args()
    if a.sample_steps is None:
        a.sample_steps = 500 if a.param == "x" else 1000

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # bare `python bench.py --gpus N`: this process is only the launcher of its N ranks (one per GPU)
        raise SystemExit(self_launch(a.gpus))
    launcher = ("self" if os.environ.get("CDC_BENCH_SELF_LAUNCHED") else "external") if "WORLD_SIZE" in os.environ else None
    if a.launch_check:
        import torch
        import torch.distributed as dist
        if int(os.environ.get("WORLD_SIZE", 1)) != a.gpus:
            raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE')}")
        init_pg(dist, "gloo")
        one = torch.ones(1)
        dist.all_reduce(one)
        if dist.get_rank() == 0:
            print(json.dumps({"launch_check": True, "ranks_seen": int(one.item()), "n_gpus": a.gpus, "launcher": launcher}), flush=True)
        dist.destroy_process_group()
        return

    import torch
    import cdc_compression_amd as cdc
    from cdc_compression_amd import _lib, parallel, synth

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")
    if a.backend == "gloo":
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or "RANK" in os.environ       # under torchrun always (exercises the RCCL path at N=1 too)
    dist = None
    if use_dist:
        import torch.distributed as dist
        if a.backend == "nccl":
            init_pg(dist, "nccl", device_id=dev)
        else:
            init_pg(dist, "gloo")

    un, diff, cfgd = build_model(a.param, local)
    B, S = a.batch, a.size
    init, ctx, gen = make_inputs(cfgd, B, S, dev, 1000 + rank)

    def decode_fn(i, c, steps=None):
        return diff.decompress(c, (c[0].shape[0], 3, S, S), sample_steps=steps or a.sample_steps, init=i)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    decode_fn(init, ctx, steps=2)      # builds the launch program, pages in code
    for _ in range(a.warmup):
        decode_fn(init, ctx)
    L, h = _lib.lib(), un._handle()
    L.cdc_prof_reset(h)
    L.cdc_prof_enable(h, max(2, a.prof_every))
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        # this rank's shard of the B*world-image job; the gather of the decoded images is the only collective
        full = parallel.sharded_decode(decode_fn, init, ctx, world, rank, dist, global_batch=B * world)
    barrier()
    dt = time.perf_counter() - t0
    ranks_seen = 1
    if use_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        one = torch.ones(1, device=dev)
        dist.all_reduce(one)
        ranks_seen = int(one.item())
    lo, hi = parallel.shard_bounds(B * world, world, rank)
    rec = full[lo:hi]
    ok = bool(torch.isfinite(full).all().item()) and tuple(full.shape) == (B * world, 3, S, S)
    arith = L.cdc_get_arith(h)

    classes, ops = read_prof(L, h)
    L.cdc_prof_enable(h, 0)
    if rank == 0 and os.environ.get("CDC_BENCH_OPS"):       # development aid: per-op table (hipEvent averages) on stderr
        order = ops if os.environ.get("CDC_BENCH_OPS_ORDER") else sorted(ops, key=lambda o: -o["ms"])   # program order / by time
        for o in order[: int(os.environ["CDC_BENCH_OPS"])]:
            print(f'[op] {o["ms"]:8.4f} ms  {o["flops"] / max(o["ms"], 1e-9) / 1e9:7.1f} TF  {o["label"]}', file=sys.stderr)
    if rank == 0 and a.dump_ops:
        with open(a.dump_ops, "w") as f:
            for i in range(L.cdc_prof_num_ops(h)):
                lab, ms, n, fl = ctypes.c_char_p(), ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
                L.cdc_prof_op(h, i, ctypes.byref(lab), ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl))
                f.write(lab.value.decode() + "\n")

    # rows of every rank's shard against batch-1 decodes on rank 0 would need the other ranks' inputs: every rank checks its own
    shard_ok = 1.0
    if world > 1 and not a.no_verify:
        v, _ = verify_rows(decode_fn, init, ctx, rec, B, a.sample_steps)
        shard_ok = 1.0 if v["ok"] else 0.0
    if use_dist:
        t = torch.tensor([shard_ok], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        shard_ok = float(t.item())

    if rank == 0:
        images = B * world * a.steps
        value = images / dt
        scale = (S / 256.0) ** 2
        out = {
            "metric": f"decoded images/sec at {S}x{S}, {a.sample_steps}-step {a.param}-param",
            "value": value, "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype_label(arith), "data": "synthetic",
            "dtype_note": ("float32 tensors and accumulation; convolution products formed on the 16-bit matrix cores from "
                           + ("two-plane fp16 split operands (3 MFMA products per fp32 product; operands carry 22-23 significant "
                              "bits, not 24; |activation| < 65504 or the range guard repeats the call in bf16x3 -- see `range_guard`, "
                              "`alt_arith`)" if arith == 1 else
                              "exact three-plane bf16 split operands (6 MFMA products per fp32 product)")),
            "config": {"workload": workload_name(a.param, B, S, a.sample_steps, world),
                       "batch_per_gpu": B, "global_batch": B * world, "sample_steps": a.sample_steps, "size": S,
                       "parallelism": f"batch-shard x{world}", "finite": ok, "rccl_ranks_seen": ranks_seen,
                       "backend": (a.backend if use_dist else None), "launcher": launcher,
                       "arith": arith_name(arith)},
            "roofline": roofline_block(classes, ops, B, S, arith, value, a.sample_steps, a.steps, dt, cfgd, a.prof_every,
                                       ceilings=measure_ceilings(L, local) if world == 1 else None),
        }
        if world > 1 and not a.no_verify:
            out["verify"] = {"every_rank_checked_rows_0_and_last_of_its_shard_against_batch1_decodes": True,
                             "tolerance": 1e-4, "ok": bool(shard_ok == 1.0)}
            out["config"]["finite"] = ok and out["verify"]["ok"]
        if world == 1 and not a.no_verify:
            # Verification + batch-1 latency (what test_xparam.py runs: one image per call): rows 0 and B-1 of the
            # timed decode are decoded again on their own (different launch plans) and must agree.
            out["verify"], out["batch1"] = verify_rows(decode_fn, init, ctx, rec, B, a.sample_steps, count_ops=lambda: launches_per_iter(L, h))
            out["config"]["finite"] = ok and out["verify"]["ok"]
        if world == 1 and not a.no_alt_arith and arith == 1:
            # The exact-split arithmetic on the record beside the default one: the same decode once more in
            # CDC_ARITH_BF16X3 (three bf16 planes per operand, six MFMA products per fp32 product, full fp32 range).
            _lib.check(h, L.cdc_set_arith(h, 0))
            decode_fn(init, ctx, steps=2)
            torch.cuda.synchronize()
            ta = time.perf_counter()
            rec_alt = decode_fn(init, ctx)
            torch.cuda.synchronize()
            tb = time.perf_counter() - ta
            den = max(1.0, float(rec.abs().max().item()))
            out["alt_arith"] = {"arith": "bf16x3", "dtype": dtype_label(0), "value": B / tb, "unit": "images/s", "ms_per_step": tb * 1e3,
                                "ms_per_ddim_iter": tb / a.sample_steps * 1e3, "peak_tflops": PEAK_16BIT_MFMA_TFLOPS / PRODUCTS[0],
                                "whole_path_tflops_canonical": cfgd["gflop_per_image_step"] * scale * 1e-3 * a.sample_steps * B / tb,
                                "max_rel_diff_vs_f16x2_decode": float((rec_alt - rec).abs().max().item()) / den,
                                "finite": bool(torch.isfinite(rec_alt).all().item()),
                                "note": "one timed decode of the same batch, outside `value`"}
            del rec_alt
            _lib.check(h, L.cdc_set_arith(h, 1))
        if world == 1 and not a.no_extras:
            # What the hipEvent sampling inside the timed region costs: the same short decode with every iteration
            # instrumented and with none; the difference per instrumented iteration x the iterations `value` carried.
            n_s = min(40, a.sample_steps)
            def short(every):
                L.cdc_prof_enable(h, every)
                torch.cuda.synchronize()
                ts = time.perf_counter()
                decode_fn(init, ctx, steps=n_s)
                torch.cuda.synchronize()
                return time.perf_counter() - ts
            short(0)
            t_off = min(short(0), short(0))
            t_on = min(short(1), short(1))
            L.cdc_prof_enable(h, 0)
            L.cdc_prof_reset(h)
            per_iter = max(0.0, (t_on - t_off) / n_s)
            n_inst = len([i for i in range(a.sample_steps) if i % max(2, a.prof_every) == 0]) * a.steps
            out["roofline"]["sampling"] = {"instrumented_iterations": n_inst, "of": a.sample_steps * a.steps,
                                           "ms_per_instrumented_iteration": per_iter * 1e3,
                                           "frac_of_timed_region": per_iter * n_inst / dt,
                                           "note": "hipEvent pairs around every launch of the DDIM iterations i % prof_every == 0, inside the "
                                                   "timed region; cost = (a 40-iteration decode with every iteration instrumented - the same "
                                                   "with none) / 40 x the instrumented iterations of the timed region"}
            out["roofline"]["sampling_cost_frac"] = per_iter * n_inst / dt
        out["range_guard"] = _lib.handle_status(h)
        headline = world == 1 and a.param == "x" and B == 32 and S == 256 and a.sample_steps == 500
        if headline and not a.no_other_configs:
            # BASELINE.json configs[2] and configs[4] on the driver's line: one timed decode each.  The headline
            # model's activations are released first (configs[4] alone holds 14 GB of them).
            del full, rec
            del diff, un
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            out["other_configs"] = [other_config("eps", 32, 256, 1000, local, dev, a.prof_every),
                                    other_config("x", 16, 512, 500, local, dev, a.prof_every)]
            gc.collect()
        if world == 1 and a.param == "x" and S % 64 == 0 and not a.no_extras:
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
            ms_ana, _ = timed(lambda: comp.analysis(img), 3)
            ms_enc, streams = timed(lambda: comp.compress_to_bytes(img), 3)
            ms_ent, ql = timed(lambda: comp.decompress_from_bytes(streams, like=img), 3)
            out["entropy_coder"] = {"encode_ms_per_batch": ms_enc - ms_ana, "decode_ms_per_batch": ms_ent, "analysis_ms_per_batch": ms_ana,
                                    "batch": B, "bytes_per_image": sum(len(x) for x in streams) / B,
                                    "max_abs_diff_vs_forward_q_latent": float((ql - fo["q_latent"]).abs().max().item()),
                                    "note": "64-lane interleaved rANS on the GPU (SURVEY 8f row 4): symbols + hyper_dec + coder + "
                                            "container, host bytes in/out; synthetic parameters, so the sizes say nothing about rate"}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.param, S, a.sample_steps)
        # The driver's record keeps `config` / `roofline` / `cpu_baseline` and drops other top-level keys: what the run measured beside
        # the headline goes into `config` as flat scalars too (the full objects stay at top level for readers of the raw line).
        c = out["config"]
        if "verify" in out:
            c["verify_ok"] = out["verify"].get("ok")
            c["verify_max_rel_err_vs_batch1"] = out["verify"].get("max_rel_err_vs_batch1_decode")
        if "batch1" in out:
            c["batch1_ms_per_ddim_iter"] = out["batch1"]["ms_per_ddim_iter"]
            c["batch1_images_per_s"] = out["batch1"]["images_per_s"]
            c["batch1_launches_per_iter"] = out["batch1"]["launches_per_iter"]
        if "alt_arith" in out:
            c["alt_arith_bf16x3_images_per_s"] = out["alt_arith"]["value"]
            c["alt_arith_bf16x3_ms_per_ddim_iter"] = out["alt_arith"]["ms_per_ddim_iter"]
            c["alt_arith_bf16x3_max_rel_diff"] = out["alt_arith"]["max_rel_diff_vs_f16x2_decode"]
        for tag, oc in zip(("configs2_eps_b32_1000", "configs4_x512_b16_500"), out.get("other_configs", [])):
            c[tag + "_images_per_s"] = oc["value"]
            c[tag + "_ms_per_ddim_iter"] = oc["ms_per_ddim_iter"]
            c[tag + "_roofline_frac"] = oc["roofline"]["frac"]
            c[tag + "_verify_ok"] = oc["verify"]["ok"]
        c["range_faults"] = out["range_guard"].get("range_faults")
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
