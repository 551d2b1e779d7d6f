#!/usr/bin/env python3
"""Development aid for a GPU box: runs every parity case WITHOUT stopping at the first failure and
prints one line per case (error, tolerance), so that one gpurun call tells the whole story."""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cdc_compression_amd as cdc  # noqa: E402
from cdc_compression_amd import synth  # noqa: E402
from cdc_compression_amd.ops import Ops  # noqa: E402
from oracle import model as om  # noqa: E402
from oracle import ops as oops  # noqa: E402
import test_gpu_parity as T  # noqa: E402
from helpers import load_case, oracle_cfg  # noqa: E402


def relerr(a, ref):
    return float(np.abs(a - ref).max()) / max(1.0, float(np.abs(ref).max()))


def run(label, fn):
    t0 = time.time()
    try:
        r = fn()
        print(f"[{'ok' if (r is None or r < 1e-4) else 'BAD'}] {label}: {r if r is None else f'{r:.3e}'}  "
              f"({time.time() - t0:.1f}s)", flush=True)
    except Exception as e:
        print(f"[EXC] {label}: {type(e).__name__}: {e}", flush=True)
        traceback.print_exc(limit=3)


def main():
    O, G = oops.OrcOps("f32"), Ops(0)
    for case in T.CONV_CASES:
        B, Ci, H, W, Co, k, s, p, fused = case
        x = synth.normal("cx", (B, Ci, H, W), 21)
        w = synth.normal("cw", (Co, Ci, k, k), 21, 1.0 / np.sqrt(Ci * k * k))
        b = synth.normal("cb", (Co,), 21, 0.1)
        ref = O.conv2d(x, w, b, s, p)
        run(f"conv {case}", lambda: relerr(G.conv2d(x, w, b, s, p), ref))
        if fused:
            g = synth.normal("cg", (Co,), 21, 0.2, 1.0)
            bb = synth.normal("cbb", (Co,), 21, 0.2)
            shift = synth.normal("cs", (B, Co), 21, 0.3)
            resid = synth.normal("cr", ref.shape, 21)
            r2 = np.maximum(O.chan_layernorm(ref, g, bb), 0) + shift[:, :, None, None] + resid
            run(f"conv+ln+relu+shift+resid {case}",
                lambda: relerr(G.conv2d(x, w, b, s, p, ln_g=g, ln_b=bb, relu=True, shift=shift,
                                        resid=resid), r2))
    for case in [(2, 5, 6, 7, 4), (1, 64, 16, 16, 64), (1, 320, 8, 8, 320), (1, 24, 12, 20, 24)]:
        B, Ci, H, W, Co = case
        x = synth.normal("tx", (B, Ci, H, W), 22)
        w = synth.normal("tw", (Ci, Co, 4, 4), 22, 1.0 / np.sqrt(Ci * 4))
        b = synth.normal("tb", (Co,), 22, 0.1)
        ref = O.conv_transpose2d(x, w, b, 2, 1)
        run(f"convT {case}", lambda: relerr(G.conv_transpose2d(x, w, b), ref))
    x = synth.normal("lx", (2, 48, 9, 7), 23, 2.0, 0.5)
    g = synth.normal("lg", (48,), 23, 0.2, 1.0)
    b = synth.normal("lb", (48,), 23, 0.2)
    run("layernorm", lambda: relerr(G.chan_layernorm(x, g, b), O.chan_layernorm(x, g, b)))
    for case in [(2, 16, 8, 8), (1, 64, 32, 32), (2, 128, 16, 16), (1, 384, 8, 8), (1, 24, 12, 20),
                 (1, 64, 64, 64)]:
        B, C, H, W = case
        x = synth.normal("ax", (B, C, H, W), 24)
        sd = {"a.fn.norm.g": synth.normal("ag", (1, C, 1, 1), 24, 0.2, 1.0),
              "a.fn.norm.b": synth.normal("ab", (1, C, 1, 1), 24, 0.2),
              "a.fn.fn.to_qkv.weight": synth.normal("aq", (3 * C, C, 1, 1), 24, 2.0 / np.sqrt(C)),
              "a.fn.fn.to_out.weight": synth.normal("ao", (C, C, 1, 1), 24, 1.0 / np.sqrt(C)),
              "a.fn.fn.to_out.bias": synth.normal("aob", (C,), 24, 0.1)}
        ref = om.attention(O, sd, "a", x)
        run(f"attention {case}", lambda: relerr(G.linear_attention(
            x, sd["a.fn.norm.g"], sd["a.fn.norm.b"], sd["a.fn.fn.to_qkv.weight"],
            sd["a.fn.fn.to_out.weight"], sd["a.fn.fn.to_out.bias"]), ref))
    for name in ["small_x", "small_eps", "odd_x", "full_x", "full_eps"]:
        def f():
            un, kw, sd, x, time_, ctx, g = T.make_unet(name)
            return relerr(un(x, time_, ctx), g["y"])
        run(f"unet {name}", f)


if __name__ == "__main__":
    main()
