#!/usr/bin/env python3
"""Development aid: repeat forwards/decodes and report error vs golden each time (flakiness hunt)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cdc_compression_amd as cdc
from cdc_compression_amd import synth
import test_gpu_parity as T
from helpers import GOLDEN

def relerr(a, ref):
    return float(np.abs(a - ref).max()) / max(1.0, float(np.abs(ref).max()))

name = sys.argv[1] if len(sys.argv) > 1 else "full_x"
for rep in range(3):
    un, kw, sd, x, time, ctx, g0 = T.make_unet(name)
    errs = [relerr(un(x, time, ctx), g0["y"]) for _ in range(6)]
    print(name, "handle", rep, "fwd errs", " ".join(f"{e:.2e}" for e in errs), flush=True)
    g = np.load(os.path.join(GOLDEN, f"decode_{name}.npz"))
    init = synth.normal("init", x.shape, seed=1, std=0.8)
    diff = cdc.GaussianDiffusionX(un, None, None, num_timesteps=8193, pred_mode="x", var_schedule="cosine")
    errs = [relerr(diff.decompress(ctx, x.shape, sample_steps=3, init=init), g["decode_3"]) for _ in range(4)]
    print(name, "handle", rep, "dec errs", " ".join(f"{e:.2e}" for e in errs), flush=True)
    errs = [relerr(un(x, time, ctx), g0["y"]) for _ in range(3)]
    print(name, "handle", rep, "fwd-after errs", " ".join(f"{e:.2e}" for e in errs), flush=True)
