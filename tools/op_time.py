#!/usr/bin/env python3
"""Time of one layer's launch program (fused Block: conv + LayerNorm + ReLU + shift + residual), from the stress mode's wall clock:
    op_time.py B Cin H W Cout k stride pad reps      -> us per execution (incl. the two compare kernels of cdc_op_stress, ~4 us)"""
import os
import sys
import time
os.environ.setdefault("CDC_DEV", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from cdc_compression_amd import synth  # noqa: E402
from cdc_compression_amd.ops import Ops  # noqa: E402

B, Ci, H, W, Co, k, s, p, reps = map(int, sys.argv[1:10])
x = synth.normal("dx", (B, Ci, H, W), 31)
w = synth.normal("dw", (Co, Ci, k, k), 31, 1.0 / np.sqrt(Ci * k * k))
b = synth.normal("db", (Co,), 31, 0.1)
g, bb = synth.normal("dg", (Co,), 31, 0.2, 1.0), synth.normal("dbb", (Co,), 31, 0.2)
Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
sh, r = synth.normal("ds", (B, Co), 31, 0.3), synth.normal("dr", (B, Co, Ho, Wo), 31)
G = Ops(0)
ts = []
for n in (reps // 10, reps):            # the difference removes upload / packing / download
    G.stress(n)
    t0 = time.perf_counter()
    y = G.conv2d(x, w, b, s, p, ln_g=g, ln_b=bb, relu=True, shift=sh, resid=r)
    ts.append(time.perf_counter() - t0)
us = (ts[1] - ts[0]) / (reps - reps // 10) * 1e6
env = " ".join(f"{a}={v}" for a, v in sorted(os.environ.items()) if a.startswith("CDC_") and a != "CDC_DEV")
print(f"block {k}x{k} s{s} {Ci}->{Co} @{H}x{W} batch {B}: {us:8.2f} us per execution  finite={bool(np.isfinite(y).all())}  [{env}]", flush=True)
