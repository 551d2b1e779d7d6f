#!/usr/bin/env python3
"""A/B of a forced conv_pf_kernel shape on a plain (bias-only) convolution: result against the default plan's, us per execution from two
stress runs (cdc_op_stress) of different length.   gpu_s2_ab.py B Cin H W Cout k stride pad PLAN   (PLAN = "MB,NPW,WM,WP")"""
import os
import sys
import time
os.environ.setdefault("CDC_DEV", "1")
os.environ.update({"CDC_PF": "1", "CDC_PF_MAXPIX": "0", "CDC_WS_MIN_WGS": "1000000000", "CDC_OP_REQUIRE_PF": "1"})
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from cdc_compression_amd import synth  # noqa: E402
from cdc_compression_amd.ops import Ops  # noqa: E402

B, Ci, H, W, Co, k, s, p = map(int, sys.argv[1:9])
plan = sys.argv[9]
x = synth.normal("dx", (B, Ci, H, W), 31)
w = synth.normal("dw", (Co, Ci, k, k), 31, 1.0 / np.sqrt(Ci * k * k))
b = synth.normal("db", (Co,), 31, 0.1)
G = Ops(0)


def timed(reps):
    ts = []
    for n in (reps // 10, reps):
        G.stress(n)
        t0 = time.perf_counter()
        y = G.conv2d(x, w, b, s, p)
        ts.append(time.perf_counter() - t0)
    G.stress(0)
    return y, (ts[1] - ts[0]) / (reps - reps // 10) * 1e6


for rep in range(2):
    os.environ.pop("CDC_PF_PLAN", None)
    y0, t0 = timed(4000)
    os.environ["CDC_PF_PLAN"] = plan
    y1, t1 = timed(4000)
    print(f"conv {k}x{k} s{s} {Ci}->{Co} @{H}x{W} batch {B}: default plan {t0:8.2f} us, plan {plan} {t1:8.2f} us, max |difference| {np.abs(y1 - y0).max():.3e} "
          f"(max |y| {np.abs(y0).max():.2f})", flush=True)
