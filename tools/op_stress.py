#!/usr/bin/env python3
"""One layer through the C-ABI's determinism stress (cdc_op_stress): N executions on the device, each compared bitwise with the first.
    op_stress.py B Cin H W Cout k stride pad repeats        (development switches from the environment, CDC_DEV=1)
prints: label, executions, executions that differ."""
import os
import sys
import time

os.environ.setdefault("CDC_DEV", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from cdc_compression_amd import synth  # noqa: E402
from cdc_compression_amd.ops import Ops  # noqa: E402


def main():
    B, Ci, H, W, Co, k, s, p, reps = map(int, sys.argv[1:10])
    G = Ops(0)
    G.stress(reps)
    x = synth.normal("dx", (B, Ci, H, W), 31)
    w = synth.normal("dw", (Co, Ci, k, k), 31, 1.0 / np.sqrt(Ci * k * k))
    b = synth.normal("db", (Co,), 31, 0.1)
    t0 = time.perf_counter()
    G.conv2d(x, w, b, s, p)
    dt = time.perf_counter() - t0
    n, d = G.stress_result()
    env = " ".join(f"{a}={v}" for a, v in sorted(os.environ.items()) if a.startswith("CDC_") and a not in ("CDC_DEV",))
    print(f"conv {k}x{k} s{s} {Ci}->{Co} @{H}x{W} batch {B}: {n} executions, {d} differ from the first   ({dt:.1f} s)  [{env}]", flush=True)


if __name__ == "__main__":
    main()
