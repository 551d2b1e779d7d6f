#!/usr/bin/env python3
"""Development aid: run the attention operator repeatedly; count run-to-run mismatches."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cdc_compression_amd import synth
from cdc_compression_amd.ops import Ops
G = Ops(0)
REP = int(os.environ.get("REP", 150))
for (B, C, H, W) in [(1, 384, 2, 2), (1, 256, 8, 8), (1, 320, 4, 4), (4, 384, 8, 8), (1, 96, 4, 4), (1, 64, 4, 4)]:
    x = synth.normal("ax", (B, C, H, W), 24)
    ng = synth.normal("ag", (1, C, 1, 1), 24, 0.2, 1.0); nb = synth.normal("ab", (1, C, 1, 1), 24, 0.2)
    wq = synth.normal("aq", (3 * C, C, 1, 1), 24, 2.0 / np.sqrt(C)); wo = synth.normal("ao", (C, C, 1, 1), 24, 1.0 / np.sqrt(C))
    bo = synth.normal("aob", (C,), 24, 0.1)
    ref = G.linear_attention(x, ng, nb, wq, wo, bo)
    bad = 0
    for _ in range(REP):
        r = G.linear_attention(x, ng, nb, wq, wo, bo)
        if np.abs(r - ref).max() != 0: bad += 1
    print(f"attention {(B,C,H,W)}: {bad}/{REP} mismatching", flush=True)
