#!/usr/bin/env python3
"""Development aid: characterise WHICH elements differ in flaky runs of the attention operator tap."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cdc_compression_amd import synth
from cdc_compression_amd.ops import Ops
G = Ops(0)
B, C, H, W = 4, 384, 8, 8
x = synth.normal("ax", (B, C, H, W), 24)
ng = synth.normal("ag", (1, C, 1, 1), 24, 0.2, 1.0); nb = synth.normal("ab", (1, C, 1, 1), 24, 0.2)
wq = synth.normal("aq", (3 * C, C, 1, 1), 24, 2.0 / np.sqrt(C)); wo = synth.normal("ao", (C, C, 1, 1), 24, 1.0 / np.sqrt(C))
bo = synth.normal("aob", (C,), 24, 0.1)
ref = G.linear_attention(x, ng, nb, wq, wo, bo)
found = 0
for it in range(600):
    r = G.linear_attention(x, ng, nb, wq, wo, bo)
    d = np.abs(r - ref)
    if d.max() != 0:
        idx = np.argwhere(d != 0)
        print(f"run {it}: {len(idx)} differing elements; b {sorted(set(idx[:,0]))} ch range {idx[:,1].min()}..{idx[:,1].max()} "
              f"(n distinct ch {len(set(idx[:,1]))}) rows {sorted(set(idx[:,2]))} cols {sorted(set(idx[:,3]))} maxdiff {d.max():.3e}", flush=True)
        chs = sorted(set(idx[:,1])); print("   channels:", chs[:40])
        found += 1
        if found >= 6: break
print("done", found)
