#!/usr/bin/env python3
"""Development aid: run single operators repeatedly and report run-to-run mismatches."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cdc_compression_amd import synth
from cdc_compression_amd.ops import Ops
G = Ops(0)
REP = 25
def stress(label, fn):
    ref = fn()
    bad = 0; worst = 0.0
    for _ in range(REP):
        r = fn()
        d = float(np.abs(r - ref).max())
        if d != 0: bad += 1; worst = max(worst, d)
    print(f"{'FLAKY' if bad else 'ok   '} {label}: {bad}/{REP} mismatching runs, worst {worst:.3e}", flush=True)

for (B, Ci, H, W, Co, k, s, p, ln) in [(1, 384, 2, 2, 384, 3, 1, 1, True), (1, 384, 4, 4, 384, 3, 1, 1, True),
        (1, 256, 8, 8, 256, 3, 1, 1, True), (1, 192, 16, 16, 192, 3, 1, 1, True), (1, 128, 32, 32, 128, 3, 1, 1, True),
        (1, 64, 64, 64, 64, 3, 1, 1, True), (1, 67, 64, 64, 64, 7, 1, 3, True), (1, 320, 4, 4, 320, 3, 2, 1, False),
        (1, 64, 64, 64, 192, 1, 1, 0, False), (2, 64, 32, 32, 64, 3, 1, 1, True)]:
    x = synth.normal("cx", (B, Ci, H, W), 21); w = synth.normal("cw", (Co, Ci, k, k), 21, 1.0 / np.sqrt(Ci * k * k))
    b = synth.normal("cb", (Co,), 21, 0.1)
    stress(f"conv {(B,Ci,H,W,Co,k,s,p)}", lambda: G.conv2d(x, w, b, s, p))
    if ln:
        g = synth.normal("cg", (Co,), 21, 0.2, 1.0); bb = synth.normal("cbb", (Co,), 21, 0.2)
        shift = synth.normal("cs", (B, Co), 21, 0.3)
        Ho = (H + 2 * p - k) // s + 1; Wo = (W + 2 * p - k) // s + 1
        resid = synth.normal("cr", (B, Co, Ho, Wo), 21)
        stress(f"conv+ln {(B,Ci,H,W,Co,k)}", lambda: G.conv2d(x, w, b, s, p, ln_g=g, ln_b=bb, relu=True, shift=shift, resid=resid))
for (B, Ci, H, W, Co) in [(1, 384, 2, 2, 384), (1, 64, 32, 32, 64), (1, 320, 4, 4, 320)]:
    x = synth.normal("tx", (B, Ci, H, W), 22); w = synth.normal("tw", (Ci, Co, 4, 4), 22, 1.0 / np.sqrt(Ci * 4)); b = synth.normal("tb", (Co,), 22, 0.1)
    stress(f"convT {(B,Ci,H,W,Co)}", lambda: G.conv_transpose2d(x, w, b))
for (B, C, H, W) in [(1, 384, 2, 2), (1, 320, 4, 4), (1, 256, 8, 8), (1, 192, 16, 16), (1, 128, 32, 32), (1, 64, 64, 64), (2, 16, 8, 8)]:
    x = synth.normal("ax", (B, C, H, W), 24)
    ng = synth.normal("ag", (1, C, 1, 1), 24, 0.2, 1.0); nb = synth.normal("ab", (1, C, 1, 1), 24, 0.2)
    wq = synth.normal("aq", (3 * C, C, 1, 1), 24, 2.0 / np.sqrt(C)); wo = synth.normal("ao", (C, C, 1, 1), 24, 1.0 / np.sqrt(C))
    bo = synth.normal("aob", (C,), 24, 0.1)
    stress(f"attention {(B,C,H,W)}", lambda: G.linear_attention(x, ng, nb, wq, wo, bo))
for (B, C, H, W) in [(1, 384, 2, 2), (1, 320, 4, 4), (2, 48, 9, 7)]:
    x = synth.normal("lx", (B, C, H, W), 23, 2.0, 0.5); g = synth.normal("lg", (C,), 23, 0.2, 1.0); b = synth.normal("lb", (C,), 23, 0.2)
    stress(f"layernorm {(B,C,H,W)}", lambda: G.chan_layernorm(x, g, b))
