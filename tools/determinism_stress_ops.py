"""Determinism stress of single operators (round 5): the same call again and again must give the same bits."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from cdc_compression_amd.ops import Ops
from cdc_compression_amd import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
G = Ops(0)
def attn(B, C, H, W):
    x = synth.normal("ax", (B, C, H, W), 24)
    args = (x, synth.normal("ag", (1, C, 1, 1), 24, 0.2, 1.0), synth.normal("ab", (1, C, 1, 1), 24, 0.2),
            synth.normal("aq", (3 * C, C, 1, 1), 24, 2.0 / np.sqrt(C)), synth.normal("ao", (C, C, 1, 1), 24, 1.0 / np.sqrt(C)),
            synth.normal("aob", (C,), 24, 0.1))
    return lambda: G.linear_attention(*args)
def conv1(B, Ci, H, W, Co, res):
    x = synth.normal("cx", (B, Ci, H, W), 21)
    w = synth.normal("cw", (Co, Ci, 1, 1), 21, 1.0 / np.sqrt(Ci))
    b = synth.normal("cb", (Co,), 21, 0.1)
    r = synth.normal("cr", (B, Co, H, W), 21) if res else None
    return lambda: G.conv2d(x, w, b, 1, 0, resid=r)
cases = [("attention 32x384x8x8", attn(32, 384, 8, 8)), ("attention 32x320x8x8", attn(32, 320, 8, 8)),
         ("attention 32x320x16x16", attn(32, 320, 16, 16)), ("attention 32x256x16x16", attn(32, 256, 16, 16)),
         ("conv1x1 32x384x8x8 -> 1152", conv1(32, 384, 8, 8, 1152, False)), ("conv1x1 32x384x8x8 -> 384 +res", conv1(32, 384, 8, 8, 384, True)),
         ("conv1x1 32x768x8x8 -> 320", conv1(32, 768, 8, 8, 320, False)), ("conv1x1 32x320x8x8 -> 960", conv1(32, 320, 8, 8, 960, False))]
bad_total = 0
for name, fn in cases:
    ref = fn()
    bad = 0
    worst = 0.0
    t0 = time.time()
    for i in range(N):
        y = fn()
        if not np.array_equal(y, ref):
            bad += 1
            d = np.abs(y - ref)
            worst = max(worst, float(d.max()))
            if bad <= 3:
                idx = np.argwhere(d > 0)
                print("  %s: call %d differs in %d values, first %s, images %s" % (name, i, len(idx), idx[0].tolist(), sorted(set(idx[:, 0].tolist()))[:8]), flush=True)
    bad_total += bad
    print("%s: %d calls, %d differ (worst %.3g; %.0f s)" % (name, N, bad, worst, time.time() - t0), flush=True)
sys.exit(1 if bad_total else 0)
