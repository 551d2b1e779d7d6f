"""Determinism stress of one attention level (round 5): the same call again and again must give the same bits."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cdc_compression_amd.ops import Ops
from cdc_compression_amd import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
B, C, H, W = [int(v) for v in (sys.argv[2:6] if len(sys.argv) > 5 else (32, 192, 64, 64))]
G = Ops(0)
x = synth.normal("ax", (B, C, H, W), 24)
args = (x, synth.normal("ag", (1, C, 1, 1), 24, 0.2, 1.0), synth.normal("ab", (1, C, 1, 1), 24, 0.2),
        synth.normal("aq", (3 * C, C, 1, 1), 24, 2.0 / np.sqrt(C)), synth.normal("ao", (C, C, 1, 1), 24, 1.0 / np.sqrt(C)),
        synth.normal("aob", (C,), 24, 0.1))
ref = G.linear_attention(*args)
bad, t0 = 0, time.time()
for i in range(N):
    y = G.linear_attention(*args)
    if not np.array_equal(y, ref):
        bad += 1
        d = np.abs(y - ref)
        idx = np.argwhere(d > 0)
        print("call %d differs: %d values, images %s, max %.3g" % (i, len(idx), sorted(set(idx[:, 0].tolist())), float(d.max())), flush=True)
print("attention %dx%dx%dx%d: %d calls, %d differ (%.0f s)" % (B, C, H, W, N, bad, time.time() - t0), flush=True)
