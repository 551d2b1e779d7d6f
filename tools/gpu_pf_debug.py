#!/usr/bin/env python3
"""Development aid: conv_pf_kernel against a float64 numpy convolution on structured inputs."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cdc_compression_amd.ops import Ops

def ref_conv(x, w, b, pad):
    B, Ci, H, W = x.shape; Co, _, k, _ = w.shape
    xp = np.pad(x.astype(np.float64), ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    out = np.zeros((B, Co, H, W))
    for ky in range(k):
        for kx in range(k):
            out += np.einsum("bchw,oc->bohw", xp[:, :, ky:ky + H, kx:kx + W], w[:, :, ky, kx].astype(np.float64))
    return out + b[None, :, None, None]

G = Ops(0)
rng = np.random.default_rng(0)
for (B, Ci, H, W, Co, k) in [(1, 16, 8, 32, 64, 1), (1, 16, 8, 32, 64, 3), (1, 64, 8, 32, 64, 3), (2, 64, 32, 32, 64, 3),
                              (1, 64, 16, 32, 128, 3), (1, 64, 16, 32, 192, 3), (1, 64, 16, 32, 256, 3)]:
    for mode in ("delta", "rand"):
        if mode == "delta":
            x = np.zeros((B, Ci, H, W), np.float32); x[0, 3, 2, 5] = 1.0; x[0, 9, 4, 20] = 2.0
            w = np.zeros((Co, Ci, k, k), np.float32)
            for co in range(Co):
                w[co, 3, :, :] = (co + 1) + 0.01 * np.arange(k * k).reshape(k, k)
                w[co, 9, :, :] = -(co + 1)
        else:
            x = rng.standard_normal((B, Ci, H, W)).astype(np.float32)
            w = (rng.standard_normal((Co, Ci, k, k)) / np.sqrt(Ci * k * k)).astype(np.float32)
        b = np.zeros(Co, np.float32)
        ref = ref_conv(x, w, b, k // 2)
        got = G.conv2d(x, w, b, 1, k // 2)
        err = np.abs(got - ref).max() / max(1, np.abs(ref).max())
        print(f"B{B} {Ci}->{Co} {H}x{W} k{k} {mode}: relerr {err:.3e}", flush=True)
        if err > 1e-4 and mode == "delta":
            bad = np.argwhere(np.abs(got - ref) > 1e-3 * max(1, np.abs(ref).max()))
            print("  first bad (b,co,y,x):", bad[:8].tolist(), " got", [float(got[tuple(i)]) for i in bad[:4]], " ref", [float(ref[tuple(i)]) for i in bad[:4]])
            nz = np.argwhere(np.abs(got) > 1e-6)
            print("  got nonzero count", len(nz), "ref nonzero count", int((np.abs(ref) > 1e-6).sum()), " got nz sample", nz[:6].tolist())
