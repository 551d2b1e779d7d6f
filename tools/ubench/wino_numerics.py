#!/usr/bin/env python3
"""Pricing Winograd F(2x2, 3x3) for the split-fp16 arithmetic, numerics half -- runs on the CPU.

Emulates, in numpy, the arithmetic a Winograd form of the plane-operand kernels would execute for one 3x3 / pad-1 layer:
  direct : a = h + l' 2^-11 (two fp16 planes), w 2^s = WH + WL (fp16), products exact, fp32 accumulation  (what runs today)
  wino   : V = B^T d B in fp32 (adds only), V split into two fp16 planes; U = G g G^T in float64, scaled by a per-layer power of
           two and split into WH + WL; 16 per-position contractions with exact products and fp32 accumulation; Y = A^T M A in fp32
against the float64 convolution.  Inputs: N(0, 1) activations after a LayerNorm + ReLU + shift (the statistics the 3x3 layers see) and
a heavy-tailed variant (a few activations at 1e3 .. 1e4, as in tests/golden/heavy_tail_small_x.npz); weights N(0, 1 / sqrt(9 Cin)).
Prints max |err| / max(1, max |ref|) -- the measure of tests/test_gpu_parity.py::relerr (bound 1e-5 for one layer)."""
import numpy as np

rng = np.random.default_rng(0)
BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)


def split_act(a):
    """a (fp32) -> value of h + l' 2^-11 with h, l' fp16 (conv_kernel.h: split2h)."""
    a = a.astype(np.float32)
    h = a.astype(np.float16)
    l = ((a - h.astype(np.float32)) * np.float32(2048)).astype(np.float16)
    return h.astype(np.float64) + l.astype(np.float64) / 2048.0


def split_w(w):
    """w (float64) -> (value of (WH + WL) 2^-s) as the library packs it: max |w| 2^s in [2^13, 2^14)."""
    wmax = np.abs(w).max()
    s = 14 - np.frexp(wmax)[1]
    v = (w * 2.0 ** s).astype(np.float32)
    wh = v.astype(np.float16)
    wl = (v - wh.astype(np.float32)).astype(np.float16)
    return (wh.astype(np.float64) + wl.astype(np.float64)) * 2.0 ** (-s)


def acc32(prod_terms):
    """fp32 accumulation of exact products along the last axis, in chunks of 16 (one MFMA = an exact-ish 16-term dot added in fp32)."""
    t = prod_terms.reshape(prod_terms.shape[:-1] + (-1, 16)).sum(-1)        # float64 inside an instruction
    acc = np.zeros(t.shape[:-1], np.float32)
    for i in range(t.shape[-1]):
        acc = (acc.astype(np.float64) + t[..., i]).astype(np.float32)
    return acc.astype(np.float64)


def run(Cin, Cout, H, W, heavy):
    x = np.maximum(rng.standard_normal((Cin, H, W)), 0) + 0.3 * rng.standard_normal((Cin, 1, 1))
    if heavy:
        idx = rng.integers(0, x.size, 40)
        x.reshape(-1)[idx] = 10.0 ** rng.uniform(3, 4, 40) * rng.choice([-1, 1], 40)
    x = x.astype(np.float32).astype(np.float64)
    w = rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(9 * Cin)
    w = w.astype(np.float32).astype(np.float64)
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1)))
    # float64 reference and the direct split form
    ref = np.zeros((Cout, H, W))
    terms = []
    xs, wsp = split_act(xp), split_w(w)
    for ky in range(3):
        for kx in range(3):
            ref += np.einsum("oc,chw->ohw", w[:, :, ky, kx], xp[:, ky:ky + H, kx:kx + W])
            terms.append(np.einsum("oc,chw->ohwc", wsp[:, :, ky, kx], xs[:, ky:ky + H, kx:kx + W]))
    direct = acc32(np.concatenate(terms, -1))           # [o,h,w, 9*Cin] in tap-major order
    # Winograd: tiles of 4x4 input, 2x2 output
    th, tw = H // 2, W // 2
    d = np.zeros((Cin, th, tw, 4, 4), np.float32)
    for i in range(4):
        for j in range(4):
            d[..., i, j] = xp[:, i:i + H:2, j:j + W:2][:, :th, :tw]
    # V = B^T d B in fp32: two passes of adds, each rounded to fp32
    t1 = np.zeros_like(d)
    t1[..., 0, :] = d[..., 0, :] - d[..., 2, :]; t1[..., 1, :] = d[..., 1, :] + d[..., 2, :]
    t1[..., 2, :] = d[..., 2, :] - d[..., 1, :]; t1[..., 3, :] = d[..., 1, :] - d[..., 3, :]
    V = np.zeros_like(d)
    V[..., :, 0] = t1[..., :, 0] - t1[..., :, 2]; V[..., :, 1] = t1[..., :, 1] + t1[..., :, 2]
    V[..., :, 2] = t1[..., :, 2] - t1[..., :, 1]; V[..., :, 3] = t1[..., :, 1] - t1[..., :, 3]
    Vs = split_act(V)
    U = np.einsum("ik,ockl,jl->ocij", G, w, G)           # float64 at load time
    Us = np.stack([split_w(U[:, :, i, j]) for i in range(4) for j in range(4)], -1).reshape(Cout, Cin, 4, 4)   # one scale per position
    M = np.zeros((Cout, th, tw, 4, 4), np.float32)
    for i in range(4):
        for j in range(4):
            M[..., i, j] = acc32(np.einsum("oc,chw->ohwc", Us[:, :, i, j], Vs[..., i, j]))
    # Y = A^T M A in fp32
    M = M.astype(np.float32)
    t2 = np.zeros(M.shape[:-2] + (2, 4), np.float32)
    t2[..., 0, :] = (M[..., 0, :] + M[..., 1, :]) + M[..., 2, :]
    t2[..., 1, :] = (M[..., 1, :] - M[..., 2, :]) - M[..., 3, :]
    Y = np.zeros(M.shape[:-2] + (2, 2), np.float32)
    Y[..., :, 0] = (t2[..., :, 0] + t2[..., :, 1]) + t2[..., :, 2]
    Y[..., :, 1] = (t2[..., :, 1] - t2[..., :, 2]) - t2[..., :, 3]
    wino = np.zeros((Cout, H, W))
    for i in range(2):
        for j in range(2):
            wino[:, i::2, j::2] = Y[..., i, j]
    den = max(1.0, np.abs(ref).max())
    return np.abs(direct - ref).max() / den, np.abs(wino - ref).max() / den, np.abs(ref).max()


if __name__ == "__main__":
    print(f"{'layer':<34} {'direct split':>13} {'Winograd split':>15} {'ratio':>7}   max |ref|")
    for (Cin, Cout, S) in ((128, 128, 32), (192, 192, 16), (256, 256, 16), (384, 128, 16)):
        for heavy in (False, True):
            e_d, e_w, m = run(Cin, Cout, S, S, heavy)
            print(f"{Cin:4d}->{Cout:<4d} @{S:<3d} {'heavy tail' if heavy else 'N(0,1) after ReLU':<18} {e_d:13.3e} {e_w:15.3e} {e_w / e_d:7.1f}   {m:.3g}")
