"""ctypes front-end of oracle/cdc_oracle.c (TEST INFRASTRUCTURE ONLY).

NCHW float32 numpy in / out, torch.nn.functional semantics.  `OrcOps(acc="f32")` loads the
float-accumulating build (the CPU baseline); `acc="f64"` loads the double-accumulating build
(tight checker).  `NumpyOps` is an independent pure-numpy restatement of the same primitives,
used in tests/ to cross-check the C code on small shapes.
"""
import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_f = ctypes.POINTER(ctypes.c_float)
_i = ctypes.c_int


def build(force=False):
    """Compile the C oracle with gcc (oracle/Makefile). Building the checker is not using it."""
    libs = [os.path.join(_HERE, n) for n in ("libcdc_oracle.so", "libcdc_oracle64.so", "libcdc_entropy_oracle.so")]
    if force or not all(os.path.exists(p) for p in libs):
        subprocess.check_call(["make", "-C", _HERE, "-B"] if force else ["make", "-C", _HERE],
                              stdout=subprocess.DEVNULL)
    return libs


def usable_cores():
    """CPUs this process may really use: affinity mask and cgroup-v2 quota (a container that sees 256 CPUs with a 16-CPU quota
    must not run 256 OpenMP threads: every barrier then waits for throttled threads)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def _ptr(a):
    return a.ctypes.data_as(_f) if a is not None else None


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class OrcOps:
    def __init__(self, acc="f32"):
        build()
        name = "libcdc_oracle.so" if acc == "f32" else "libcdc_oracle64.so"
        self.lib = ctypes.CDLL(os.path.join(_HERE, name))
        L = self.lib
        L.orc_conv2d.argtypes = [_f, _f, _f, _f] + [_i] * 9
        L.orc_conv_transpose2d.argtypes = [_f, _f, _f, _f] + [_i] * 10
        L.orc_chan_layernorm.argtypes = [_f, _f, _f, _f, _i, _i, _i, ctypes.c_float]
        L.orc_linear_attention_core.argtypes = [_f, _f, _i, _i, _i, ctypes.c_float]
        assert L.orc_acc_bytes() == (4 if acc == "f32" else 8)
        L.orc_set_threads(usable_cores())
        self.threads = int(L.orc_get_threads())

    def conv2d(self, x, w, b=None, stride=1, padding=0):
        x, w = _c(x), _c(w)
        b = _c(b) if b is not None else None
        B, Cin, H, W = x.shape
        Cout, Cin2, KH, KW = w.shape
        assert Cin == Cin2, (x.shape, w.shape)
        Ho = (H + 2 * padding - KH) // stride + 1
        Wo = (W + 2 * padding - KW) // stride + 1
        y = np.empty((B, Cout, Ho, Wo), np.float32)
        self.lib.orc_conv2d(_ptr(x), _ptr(w), _ptr(b), _ptr(y), B, Cin, H, W, Cout, KH, KW,
                            stride, padding)
        return y

    def conv_transpose2d(self, x, w, b=None, stride=1, padding=0, output_padding=0):
        x, w = _c(x), _c(w)
        b = _c(b) if b is not None else None
        B, Cin, H, W = x.shape
        Cin2, Cout, KH, KW = w.shape
        assert Cin == Cin2
        Ho = (H - 1) * stride - 2 * padding + KH + output_padding
        Wo = (W - 1) * stride - 2 * padding + KW + output_padding
        y = np.empty((B, Cout, Ho, Wo), np.float32)
        self.lib.orc_conv_transpose2d(_ptr(x), _ptr(w), _ptr(b), _ptr(y), B, Cin, H, W, Cout,
                                      KH, KW, stride, padding, output_padding)
        return y

    def chan_layernorm(self, x, g, b, eps=1e-5):
        x = _c(x)
        g, b = _c(g).reshape(-1), _c(b).reshape(-1)
        B, C, H, W = x.shape
        y = np.empty_like(x)
        self.lib.orc_chan_layernorm(_ptr(x), _ptr(g), _ptr(b), _ptr(y), B, C, H * W, eps)
        return y

    def linear_attention_core(self, qkv, scale):
        qkv = _c(qkv)
        B, C3, H, W = qkv.shape
        C = C3 // 3
        out = np.empty((B, C, H, W), np.float32)
        self.lib.orc_linear_attention_core(_ptr(qkv), _ptr(out), B, C, H * W, scale)
        return out


class NumpyOps:
    """Independent slow restatement (numpy, float64 accumulate) for cross-checking OrcOps."""

    def conv2d(self, x, w, b=None, stride=1, padding=0):
        x = np.asarray(x, np.float64)
        w = np.asarray(w, np.float64)
        B, Cin, H, W = x.shape
        Cout, _, KH, KW = w.shape
        xp = np.pad(x, ((0, 0), (0, 0), (padding, padding), (padding, padding)))
        Ho = (H + 2 * padding - KH) // stride + 1
        Wo = (W + 2 * padding - KW) // stride + 1
        y = np.zeros((B, Cout, Ho, Wo))
        for ky in range(KH):
            for kx in range(KW):
                patch = xp[:, :, ky:ky + (Ho - 1) * stride + 1:stride,
                           kx:kx + (Wo - 1) * stride + 1:stride]
                y += np.einsum("bchw,oc->bohw", patch, w[:, :, ky, kx])
        if b is not None:
            y += np.asarray(b, np.float64)[None, :, None, None]
        return y.astype(np.float32)

    def conv_transpose2d(self, x, w, b=None, stride=1, padding=0, output_padding=0):
        x = np.asarray(x, np.float64)
        w = np.asarray(w, np.float64)
        B, Cin, H, W = x.shape
        _, Cout, KH, KW = w.shape
        Hf = (H - 1) * stride + KH + output_padding
        Wf = (W - 1) * stride + KW + output_padding
        full = np.zeros((B, Cout, Hf, Wf))
        for ky in range(KH):
            for kx in range(KW):
                full[:, :, ky:ky + (H - 1) * stride + 1:stride,
                     kx:kx + (W - 1) * stride + 1:stride] += np.einsum(
                         "bchw,co->bohw", x, w[:, :, ky, kx])
        y = full[:, :, padding:Hf - padding, padding:Wf - padding]
        if b is not None:
            y = y + np.asarray(b, np.float64)[None, :, None, None]
        return np.ascontiguousarray(y).astype(np.float32)

    def chan_layernorm(self, x, g, b, eps=1e-5):
        x = np.asarray(x, np.float64)
        mean = x.mean(1, keepdims=True)
        var = x.var(1, keepdims=True)
        y = (x - mean) / np.sqrt(var + eps) * np.asarray(g, np.float64).reshape(1, -1, 1, 1) \
            + np.asarray(b, np.float64).reshape(1, -1, 1, 1)
        return y.astype(np.float32)

    def linear_attention_core(self, qkv, scale):
        qkv = np.asarray(qkv, np.float64)
        B, C3, H, W = qkv.shape
        C = C3 // 3
        q, k, v = [t.reshape(B, C, H * W) for t in np.split(qkv, 3, axis=1)]
        q = q * scale
        k = np.exp(k - k.max(-1, keepdims=True))
        k = k / k.sum(-1, keepdims=True)
        ctx = np.einsum("bdn,ben->bde", k, v)
        out = np.einsum("bde,bdn->ben", ctx, q)
        return out.reshape(B, C, H, W).astype(np.float32)


def gelu_erf(x):
    """nn.GELU() default (exact erf form), reference unet.py:41."""
    x = np.asarray(x, np.float32)
    erf = np.vectorize(math.erf, otypes=[np.float64])
    return (0.5 * x.astype(np.float64) * (1.0 + erf(x.astype(np.float64) / math.sqrt(2.0)))
            ).astype(np.float32)
