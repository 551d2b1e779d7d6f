"""ctypes front-end of oracle/entropy_oracle.c (TEST INFRASTRUCTURE ONLY): the independent CPU restatement of the
entropy coder.  "Parity unpinned": the reference has no entropy coder; the probability
models are the reference's (FlexiblePrior.likelihood, NormalDistribution.likelihood) and are pinned through bpp()."""
import ctypes
import os

import numpy as np

from . import ops as _ops

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        _ops.build()
        L = ctypes.CDLL(os.path.join(_HERE, "libcdc_entropy_oracle.so"))
        vp, i, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
        L.orc_entropy_edges.argtypes = [vp]
        u32 = ctypes.c_uint32
        L.orc_entropy_encode_hyper.argtypes = [vp, i, i, vp, vp, vp, sz, ctypes.POINTER(u32)]
        L.orc_entropy_encode_hyper.restype = sz
        L.orc_entropy_decode_hyper.argtypes = [vp, sz, u32, i, i, vp, vp, vp]
        L.orc_entropy_encode_latent.argtypes = [vp, vp, ctypes.c_longlong, vp, sz, ctypes.POINTER(u32)]
        L.orc_entropy_encode_latent.restype = sz
        L.orc_entropy_decode_latent.argtypes = [vp, sz, u32, vp, ctypes.c_longlong, vp]
        L.orc_entropy_ideal_bits_latent.argtypes = [vp, vp, ctypes.c_longlong]
        L.orc_entropy_ideal_bits_latent.restype = ctypes.c_double
        L.orc_entropy_model_hash.argtypes = [i, vp, vp]
        L.orc_entropy_model_hash.restype = ctypes.c_uint32
        L.orc_entropy_symbol_hash.argtypes = [vp, sz, vp, sz]
        L.orc_entropy_symbol_hash.restype = ctypes.c_uint32
        L.orc_entropy_gauss_table.argtypes = [i, vp, i]
        _lib = L
    return _lib


def raw_prior(state_dict, C):
    """Reference FlexiblePrior parameters -> [C, 44] float32: W0[3] b0[3] a0[3] | W1[9] b1[3] a1[3] | W2[9] b2[3] a2[3] | W3[3] b3."""
    out = np.zeros((C, 44), np.float32)
    pos = 0
    for i in range(4):
        w = np.asarray(state_dict[f"prior.affine.{i}.weight"], np.float32).reshape(C, -1)
        b = np.asarray(state_dict[f"prior.affine.{i}.bias"], np.float32).reshape(C, -1)
        out[:, pos:pos + w.shape[1]] = w; pos += w.shape[1]
        out[:, pos:pos + b.shape[1]] = b; pos += b.shape[1]
        if i < 3:
            a = np.asarray(state_dict[f"prior.a.{i}"], np.float32).reshape(C, -1)
            out[:, pos:pos + a.shape[1]] = a; pos += a.shape[1]
    assert pos == 43
    return out


def edges():
    e = np.zeros(128, np.float32)
    lib().orc_entropy_edges(e.ctypes.data)
    return e


def encode_hyper(sym, prior44, medians):
    """sym [C, h, w] int32 -> (section bytes, number of escape payloads) in the 64-lane interleaved format."""
    sym = np.ascontiguousarray(sym, np.int32)
    C, per = sym.shape[0], int(np.prod(sym.shape[1:]))
    cap = 512 + 8 * sym.size
    out = np.zeros(cap, np.uint8)
    ne = ctypes.c_uint32(0)
    p, m = np.ascontiguousarray(prior44, np.float32), np.ascontiguousarray(medians, np.float32).reshape(-1)
    n = lib().orc_entropy_encode_hyper(sym.ctypes.data, C, per, p.ctypes.data, m.ctypes.data, out.ctypes.data, cap, ctypes.byref(ne))
    assert n > 0
    return out[:n].tobytes(), int(ne.value)


def decode_hyper(data, n_esc, C, per, prior44, medians):
    buf = np.frombuffer(data, np.uint8).copy()
    sym = np.zeros(C * per, np.int32)
    p, m = np.ascontiguousarray(prior44, np.float32), np.ascontiguousarray(medians, np.float32).reshape(-1)
    bad = lib().orc_entropy_decode_hyper(buf.ctypes.data, buf.size, n_esc, C, per, p.ctypes.data, m.ctypes.data, sym.ctypes.data)
    assert not bad
    return sym


def encode_latent(sym, scale):
    sym = np.ascontiguousarray(sym, np.int32).reshape(-1)
    scale = np.ascontiguousarray(scale, np.float32).reshape(-1)
    cap = 512 + 8 * sym.size
    out = np.zeros(cap, np.uint8)
    ne = ctypes.c_uint32(0)
    n = lib().orc_entropy_encode_latent(sym.ctypes.data, scale.ctypes.data, sym.size, out.ctypes.data, cap, ctypes.byref(ne))
    assert n > 0
    return out[:n].tobytes(), int(ne.value)


def decode_latent(data, n_esc, scale, check=True):
    buf = np.frombuffer(data, np.uint8).copy()
    scale = np.ascontiguousarray(scale, np.float32).reshape(-1)
    sym = np.zeros(scale.size, np.int32)
    bad = lib().orc_entropy_decode_latent(buf.ctypes.data, buf.size, n_esc, scale.ctypes.data, scale.size, sym.ctypes.data)
    if check:
        assert not bad
        return sym
    return sym, bool(bad)


def ideal_bits_latent(sym, scale):
    sym = np.ascontiguousarray(sym, np.int32).reshape(-1)
    scale = np.ascontiguousarray(scale, np.float32).reshape(-1)
    return float(lib().orc_entropy_ideal_bits_latent(sym.ctypes.data, scale.ctypes.data, sym.size))


def model_hash(prior44, medians):
    p, m = np.ascontiguousarray(prior44, np.float32), np.ascontiguousarray(medians, np.float32).reshape(-1)
    return int(lib().orc_entropy_model_hash(p.shape[0], p.ctypes.data, m.ctypes.data))


def symbol_hash(sym_h, sym_l):
    a, b = np.ascontiguousarray(sym_h, np.int32).reshape(-1), np.ascontiguousarray(sym_l, np.int32).reshape(-1)
    return int(lib().orc_entropy_symbol_hash(a.ctypes.data, a.size, b.ctypes.data, b.size))


def gauss_table(bin_index):
    """(K, freq[2K+2]) of one scale bin."""
    f = np.zeros(4096, np.uint32)
    K = lib().orc_entropy_gauss_table(bin_index, f.ctypes.data, f.size)
    assert K >= 0
    return K, f[:2 * K + 2].copy()


HEADER = 34


def stream(arith, hh, wh, hyper, latent, model, symbols):
    """The version-3 container of include/cdc_hip.h: 'CDC' 3 | arith | 0 | hh u16 | wh u16 | n_hyper u32 | n_latent u32 |
    model hash u32 | symbol checksum u32 | hyper escapes u32 | latent escapes u32 | hyper section | latent section.
    hyper / latent = (section bytes, escape count) as returned by encode_hyper / encode_latent."""
    import struct
    (hb, he), (lb, le) = hyper, latent
    return b"CDC\x03" + struct.pack("<BBHHIIIIII", arith, 0, hh, wh, len(hb), len(lb), model, symbols, he, le) + hb + lb
