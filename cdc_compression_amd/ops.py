"""Single-operator entry points of libcdc_hip.so (the same kernels the U-Net program launches);
numpy in / numpy out, torch.nn.functional semantics.  Used by the parity tests."""
import ctypes

import numpy as np

from . import _lib


def _c(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data


class Ops:
    def __init__(self, device=0):
        L = _lib.lib()
        cfg = _lib.UnetConfig()
        cfg.dim, cfg.channels, cfg.context_channels, cfg.out_dim = 32, 3, 3, 3
        cfg.n_dim_mults = 1
        cfg.dim_mults[0] = 1
        cfg.n_context_dim_mults = 1
        cfg.context_dim_mults[0] = 1
        self._h = ctypes.c_void_p()
        rc = L.cdc_create(ctypes.byref(cfg), device, ctypes.byref(self._h))
        if rc != 0:
            raise _lib.CdcError(f"cdc_create failed ({rc}): {L.cdc_last_error(None).decode()}")

    def __del__(self):
        try:
            _lib.lib().cdc_destroy(self._h)
        except Exception:
            pass

    def status(self):
        """{"arith", "range_faults", "nonfinite_results"} of this handle (range guard of the fp16 arithmetic)."""
        return _lib.handle_status(self._h)

    def stress(self, repeats):
        """Every following operator call launches its program `repeats` more times and counts the executions whose result is not
        bit-identical to the first one (on the device; include/cdc_hip.h: cdc_op_stress).  0 turns it off."""
        _lib.check(self._h, _lib.lib().cdc_op_stress(self._h, int(repeats)))

    def stress_result(self):
        """(executions compared, executions that differed) of the last operator call."""
        n, d = ctypes.c_int64(), ctypes.c_int64()
        _lib.check(self._h, _lib.lib().cdc_op_stress_result(self._h, ctypes.byref(n), ctypes.byref(d)))
        return n.value, d.value

    def prof(self, on=True):
        L = _lib.lib()
        L.cdc_prof_reset(self._h)
        L.cdc_prof_enable(self._h, 1 if on else 0)

    def prof_total_ms(self):
        """(total ms, launches, flops) accumulated since prof(True) over all kernel classes."""
        L = _lib.lib()
        tot, n, fl = 0.0, 0, 0.0
        for c in range(L.cdc_prof_num_classes()):
            ms, k, f, b = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
            L.cdc_prof_get(self._h, c, ctypes.byref(ms), ctypes.byref(k), ctypes.byref(f), ctypes.byref(b))
            tot += ms.value; n += k.value; fl += f.value
        return tot, n, fl

    def conv2d(self, x, w, b=None, stride=1, padding=0, ln_g=None, ln_b=None, relu=False, shift=None,
               resid=None):
        x, w, b, ln_g, ln_b, shift, resid = map(_c, (x, w, b, ln_g, ln_b, shift, resid))
        B, Cin, H, W = x.shape
        Cout, _, KH, KW = w.shape
        Ho = (H + 2 * padding - KH) // stride + 1
        Wo = (W + 2 * padding - KW) // stride + 1
        y = np.empty((B, Cout, Ho, Wo), np.float32)
        if ln_g is not None:
            ln_g, ln_b = ln_g.reshape(-1), ln_b.reshape(-1)
        _lib.check(self._h, _lib.lib().cdc_op_conv2d(
            self._h, _p(x), _p(w), _p(b), _p(y), B, Cin, H, W, Cout, KH, KW, stride, padding, _p(ln_g),
            _p(ln_b), int(relu), _p(shift), _p(resid)))
        return y

    def conv_transpose2d(self, x, w, b=None):
        x, w, b = map(_c, (x, w, b))
        B, Cin, H, W = x.shape
        Cout = w.shape[1]
        y = np.empty((B, Cout, 2 * H, 2 * W), np.float32)
        _lib.check(self._h, _lib.lib().cdc_op_conv_transpose2d(self._h, _p(x), _p(w), _p(b), _p(y), B,
                                                               Cin, H, W, Cout))
        return y

    def chan_layernorm(self, x, g, b):
        x, g, b = _c(x), _c(g).reshape(-1), _c(b).reshape(-1)
        B, C, H, W = x.shape
        y = np.empty_like(x)
        _lib.check(self._h, _lib.lib().cdc_op_chan_layernorm(self._h, _p(x), _p(g), _p(b), _p(y), B, C,
                                                             H * W))
        return y

    def linear_attention(self, x, norm_g, norm_b, w_qkv, w_out, b_out):
        x, norm_g, norm_b, w_qkv, w_out, b_out = map(_c, (x, norm_g, norm_b, w_qkv, w_out, b_out))
        B, C, H, W = x.shape
        y = np.empty_like(x)
        _lib.check(self._h, _lib.lib().cdc_op_linear_attention(
            self._h, _p(x), _p(norm_g.reshape(-1)), _p(norm_b.reshape(-1)), _p(w_qkv), _p(w_out),
            _p(b_out), _p(y), B, C, H, W))
        return y
