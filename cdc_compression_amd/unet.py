"""Host-side mirror of the reference denoising U-Net interface, backed by libcdc_hip.so.

Same constructor arguments, `state_dict` key names, `load_state_dict`, `.to()`, `.eval()` and
`forward(x, time, context)` as `Unet` in xparam/modules/unet.py:18-135 (and
epsilonparam/modules/unet.py:17-124, which lacks `embd_type`), so that test_xparam.py /
test_epsilonparam.py style drivers run unchanged.  All arithmetic happens in hand-written HIP
kernels behind the C-ABI; this file only moves pointers.
"""
import ctypes

import numpy as np

from . import _lib


def _is_torch(t):
    return type(t).__module__.startswith("torch")


def _as_host_f32(t):
    if _is_torch(t):
        t = t.detach().cpu().numpy()
    return np.ascontiguousarray(t, dtype=np.float32)


class _Arg:
    """Uniform view of a numpy / torch-cpu / torch-cuda float32 tensor as (pointer, mem kind)."""

    def __init__(self, t, device_index):
        self.keep = None
        if _is_torch(t) and t.is_cuda:
            import torch
            if t.device.index != device_index:
                raise _lib.CdcError(f"tensor on cuda:{t.device.index}, model on cuda:{device_index}")
            t = t.detach()
            if t.dtype != torch.float32 or not t.is_contiguous():
                t = t.float().contiguous()
            self.keep = t
            self.ptr = t.data_ptr()
            self.mem = _lib.CDC_MEM_DEVICE
            self.shape = tuple(t.shape)
        else:
            a = _as_host_f32(t)
            self.keep = a
            self.ptr = a.ctypes.data
            self.mem = _lib.CDC_MEM_HOST
            self.shape = a.shape


def _result_like(proto, shape, device_index):
    """Allocate the output in the same container family as `proto`."""
    if _is_torch(proto):
        import torch
        if proto.is_cuda:
            t = torch.empty(shape, dtype=torch.float32, device=proto.device)
            return t, t.data_ptr(), _lib.CDC_MEM_DEVICE
        t = torch.empty(shape, dtype=torch.float32)
        return t, t.data_ptr(), _lib.CDC_MEM_HOST
    a = np.empty(shape, np.float32)
    return a, a.ctypes.data, _lib.CDC_MEM_HOST


def _current_stream(mem):
    if mem == _lib.CDC_MEM_DEVICE:
        import torch
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    return None


class Unet:
    def __init__(self, dim, out_dim=None, dim_mults=(1, 2, 4, 8), context_dim_mults=(1, 2, 3, 3),
                 channels=3, context_channels=3, with_time_emb=True, embd_type="01", device=0):
        if not with_time_emb:
            raise NotImplementedError("with_time_emb=False is not on the decode path")
        if embd_type != "01":
            raise NotImplementedError('only embd_type="01" (unet.py:39-41) is implemented')
        self.dim = dim
        self.channels = channels
        self.context_channels = context_channels
        self.out_dim = channels if out_dim is None else out_dim
        self.dim_mults = tuple(dim_mults)
        self.context_dim_mults = tuple(context_dim_mults)
        self.embd_type = embd_type
        self.training = False
        self.device_index = int(device) if not hasattr(device, "index") else (device.index or 0)
        self._h = None
        self._sd = {}
        self._finalized = False
        self._want_final = False     # load_state_dict() completed once: re-finalize after a device change

    # ---- handle management ----------------------------------------------------------------
    def _handle(self):
        if self._h is None:
            L = _lib.lib()
            cfg = _lib.UnetConfig()
            cfg.dim, cfg.channels, cfg.context_channels = self.dim, self.channels, self.context_channels
            cfg.out_dim = self.out_dim
            cfg.n_dim_mults = len(self.dim_mults)
            cfg.n_context_dim_mults = len(self.context_dim_mults)
            for i, m in enumerate(self.dim_mults):
                cfg.dim_mults[i] = m
            for i, m in enumerate(self.context_dim_mults):
                cfg.context_dim_mults[i] = m
            h = ctypes.c_void_p()
            rc = L.cdc_create(ctypes.byref(cfg), self.device_index, ctypes.byref(h))
            if rc != 0:
                raise _lib.CdcError(f"cdc_create failed ({rc}): {L.cdc_last_error(None).decode()}")
            self._h = h
            for k, v in self._sd.items():
                self._load_one(k, v)
            if self._sd and self._want_final:
                # a handle re-created after .to(other device): the replayed parameters are final again
                _lib.check(h, L.cdc_finalize_weights(h))
                self._finalized = True
        return self._h

    def status(self):
        """Arithmetic mode and range-guard counters of the library handle (see _lib.handle_status)."""
        return _lib.handle_status(self._h)

    @property
    def range_faults(self):
        return self.status()["range_faults"]

    def __del__(self):
        try:
            if self._h is not None:
                _lib.lib().cdc_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def to(self, device):
        idx = device if isinstance(device, int) else getattr(device, "index", None)
        if isinstance(device, str):
            idx = int(device.split(":")[1]) if ":" in device else 0
        idx = 0 if idx is None else int(idx)
        if idx != self.device_index and self._h is not None:
            _lib.lib().cdc_destroy(self._h)
            self._h = None
            self._finalized = False
        self.device_index = idx
        return self

    def eval(self):
        self.training = False
        return self

    # ---- parameters -----------------------------------------------------------------------
    def manifest(self):
        """[(name, shape)] in the order of the reference Unet.state_dict()."""
        L, h = _lib.lib(), self._handle()
        out = []
        for i in range(L.cdc_num_tensors(h)):
            name = ctypes.c_char_p()
            shape = (ctypes.c_int64 * 4)()
            nd = ctypes.c_int()
            _lib.check(h, L.cdc_tensor_info(h, i, ctypes.byref(name), shape, ctypes.byref(nd)))
            out.append((name.value.decode(), tuple(shape[j] for j in range(nd.value))))
        return out

    def _load_one(self, name, value):
        L, h = _lib.lib(), self._h
        a = _as_host_f32(value)
        shape = (ctypes.c_int64 * max(a.ndim, 1))(*a.shape)
        _lib.check(h, L.cdc_load_tensor(h, name.encode(), a.ctypes.data, shape, a.ndim))

    def load_state_dict(self, state_dict, strict=True):
        h = self._handle()
        names = [n for n, _ in self.manifest()]
        missing = [n for n in names if n not in state_dict]
        unexpected = [k for k in state_dict if k not in names]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for Unet: missing {missing[:3]}"
                               f"{'...' if len(missing) > 3 else ''}, unexpected {unexpected[:3]}")
        for n in names:
            if n in state_dict:
                self._sd[n] = _as_host_f32(state_dict[n])
                self._load_one(n, self._sd[n])
        _lib.check(h, _lib.lib().cdc_finalize_weights(h))
        self._finalized = True
        self._want_final = True
        return self

    def state_dict(self):
        return dict(self._sd)

    # ---- forward --------------------------------------------------------------------------
    def forward(self, x, time=None, context=None):
        """Unet.forward (unet.py:131-135): x [B,C,H,W], time [B,1] or [B], context list."""
        L, h = _lib.lib(), self._handle()
        if not self._finalized:
            raise _lib.CdcError("load_state_dict() has not been called")
        if time is None or context is None:
            raise _lib.CdcError("time and context are required on the decode path")
        ax = _Arg(x, self.device_index)
        B, C, H, W = ax.shape
        mem = ax.mem
        if mem == _lib.CDC_MEM_DEVICE:
            at = _Arg(time.reshape(-1) if _is_torch(time) else time, self.device_index)
            actx = [_Arg(c, self.device_index) for c in context]
            if at.mem != mem or any(c.mem != mem for c in actx):
                raise _lib.CdcError("x, time and context must live on the same device")
        else:
            at = _Arg(_as_host_f32(time).reshape(-1), self.device_index)
            actx = [_Arg(_as_host_f32(c), self.device_index) for c in context]
        ptrs = (ctypes.c_void_p * max(len(actx), 1))(*[c.ptr for c in actx])
        out, optr, _ = _result_like(x, (B, self.out_dim, H, W), self.device_index)
        _lib.check(h, L.cdc_unet_forward(h, ax.ptr, at.ptr, ptrs, len(actx), optr, B, H, W, mem,
                                         _current_stream(mem)))
        return out

    __call__ = forward

    def tap(self, name):
        """Intermediate activation of the last forward / DDIM iteration by the reference's module path
        ("downs.0.0", "downs.1.3", "mid_block1", "ups.0", ...): what a forward hook on that module records."""
        import numpy as np
        L, h = _lib.lib(), self._handle()
        shape = (ctypes.c_int64 * 4)()
        _lib.check(h, L.cdc_unet_tap(h, name.encode(), None, shape))
        out = np.empty(tuple(shape), np.float32)
        _lib.check(h, L.cdc_unet_tap(h, name.encode(), out.ctypes.data, shape))
        return out
