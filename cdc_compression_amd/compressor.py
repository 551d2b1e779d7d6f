"""Host-side mirror of the DECODER half of the reference context model, backed by libcdc_hip.so.

`ResnetCompressor` (xparam/modules/compress_modules.py:110-177) and `BigCompressor`
(epsilonparam/modules/compress_modules.py:112-185) keep the reference constructors' argument names;
`decode(q_latent)` (compress_modules.py:68-74) -- the synthesis transform that turns the transmitted latents
into the context pyramid of the denoising U-Net -- is SURVEY section 8f row 1; `encode(images)` / `forward(images)`
(analysis transform + hyper encoder + quantisers, :43-66, :92-103) are row 3, so that a whole
`GaussianDiffusion.compress()` runs on the GPU with no reference module in the loop.  `load_state_dict` accepts the
reference compressor's full state_dict: `dec.*` is mandatory, `hyper_dec.*` / `prior.*` / `enc.*` / `hyper_enc.*`
are taken when present (each enables the corresponding entry points: `hyper_decode` + `bpp`, `encode` + `forward`).

Decode side of the hyperprior: `hyper_decode(q_hyper_latent)` runs
`hyper_dec` (compress_modules.py:54-59) and returns `(mean, scale.clamp(min=0.1))`; `dequantize(x, offset)`
is `quantize(x, "dequantize", offset)` (utils.py:72-85).
"""
import ctypes

import numpy as np

from . import _lib
from .unet import _Arg, _as_host_f32, _current_stream, _result_like


class NormalDistribution:
    """Holder with the attribute names of utils.py:134-145 (`loc`, `scale`, `mean`)."""

    def __init__(self, loc, scale):
        self.loc, self.scale = loc, scale

    @property
    def mean(self):
        return self.loc


class _ContextDecoder:
    _up_index = 1

    def __init__(self, dim, rev_mults, out_channels, device=0):
        self.dim = dim
        self.rev_mults = tuple(rev_mults)
        self.out_channels = out_channels
        self.reversed_dims = [dim * m for m in self.rev_mults] + [out_channels]
        self.training = False
        self.device_index = int(device) if not hasattr(device, "index") else (device.index or 0)
        self._h = None
        self._sd = {}
        self._finalized = False
        self._hh = None
        self._hyper_finalized = False
        self._prior_loaded = False
        self._medians = None
        self._eh = None
        self._enc_finalized = False
        self.reversed_hyper_dims = None
        self._full_sd = None          # host copy of every entry load_state_dict() used (replayed by .to())

    def status(self):
        """Per library handle (context decoder, hyper decoder, encoder): arithmetic mode and range-guard counters."""
        return {name: _lib.handle_status(getattr(self, attr, None)) for name, attr in (("dec", "_h"), ("hyper_dec", "_hh"), ("enc", "_eh"))}

    @property
    def range_faults(self):
        return sum(v["range_faults"] for v in self.status().values())

    # ---- handle management ----------------------------------------------------------------
    def _handle(self):
        if self._h is None:
            L = _lib.lib()
            cfg = _lib.CtxdecConfig()
            cfg.dim, cfg.out_channels, cfg.up_index = self.dim, self.out_channels, self._up_index
            cfg.n_rev_mults = len(self.rev_mults)
            for i, m in enumerate(self.rev_mults):
                cfg.rev_mults[i] = m
            h = ctypes.c_void_p()
            rc = L.cdc_ctxdec_create(ctypes.byref(cfg), self.device_index, ctypes.byref(h))
            if rc != 0:
                raise _lib.CdcError(f"cdc_ctxdec_create failed ({rc}): {L.cdc_last_error(None).decode()}")
            self._h = h
            for k, v in self._sd.items():
                self._load_one(k, v)
        return self._h

    def __del__(self):
        try:
            if self._h is not None:
                _lib.lib().cdc_destroy(self._h)
                self._h = None
            if self._hh is not None:
                _lib.lib().cdc_destroy(self._hh)
                self._hh = None
            if self._eh is not None:
                _lib.lib().cdc_destroy(self._eh)
                self._eh = None
        except Exception:
            pass

    def to(self, device):
        idx = device if isinstance(device, int) else getattr(device, "index", None)
        if isinstance(device, str):
            idx = int(device.split(":")[1]) if ":" in device else 0
        idx = 0 if idx is None else int(idx)
        if idx != self.device_index:
            # the handles are bound to a device: drop them and replay the parameters on the new one
            for attr in ("_h", "_hh", "_eh"):
                if getattr(self, attr) is not None:
                    _lib.lib().cdc_destroy(getattr(self, attr))
                    setattr(self, attr, None)
            self._finalized = self._hyper_finalized = self._enc_finalized = self._prior_loaded = False
            self.device_index = idx
            if self._full_sd:
                full, self._sd = self._full_sd, {}
                self.load_state_dict(full, strict=False)
        self.device_index = idx
        return self

    def eval(self):
        self.training = False
        return self

    # ---- parameters -----------------------------------------------------------------------
    def manifest(self):
        """[(name, shape)] of the `dec.*` entries, in the reference's registration order."""
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
        """Takes the `dec.*` entries; the encoder / hyperprior entries of a full reference state_dict are
        not this module's.  strict: every `dec.*` key must match the manifest."""
        h = self._handle()
        self._full_sd = {k: _as_host_f32(v).copy() for k, v in state_dict.items()
                         if k.split(".")[0] in ("dec", "hyper_dec", "enc", "hyper_enc", "prior")}
        names = [n for n, _ in self.manifest()]
        missing = [n for n in names if n not in state_dict]
        unexpected = [k for k in state_dict if k.startswith("dec.") and k not in names]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for {type(self).__name__}: missing "
                               f"{missing[:3]}{'...' if len(missing) > 3 else ''}, unexpected {unexpected[:3]}")
        for n in names:
            if n in state_dict:
                self._sd[n] = _as_host_f32(state_dict[n])
                self._load_one(n, self._sd[n])
        _lib.check(h, _lib.lib().cdc_finalize_weights(h))
        self._finalized = True
        if any(k.startswith("hyper_dec.") for k in state_dict):
            self.load_hyper_state_dict(state_dict)      # also takes prior.* when present
        if any(k.startswith("enc.") for k in state_dict):
            self.load_encoder_state_dict(state_dict)
        return self

    def state_dict(self):
        return dict(self._sd)

    # ---- Compressor.decode ------------------------------------------------------------------
    def decode(self, input, cond=None):
        """q_latent [B, reversed_dims[0], h, w] -> [ctx@16h, ctx@8h, ctx@4h, ctx@2h] (finest first)."""
        if cond is not None:
            raise NotImplementedError("vbr conditioning (VBRCondition) is not on the decode path (vbr=False)")
        L, h = _lib.lib(), self._handle()
        if not self._finalized:
            raise _lib.CdcError("load_state_dict() has not been called")
        aq = _Arg(input, self.device_index)
        B, C, hl, wl = aq.shape
        if C != self.reversed_dims[0]:
            raise _lib.CdcError(f"q_latent has {C} channels, the decoder expects {self.reversed_dims[0]}")
        n = len(self.rev_mults)
        outs, ptrs = [], []
        for i in range(n):                     # outs[0] = finest
            lvl = n - 1 - i
            o, p, _ = _result_like(input, (B, self.reversed_dims[lvl + 1], hl << (lvl + 1), wl << (lvl + 1)),
                                   self.device_index)
            outs.append(o)
            ptrs.append(p)
        arr = (ctypes.c_void_p * n)(*ptrs)
        _lib.check(h, L.cdc_ctxdec_decode(h, aq.ptr, arr, n, B, hl, wl, aq.mem, _current_stream(aq.mem)))
        return outs

    # ---- hyperprior, decode side -----------------------------------------------------------
    def _hyper_handle(self):
        if self._hh is None:
            L = _lib.lib()
            cfg = _lib.HyperdecConfig()
            cfg.n_layers = len(self.reversed_hyper_dims) - 1
            for i, d in enumerate(self.reversed_hyper_dims):
                cfg.dims[i] = d
            h = ctypes.c_void_p()
            rc = L.cdc_hyperdec_create(ctypes.byref(cfg), self.device_index, ctypes.byref(h))
            if rc != 0:
                raise _lib.CdcError(f"cdc_hyperdec_create failed ({rc}): {L.cdc_last_error(None).decode()}")
            self._hh = h
        return self._hh

    def hyper_manifest(self):
        L, h = _lib.lib(), self._hyper_handle()
        out = []
        for i in range(L.cdc_num_tensors(h)):
            name = ctypes.c_char_p()
            shape = (ctypes.c_int64 * 4)()
            nd = ctypes.c_int()
            _lib.check(h, L.cdc_tensor_info(h, i, ctypes.byref(name), shape, ctypes.byref(nd)))
            out.append((name.value.decode(), tuple(shape[j] for j in range(nd.value))))
        return out

    def load_hyper_state_dict(self, state_dict):
        """`hyper_dec.*` entries of the reference compressor's state_dict."""
        L, h = _lib.lib(), self._hyper_handle()
        names = [n for n, _ in self.hyper_manifest()]
        missing = [n for n in names if n not in state_dict]
        unexpected = [k for k in state_dict if k.startswith("hyper_dec.") and k not in names]
        if missing or unexpected:
            raise RuntimeError(f"Error(s) in loading state_dict for {type(self).__name__}.hyper_dec: missing "
                               f"{missing[:3]}, unexpected {unexpected[:3]}")
        for n in names:
            a = _as_host_f32(state_dict[n])
            shape = (ctypes.c_int64 * max(a.ndim, 1))(*a.shape)
            _lib.check(h, L.cdc_load_tensor(h, n.encode(), a.ctypes.data, shape, a.ndim))
        # FlexiblePrior (rate estimate only): reference shapes [C,1,1,in,out] / [C,1,1,1,out], singleton axes squeezed
        self._prior_loaded = False
        pk = [k for k in state_dict if k.startswith("prior.affine.") or k.startswith("prior.a.")]
        if pk:
            C = self.reversed_hyper_dims[0]
            for k in pk:
                a = _as_host_f32(state_dict[k])
                a = np.ascontiguousarray(a.reshape((C,) + tuple(d for d in a.shape[3:] if True))
                                         if a.ndim == 5 else a)
                if a.ndim == 3 and not k.endswith(".weight"):
                    a = np.ascontiguousarray(a.reshape(C, -1))
                shape = (ctypes.c_int64 * a.ndim)(*a.shape)
                _lib.check(h, L.cdc_load_tensor(h, k.encode(), a.ctypes.data, shape, a.ndim))
            self._prior_loaded = True
        if "prior._medians" in state_dict:
            self._medians = _as_host_f32(state_dict["prior._medians"]).reshape(1, -1, 1, 1)
        _lib.check(h, L.cdc_finalize_weights(h))
        self._hyper_finalized = True
        return self

    def hyper_decode(self, q_hyper_latent, scale_min=0.1):
        """compress_modules.py:54-59: (mean, scale) of the latent distribution from q_hyper_latent."""
        L, h = _lib.lib(), self._hyper_handle()
        if not self._hyper_finalized:
            raise _lib.CdcError("load_hyper_state_dict() has not been called")
        aq = _Arg(q_hyper_latent, self.device_index)
        B, C, hh, wh = aq.shape
        if C != self.reversed_hyper_dims[0]:
            raise _lib.CdcError(f"q_hyper_latent has {C} channels, hyper_dec expects {self.reversed_hyper_dims[0]}")
        up = 2 ** (len(self.reversed_hyper_dims) - 2)       # one stride-2 ConvTranspose2d per hyper_dec layer but the last
        shape = (B, self.reversed_hyper_dims[-1] // 2, hh * up, wh * up)
        mean, pm, _ = _result_like(q_hyper_latent, shape, self.device_index)
        scale, ps, _ = _result_like(q_hyper_latent, shape, self.device_index)
        _lib.check(h, L.cdc_hyperdec_decode(h, aq.ptr, pm, ps, B, hh, wh, ctypes.c_float(scale_min), aq.mem,
                                            _current_stream(aq.mem)))
        return mean, scale

    def dequantize(self, x, offset):
        """quantize(x, "dequantize", offset) (utils.py:72-85)."""
        L, h = _lib.lib(), self._handle()
        ax, ao = _Arg(x, self.device_index), _Arg(offset, self.device_index)
        if ax.mem != ao.mem or ax.shape != ao.shape:
            raise _lib.CdcError("x and offset must have the same shape and live in the same memory")
        out, po, _ = _result_like(x, ax.shape, self.device_index)
        n = 1
        for d in ax.shape:
            n *= d
        _lib.check(h, L.cdc_dequantize(h, ax.ptr, ao.ptr, po, n, ax.mem, _current_stream(ax.mem)))
        return out

    def rate(self, q_hyper_latent, q_latent, mean, scale, image_hw):
        """bits per pixel of already quantised latents: the two likelihood sums of Compressor.bpp
        (compress_modules.py:84-88) on the GPU."""
        L, h = _lib.lib(), self._hyper_handle()
        if not (self._hyper_finalized and self._prior_loaded):
            raise _lib.CdcError("the prior.* tensors have not been loaded (load_state_dict with the full state_dict)")
        args = [_Arg(t, self.device_index) for t in (q_hyper_latent, q_latent, mean, scale)]
        if len({a.mem for a in args}) != 1:
            raise _lib.CdcError("all inputs must live in the same memory")
        B, _, hh, wh = args[0].shape
        up = 2 ** (len(self.reversed_hyper_dims) - 2)       # as in hyper_decode: one stride-2 ConvTranspose2d per hyper_dec layer but the last
        if args[1].shape != (B, self.reversed_hyper_dims[-1] // 2, up * hh, up * wh) or \
                args[2].shape != args[1].shape or args[3].shape != args[1].shape:
            raise _lib.CdcError(f"latent shapes {args[1].shape} do not belong to a {args[0].shape} hyper latent")
        out, po, _ = _result_like(q_hyper_latent, (B,), self.device_index)
        _lib.check(h, L.cdc_bpp(h, args[0].ptr, args[1].ptr, args[2].ptr, args[3].ptr, po, B, hh, wh,
                                int(image_hw[0]), int(image_hw[1]), args[0].mem, _current_stream(args[0].mem)))
        return out

    def bpp(self, shape, state4bpp):
        """Compressor.bpp (compress_modules.py:76-90) in eval mode: quantise with the medians / the predicted
        mean, then the rate of both latents."""
        B, _, H, W = shape
        dist = state4bpp["latent_distribution"]
        mean, scale = (dist.mean, dist.scale) if hasattr(dist, "mean") else dist
        hyper = state4bpp["hyper_latent"]
        q_hyper = self.dequantize(hyper, self._medians_like(hyper))
        q_latent = self.dequantize(state4bpp["latent"], mean)
        return self.rate(q_hyper, q_latent, mean, scale, (H, W))

    # ---- encoder ---------------------------------------------------
    def _enc_handle(self):
        if self._eh is None:
            L = _lib.lib()
            cfg = _lib.EncoderConfig()
            cfg.dim, cfg.channels, cfg.down_index = self.dim, self.channels, self._up_index
            cfg.n_dim_mults, cfg.n_hyper_mults = len(self.dim_mults), len(self.hyper_dims_mults)
            for i, m in enumerate(self.dim_mults):
                cfg.dim_mults[i] = m
            for i, m in enumerate(self.hyper_dims_mults):
                cfg.hyper_mults[i] = m
            h = ctypes.c_void_p()
            rc = L.cdc_encoder_create(ctypes.byref(cfg), self.device_index, ctypes.byref(h))
            if rc != 0:
                raise _lib.CdcError(f"cdc_encoder_create failed ({rc}): {L.cdc_last_error(None).decode()}")
            self._eh = h
        return self._eh

    def encoder_manifest(self):
        L, h = _lib.lib(), self._enc_handle()
        out = []
        for i in range(L.cdc_num_tensors(h)):
            name = ctypes.c_char_p()
            shape = (ctypes.c_int64 * 4)()
            nd = ctypes.c_int()
            _lib.check(h, L.cdc_tensor_info(h, i, ctypes.byref(name), shape, ctypes.byref(nd)))
            out.append((name.value.decode(), tuple(shape[j] for j in range(nd.value))))
        return out

    def load_encoder_state_dict(self, state_dict):
        """`enc.*` and `hyper_enc.*` entries of the reference compressor's state_dict."""
        L, h = _lib.lib(), self._enc_handle()
        names = [n for n, _ in self.encoder_manifest()]
        missing = [n for n in names if n not in state_dict]
        unexpected = [k for k in state_dict if (k.startswith("enc.") or k.startswith("hyper_enc.")) and k not in names]
        if missing or unexpected:
            raise RuntimeError(f"Error(s) in loading state_dict for {type(self).__name__}.enc: missing "
                               f"{missing[:3]}, unexpected {unexpected[:3]}")
        for n in names:
            a = _as_host_f32(state_dict[n])
            shape = (ctypes.c_int64 * max(a.ndim, 1))(*a.shape)
            _lib.check(h, L.cdc_load_tensor(h, n.encode(), a.ctypes.data, shape, a.ndim))
        _lib.check(h, L.cdc_finalize_weights(h))
        self._enc_finalized = True
        return self

    def analysis(self, images):
        """The unquantised (latent, hyper_latent) of Compressor.encode (compress_modules.py:43-51)."""
        L, h = _lib.lib(), self._enc_handle()
        if not self._enc_finalized:
            raise _lib.CdcError("load_encoder_state_dict() has not been called")
        ax = _Arg(images, self.device_index)
        B, C, H, W = ax.shape
        n, nh = len(self.dim_mults), len(self.hyper_dims_mults)
        lat, pl, _ = _result_like(images, (B, self.dim * self.dim_mults[-1], H >> n, W >> n), self.device_index)
        hyp, ph, _ = _result_like(images, (B, self.dim * self.hyper_dims_mults[-1], H >> (n + nh - 1), W >> (n + nh - 1)),
                                  self.device_index)
        _lib.check(h, L.cdc_encoder_encode(h, ax.ptr, pl, ph, B, H, W, ax.mem, _current_stream(ax.mem)))
        return lat, hyp

    def _medians_like(self, t):
        shape = tuple(_Arg(t, self.device_index).shape)
        med = self._medians if self._medians is not None else np.zeros((1, shape[1], 1, 1), np.float32)
        med = np.broadcast_to(med, shape).copy()
        if type(t).__module__.startswith("torch"):
            import torch
            med = torch.from_numpy(med).to(t.device)
        return med

    def encode(self, input, cond=None):
        """Compressor.encode (compress_modules.py:43-66): (q_latent, q_hyper_latent, state4bpp)."""
        if cond is not None:
            raise NotImplementedError("vbr conditioning is not implemented (vbr=False)")
        latent, hyper_latent = self.analysis(input)
        q_hyper_latent = self.dequantize(hyper_latent, self._medians_like(hyper_latent))
        mean, scale = self.hyper_decode(q_hyper_latent)
        q_latent = self.dequantize(latent, mean)
        state4bpp = {"latent": latent, "hyper_latent": hyper_latent,
                     "latent_distribution": NormalDistribution(mean, scale)}
        return q_latent, q_hyper_latent, state4bpp

    # ---- entropy coder (no reference counterpart) ---------------------
    def _median_vector(self):
        C = self.reversed_hyper_dims[0]
        med = self._medians.reshape(-1) if self._medians is not None else np.zeros(C, np.float32)
        return np.ascontiguousarray(med, dtype=np.float32)

    def compress_to_bytes(self, images):
        """images [B, 3, H, W] -> list of B bitstreams (bytes): analysis transform + hyper encoder on the GPU, then the
        range-ANS coder of include/cdc_hip.h (cdc_entropy_encode) over exactly the symbols `bpp()` prices."""
        latent, hyper = self.analysis(images)
        return self.latents_to_bytes(latent, hyper)

    def latents_to_bytes(self, latent, hyper):
        """The UNquantised outputs of `analysis()` -> list of B bitstreams.  The coder's determinism contract starts here: the
        same (latent, hyper) rows give the same bytes whatever the batch they are coded in (the analysis transform itself is
        an ordinary batched forward: its last bits may depend on the batch size, like any other entry point's)."""
        L, h = _lib.lib(), self._hyper_handle()
        if not (self._hyper_finalized and self._prior_loaded):
            raise _lib.CdcError("the prior.* tensors have not been loaded (load_state_dict with the full state_dict)")
        al, ah = _Arg(latent, self.device_index), _Arg(hyper, self.device_index)
        B, _, hh, wh = ah.shape
        nsym = int(np.prod(al.shape[1:])) + int(np.prod(ah.shape[1:]))
        cap = B * (64 + 2 * 320 + 6 * nsym)              # <= 2 renormalisation bytes + a 4-byte escape payload per symbol
        buf = np.empty(cap, dtype=np.uint8)
        offs = (ctypes.c_size_t * (B + 1))()
        med = self._median_vector()
        _lib.check(h, L.cdc_entropy_encode(h, al.ptr, ah.ptr, med.ctypes.data, B, hh, wh, buf.ctypes.data, cap, offs, al.mem,
                                           _current_stream(al.mem)))
        raw = buf[: offs[B]].tobytes()
        return [raw[offs[b]: offs[b + 1]] for b in range(B)]

    def decompress_from_bytes(self, streams, like=None, return_hyper=False, max_image_hw=None):
        """list of B bitstreams -> q_latent [B, C, h, w] exactly as the encoder dequantised it (numpy, or a tensor on
        `like`'s device); all streams must have the same latent size.  max_image_hw=(H, W): refuse streams whose header
        describes a larger image before anything is allocated (untrusted input; default: the library's 2^22-position bound)."""
        L, h = _lib.lib(), self._hyper_handle()
        # the limit is handle state in the library: set it on EVERY call (None -> the library's default bound), so that one
        # restricted call does not restrict the next; the product is clamped before it is handed over as a C int
        limit = 1 << 22
        if max_image_hw is not None:
            down = 2 ** (len(self.reversed_dims) - 1 + len(self.reversed_hyper_dims) - 2) if hasattr(self, "reversed_dims") else 64
            limit = min(limit, max(1, -(-int(max_image_hw[0]) // down)) * max(1, -(-int(max_image_hw[1]) // down)))
        _lib.check(h, L.cdc_entropy_set_limit(h, int(limit)))
        if not (self._hyper_finalized and self._prior_loaded):
            raise _lib.CdcError("the prior.* tensors have not been loaded (load_state_dict with the full state_dict)")
        streams = [bytes(s) for s in streams]
        B = len(streams)
        hh, wh, ar = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        dims = set()
        for s in streams:
            if L.cdc_entropy_peek(s, len(s), ctypes.byref(hh), ctypes.byref(wh), ctypes.byref(ar)) != 0:
                raise _lib.CdcError("not a CDC bitstream")
            dims.add((hh.value, wh.value))
        if len(dims) != 1:
            raise _lib.CdcError("the streams of one call must share the latent size")
        hh, wh = dims.pop()
        up = 2 ** (len(self.reversed_hyper_dims) - 2)
        C = self.reversed_hyper_dims[-1] // 2
        proto = like if like is not None else np.empty(0, np.float32)
        q, pq, mem = _result_like(proto, (B, C, hh * up, wh * up), self.device_index)
        qh, ph, _ = _result_like(proto, (B, self.reversed_hyper_dims[0], hh, wh), self.device_index)
        blob = b"".join(streams)
        offs = (ctypes.c_size_t * (B + 1))()
        pos = 0
        for b, s in enumerate(streams):
            offs[b] = pos
            pos += len(s)
        offs[B] = pos
        med = self._median_vector()
        _lib.check(h, L.cdc_entropy_decode(h, blob, offs, med.ctypes.data, B, pq, ph, mem, _current_stream(mem)))
        return (q, qh) if return_hyper else q

    def forward(self, input, cond=None):
        """Compressor.forward (compress_modules.py:92-103)."""
        q_latent, q_hyper_latent, state4bpp = self.encode(input, cond)
        shape = tuple(_Arg(input, self.device_index).shape)
        return {"output": self.decode(q_latent), "bpp": self.bpp(shape, state4bpp), "q_latent": q_latent,
                "q_hyper_latent": q_hyper_latent}

    __call__ = forward


class ResnetCompressor(_ContextDecoder):
    """xparam/modules/compress_modules.py:110-177 (decoder half)."""
    _up_index = 1

    def __init__(self, dim=64, dim_mults=(1, 2, 3, 4), reverse_dim_mults=(4, 3, 2, 1),
                 hyper_dims_mults=(4, 4, 4), channels=3, out_channels=3, device=0):
        if dim * dim_mults[-1] != dim * reverse_dim_mults[0]:
            raise AssertionError("dims[-1] == reversed_dims[0]")       # compress_modules.py:23
        super().__init__(dim, reverse_dim_mults, out_channels, device)
        self.dim_mults, self.hyper_dims_mults, self.channels = tuple(dim_mults), tuple(hyper_dims_mults), channels
        # compress_modules.py:26-31
        self.reversed_hyper_dims = list(reversed([dim * dim_mults[-1] * 2] + [dim * m for m in hyper_dims_mults]))


class BigCompressor(_ContextDecoder):
    """epsilonparam/modules/compress_modules.py:112-185 (decoder half, vbr=False)."""
    _up_index = 2

    def __init__(self, dim=64, dim_mults=(1, 3, 3, 3), hyper_dims_mults=(3, 3, 3), channels=3,
                 out_channels=3, vbr=False, device=0):
        if vbr:
            raise NotImplementedError("vbr=True (VBRCondition scalers) is not on the decode path")
        super().__init__(dim, tuple(reversed(dim_mults)), out_channels, device)
        self.dim_mults, self.hyper_dims_mults, self.channels = tuple(dim_mults), tuple(hyper_dims_mults), channels
        self.reversed_hyper_dims = list(reversed([dim * dim_mults[-1] * 2] + [dim * m for m in hyper_dims_mults]))
