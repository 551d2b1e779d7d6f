"""Host-side mirror of the reference GaussianDiffusion sampler interface (inference half).

  x-param   : xparam/modules/denoising_diffusion.py:12-231   (pred_mode "x", cosine schedule)
  eps-param : epsilonparam/modules/denoising_diffusion.py:12-215 (pred_mode "noise", ddim only;
              the reference's "ddpm" branch is broken: posterior_mean_coef1 is never defined)

`compress()` keeps the reference signature and return value (reconstruction, bpp); the new
`decompress()` is the decode half alone (context pyramid in, reconstruction out).  The N-step
loop runs inside libcdc_hip.so (cdc_decode); eta != 0 falls back to per-step cdc_ddim_step
calls because the reference draws torch.randn_like on the host RNG every step.
"""
import ctypes

import numpy as np

from . import _lib
from .schedule import SampleSchedule
from .unet import _Arg, _current_stream, _is_torch, _result_like


class _GaussianDiffusionBase:
    _param = None   # "x" | "eps"

    def _init_common(self, denoise_fn, context_fn, num_timesteps, pred_mode, var_schedule):
        assert pred_mode in ["noise", "x", "v"]
        self.denoise_fn = denoise_fn
        self.context_fn = context_fn
        self.num_timesteps = int(num_timesteps)
        self.pred_mode = pred_mode
        self.var_schedule = var_schedule
        self.sample_steps = None
        self.training = False
        self._sched = None

    def eval(self):
        self.training = False
        self.denoise_fn.eval()
        return self

    def to(self, device):
        self.denoise_fn.to(device)
        if hasattr(self.context_fn, "to"):
            self.context_fn.to(device)
        return self

    def load_state_dict(self, state_dict, strict=True):
        """Accepts the reference GaussianDiffusion.state_dict(): keys "denoise_fn.*" feed the HIP
        U-Net; "context_fn.*" are forwarded to context_fn if it has load_state_dict; train_* buffers
        are derived constants and ignored."""
        un = {k[len("denoise_fn."):]: v for k, v in state_dict.items() if k.startswith("denoise_fn.")}
        self.denoise_fn.load_state_dict(un, strict=strict)
        cf = {k[len("context_fn."):]: v for k, v in state_dict.items() if k.startswith("context_fn.")}
        if cf and hasattr(self.context_fn, "load_state_dict"):
            self.context_fn.load_state_dict(cf, strict=strict)
        return self

    # ---- schedule ---------------------------------------------------------------------------
    def set_sample_schedule(self, sample_steps, device=None):
        # the schedule flavour (index / time rule, sigma formula) belongs to the TREE, not to pred_mode
        s = SampleSchedule(self.num_timesteps, self.var_schedule, self._param, sample_steps)
        self.sample_steps = sample_steps
        self._sched = s
        self.index = s.index
        self.alphas_cumprod = s.alphas_cumprod
        self.alphas_cumprod_prev = s.alphas_cumprod_prev
        self.sigma = s.sigma
        L, h = _lib.lib(), self.denoise_fn._handle()
        p = lambda a: a.ctypes.data                                   # noqa: E731
        _lib.check(h, L.cdc_set_schedule(h, s.steps, p(s.time_in), p(s.sqrt_recip), p(s.sqrt_recipm1),
                                         p(s.sqrt_ac_prev), p(s.one_minus_ac_prev), p(s.sigma)))
        if self._param == "x" and self.pred_mode == "v":             # predict_start_from_v reads two more tables (x :128-139)
            _lib.check(h, L.cdc_set_schedule_v(h, s.steps, p(s.sqrt_ac), p(s.sqrt_one_minus_ac)))

    # ---- sampler ----------------------------------------------------------------------------
    def _clip_flag(self, clip_denoised):
        if self._param == "x":
            return _lib.CDC_CLIP_ALL if clip_denoised else _lib.CDC_CLIP_NONE
        if clip_denoised == "half":                       # eps :142-143: x_recon[: B // 2].clamp_(-1, 1)
            return _lib.CDC_CLIP_HALF
        return _lib.CDC_CLIP_ALL if clip_denoised == "full" else _lib.CDC_CLIP_NONE

    def _pred_flag(self):
        if self._param == "x":
            return {"x": _lib.CDC_PRED_X, "noise": _lib.CDC_PRED_NOISE_XTREE, "v": _lib.CDC_PRED_V}[self.pred_mode]
        return _lib.CDC_PRED_NOISE                        # (the eps tree's ddim ignores pred_mode: eps :137-139)

    def _loop(self, shape, context, clip_denoised, init, eta):
        L, un = _lib.lib(), self.denoise_fn
        h = un._handle()
        B, C, H, W = shape
        proto = init if init is not None else context[0]
        dev = un.device_index
        actx = [_Arg(c, dev) for c in context]
        mem = actx[0].mem
        if any(c.mem != mem for c in actx):
            raise _lib.CdcError("context tensors must all be host or all be on the model's device")
        ptrs = (ctypes.c_void_p * len(actx))(*[c.ptr for c in actx])
        pred = self._pred_flag()
        clip = self._clip_flag(clip_denoised)
        out, optr, omem = _result_like(proto, (B, C, H, W), dev)
        if omem != mem:
            raise _lib.CdcError("init and context must live in the same memory space")
        stream = _current_stream(mem)
        if eta == 0:
            ai = _Arg(init, dev) if init is not None else None
            if ai is not None and ai.mem != mem:
                raise _lib.CdcError("init and context must live in the same memory space")
            _lib.check(h, L.cdc_decode(h, ai.ptr if ai else None, ptrs, len(actx), optr, B, H, W, pred,
                                       clip, mem, stream))
            return out
        # eta != 0: the reference draws torch.randn_like(noise) per step (x :172, eps :150)
        if init is None:
            img = out
            if _is_torch(img):
                img.zero_()
            else:
                img[...] = 0
        else:
            img = init
        for i in reversed(range(self.sample_steps)):
            if _is_torch(proto):
                import torch
                noise = torch.randn_like(img)
            else:
                noise = np.random.standard_normal(img.shape).astype(np.float32)
            ax, an = _Arg(img, dev), _Arg(noise, dev)
            _lib.check(h, L.cdc_ddim_step(h, ax.ptr, i, ptrs, len(actx), an.ptr, float(eta), optr, B, H,
                                          W, pred, clip, mem, stream))
            if _is_torch(out):
                img = out.clone()
            else:
                img = out.copy()
        return img

    def decompress(self, context, shape, sample_steps=None, init=None, eta=0, clip_denoised=None):
        """Decode half of compress(): context pyramid (= context_fn(...)["output"]) -> image.  `context`
        may also be the transmitted q_latent tensor [B, C, H/16, W/16]: it then goes through
        `context_fn.decode` first (compress_modules.py:68-74; cdc_compression_amd.compressor on the GPU)."""
        if isinstance(context, (bytes, bytearray)):
            context = [context]
        if isinstance(context, (list, tuple)) and context and isinstance(context[0], (bytes, bytearray)):
            # entropy-coded bitstreams (compress_to_bytes): range-ANS decode -> q_latent
            context = self.context_fn.decompress_from_bytes(context, like=init)
        if not isinstance(context, (list, tuple)):
            if self.context_fn is None or not hasattr(self.context_fn, "decode"):
                raise RuntimeError("decompress(q_latent, ...) needs a context_fn with decode()")
            context = self.context_fn.decode(context)
        self.set_sample_schedule(self.num_timesteps if sample_steps is None else sample_steps)
        if clip_denoised is None:
            clip_denoised = True if self._param == "x" else getattr(self, "clip_noise", "none")
        return self._loop(tuple(shape), context, clip_denoised, init, eta)


    def compress_to_bytes(self, images):
        """The transmitted half of compress(): images -> one entropy-coded bitstream per image.
        `decompress(streams, shape, sample_steps, init)` reconstructs from them."""
        return self.context_fn.compress_to_bytes(images)


class GaussianDiffusionX(_GaussianDiffusionBase):
    """xparam/modules/denoising_diffusion.py:12-231."""
    _param = "x"

    def __init__(self, denoise_fn, context_fn, ae_fn=None, num_timesteps=1000, loss_type="l1",
                 lagrangian=1e-3, pred_mode="noise", var_schedule="linear", aux_loss_weight=0,
                 aux_loss_type="l1", use_loss_weight=False, loss_weight_min=5,
                 use_aux_loss_weight_schedule=False):
        if ae_fn is not None:
            raise NotImplementedError("ae_fn (latent diffusion) is not on the tested decode path")
        self._init_common(denoise_fn, context_fn, num_timesteps, pred_mode, var_schedule)
        self.ae_fn = None
        self.loss_type = loss_type
        self.lagrangian_beta = lagrangian

    def p_sample_loop(self, shape, context, clip_denoised=False, init=None, eta=0):
        return self._loop(tuple(shape), context, clip_denoised, init, eta)

    def compress(self, images, sample_steps=None, bpp_return_mean=True, init=None, eta=0):
        context_dict = self.context_fn(images)                                      # :216
        self.set_sample_schedule(self.num_timesteps if sample_steps is None else sample_steps)
        rec = self.p_sample_loop(tuple(images.shape), context_dict["output"], clip_denoised=True,
                                 init=init, eta=eta)                                  # :223
        bpp = context_dict["bpp"]
        return rec, (bpp.mean() if bpp_return_mean else bpp)


class GaussianDiffusionEps(_GaussianDiffusionBase):
    """epsilonparam/modules/denoising_diffusion.py:12-215."""
    _param = "eps"

    def __init__(self, denoise_fn, context_fn, channels=3, num_timesteps=1000, loss_type="l1",
                 clip_noise="half", vbr=False, lagrangian=1e-3, pred_mode="noise", var_schedule="linear",
                 aux_loss_weight=0, aux_loss_type="l1"):
        self._init_common(denoise_fn, context_fn, num_timesteps, pred_mode, var_schedule)
        if pred_mode != "noise":
            raise NotImplementedError('eps-param tree: only pred_mode="noise" reaches ddim()')
        self.channels = channels
        self.clip_noise = clip_noise
        self.vbr = vbr

    def p_sample_loop(self, shape, context, sample_mode, init=None, eta=0):
        if sample_mode != "ddim":
            raise NotImplementedError('sample_mode "ddpm" raises AttributeError in the reference '
                                      "(posterior_mean_coef1 undefined); only \"ddim\" is implemented")
        return self._loop(tuple(shape), context, self.clip_noise, init, eta)

    def compress(self, images, sample_steps=None, bitrate_scale=None, sample_mode="ddpm",
                 bpp_return_mean=True, init=None, eta=0):
        context_dict = self.context_fn(images, bitrate_scale)                        # :205
        self.set_sample_schedule(self.num_timesteps if sample_steps is None else sample_steps)
        rec = self.p_sample_loop(tuple(images.shape), context_dict["output"], sample_mode, init=init,
                                 eta=eta)
        bpp = context_dict["bpp"]
        return rec, (bpp.mean() if bpp_return_mean else bpp)
