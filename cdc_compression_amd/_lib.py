"""ctypes binding of libcdc_hip.so (include/cdc_hip.h).  There is NO CPU fallback: if the HIP
library is missing or cannot be loaded, importing the product path fails loudly."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# CDC_HIP_LIB selects another build of the same HIP library (A/B kernel experiments, tools/build_variant.sh)
LIB_PATH = os.environ.get("CDC_HIP_LIB") or os.path.join(_HERE, "libcdc_hip.so")

CDC_MEM_HOST, CDC_MEM_DEVICE = 0, 1
CDC_PRED_X, CDC_PRED_NOISE, CDC_PRED_NOISE_XTREE, CDC_PRED_V = 0, 1, 2, 3
CDC_CLIP_NONE, CDC_CLIP_ALL, CDC_CLIP_HALF = 0, 1, 2
CDC_MAX_LEVELS = 8

_f = ctypes.POINTER(ctypes.c_float)
_i = ctypes.c_int
_vp = ctypes.c_void_p


class UnetConfig(ctypes.Structure):
    _fields_ = [("dim", ctypes.c_int32), ("channels", ctypes.c_int32),
                ("context_channels", ctypes.c_int32), ("out_dim", ctypes.c_int32),
                ("n_dim_mults", ctypes.c_int32), ("dim_mults", ctypes.c_int32 * CDC_MAX_LEVELS),
                ("n_context_dim_mults", ctypes.c_int32),
                ("context_dim_mults", ctypes.c_int32 * CDC_MAX_LEVELS)]


class CtxdecConfig(ctypes.Structure):
    _fields_ = [("dim", ctypes.c_int32), ("n_rev_mults", ctypes.c_int32),
                ("rev_mults", ctypes.c_int32 * CDC_MAX_LEVELS), ("out_channels", ctypes.c_int32),
                ("up_index", ctypes.c_int32)]


class HyperdecConfig(ctypes.Structure):
    _fields_ = [("n_layers", ctypes.c_int32), ("dims", ctypes.c_int32 * (CDC_MAX_LEVELS + 1))]


class EncoderConfig(ctypes.Structure):
    _fields_ = [("dim", ctypes.c_int32), ("channels", ctypes.c_int32), ("n_dim_mults", ctypes.c_int32),
                ("dim_mults", ctypes.c_int32 * CDC_MAX_LEVELS), ("n_hyper_mults", ctypes.c_int32),
                ("hyper_mults", ctypes.c_int32 * CDC_MAX_LEVELS), ("down_index", ctypes.c_int32)]


class CdcError(RuntimeError):
    pass


def build(force=False):
    """Compile libcdc_hip.so for gfx950 with hipcc (csrc/Makefile)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return LIB_PATH


def kernel_source_hash():
    """sha256[:16] over the HIP sources of the library (csrc/*.hip, *.h in name order): names the build a counter file was
    collected on (profiles/pmc_*_traffic.json carries it; bench.py quotes such a file only for the same kernels)."""
    import hashlib
    d = os.path.join(_HERE, "csrc")
    hsh = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            hsh.update(f.encode())
            hsh.update(open(os.path.join(d, f), "rb").read())
    return hsh.hexdigest()[:16]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CdcError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                       "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    try:
        import torch  # noqa: F401  (share torch's libamdhip64.so.7 so device pointers are compatible)
    except Exception:
        pass
    L = ctypes.CDLL(LIB_PATH)
    H = _vp
    pp = ctypes.POINTER(_vp)
    L.cdc_create.argtypes = [ctypes.POINTER(UnetConfig), _i, ctypes.POINTER(H)]
    L.cdc_destroy.argtypes = [H]
    L.cdc_destroy.restype = None
    L.cdc_last_error.argtypes = [H]
    L.cdc_last_error.restype = ctypes.c_char_p
    L.cdc_version.restype = ctypes.c_char_p
    L.cdc_set_arith.argtypes = [H, _i]
    L.cdc_get_arith.argtypes = [H]
    L.cdc_get_range_faults.argtypes = [H]
    L.cdc_get_nonfinite_results.argtypes = [H]
    L.cdc_num_tensors.argtypes = [H]
    L.cdc_tensor_info.argtypes = [H, _i, ctypes.POINTER(ctypes.c_char_p),
                                  ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(_i)]
    L.cdc_load_tensor.argtypes = [H, ctypes.c_char_p, _vp, ctypes.POINTER(ctypes.c_int64), _i]
    L.cdc_finalize_weights.argtypes = [H]
    L.cdc_unet_forward.argtypes = [H, _vp, _vp, pp, _i, _vp, _i, _i, _i, _i, _vp]
    L.cdc_unet_tap.argtypes = [H, ctypes.c_char_p, _vp, ctypes.POINTER(ctypes.c_int64)]
    L.cdc_ctxdec_create.argtypes = [ctypes.POINTER(CtxdecConfig), _i, ctypes.POINTER(H)]
    L.cdc_ctxdec_decode.argtypes = [H, _vp, pp, _i, _i, _i, _i, _i, _vp]
    L.cdc_encoder_create.argtypes = [ctypes.POINTER(EncoderConfig), _i, ctypes.POINTER(H)]
    L.cdc_encoder_encode.argtypes = [H, _vp, _vp, _vp, _i, _i, _i, _i, _vp]
    L.cdc_hyperdec_create.argtypes = [ctypes.POINTER(HyperdecConfig), _i, ctypes.POINTER(H)]
    L.cdc_hyperdec_decode.argtypes = [H, _vp, _vp, _vp, _i, _i, _i, ctypes.c_float, _i, _vp]
    L.cdc_entropy_encode.argtypes = [H, _vp, _vp, _vp, _i, _i, _i, _vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t), _i, _vp]
    L.cdc_entropy_peek.argtypes = [_vp, ctypes.c_size_t, ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(_i)]
    L.cdc_entropy_set_limit.argtypes = [H, _i]
    L.cdc_entropy_decode.argtypes = [H, _vp, ctypes.POINTER(ctypes.c_size_t), _vp, _i, _vp, _vp, _i, _vp]
    L.cdc_dequantize.argtypes = [H, _vp, _vp, _vp, ctypes.c_longlong, _i, _vp]
    L.cdc_bpp.argtypes = [H, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]
    L.cdc_set_schedule.argtypes = [H, _i, _vp, _vp, _vp, _vp, _vp, _vp]
    L.cdc_set_schedule_v.argtypes = [H, _i, _vp, _vp]
    L.cdc_ddim_step.argtypes = [H, _vp, _i, pp, _i, _vp, ctypes.c_float, _vp, _i, _i, _i, _i, _i,
                                _i, _vp]
    L.cdc_decode.argtypes = [H, _vp, pp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp]
    L.cdc_prof_enable.argtypes = [H, _i]
    L.cdc_prof_name.argtypes = [_i]
    L.cdc_prof_name.restype = ctypes.c_char_p
    L.cdc_prof_get.argtypes = [H, _i, ctypes.POINTER(ctypes.c_double),
                               ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_double),
                               ctypes.POINTER(ctypes.c_double)]
    L.cdc_prof_reset.argtypes = [H]
    L.cdc_prof_num_ops.argtypes = [H]
    L.cdc_prof_op.argtypes = [H, _i, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_double),
                              ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_double)]
    L.cdc_op_conv2d.argtypes = [H, _vp, _vp, _vp, _vp] + [_i] * 9 + [_vp, _vp, _i, _vp, _vp]
    L.cdc_op_conv_transpose2d.argtypes = [H, _vp, _vp, _vp, _vp] + [_i] * 5
    L.cdc_op_chan_layernorm.argtypes = [H, _vp, _vp, _vp, _vp, _i, _i, _i]
    L.cdc_op_linear_attention.argtypes = [H] + [_vp] * 7 + [_i] * 4
    L.cdc_op_stress.argtypes = [H, _i]
    L.cdc_op_stress_result.argtypes = [H, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
    L.cdc_probe_mfma_f16.argtypes = [_i, _i, _i, ctypes.POINTER(ctypes.c_double)]
    L.cdc_probe_hbm_copy.argtypes = [_i, ctypes.c_size_t, _i, ctypes.POINTER(ctypes.c_double)]
    _lib = L
    return L


EXPORTS = ["cdc_create", "cdc_destroy", "cdc_last_error", "cdc_version", "cdc_num_tensors",
           "cdc_tensor_info", "cdc_load_tensor", "cdc_finalize_weights", "cdc_unet_forward",
           "cdc_set_schedule", "cdc_ddim_step", "cdc_decode", "cdc_prof_enable",
           "cdc_prof_num_classes", "cdc_prof_name", "cdc_prof_get", "cdc_prof_reset",
           "cdc_op_conv2d", "cdc_op_conv_transpose2d", "cdc_op_chan_layernorm",
           "cdc_op_linear_attention", "cdc_ctxdec_create", "cdc_ctxdec_decode", "cdc_hyperdec_create",
           "cdc_hyperdec_decode", "cdc_dequantize", "cdc_bpp", "cdc_encoder_create",
           "cdc_encoder_encode", "cdc_set_arith", "cdc_get_arith", "cdc_unet_tap", "cdc_prof_num_ops", "cdc_prof_op",
           "cdc_entropy_encode", "cdc_entropy_peek", "cdc_entropy_set_limit", "cdc_entropy_decode", "cdc_get_range_faults",
           "cdc_get_nonfinite_results", "cdc_set_schedule_v", "cdc_probe_mfma_f16", "cdc_probe_hbm_copy",
           "cdc_op_stress", "cdc_op_stress_result"]


def handle_status(handle):
    """{'arith', 'range_faults', 'nonfinite_results'} of one library handle (None -> zeros): the fp16-range guard of
    include/cdc_hip.h repeats a call in the full-range arithmetic and LEAVES the handle there -- these tell."""
    if handle is None:
        return {"arith": None, "range_faults": 0, "nonfinite_results": 0}
    L = lib()
    return {"arith": int(L.cdc_get_arith(handle)), "range_faults": int(L.cdc_get_range_faults(handle)),
            "nonfinite_results": int(L.cdc_get_nonfinite_results(handle))}


def check(handle, rc):
    if rc != 0:
        msg = lib().cdc_last_error(handle)
        raise CdcError(f"libcdc_hip error {rc}: {msg.decode() if msg else ''}")
