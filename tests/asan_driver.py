"""Runs inside a subprocess with the AddressSanitizer build of the C-ABI preloaded (tests/test_host_logic.py::
test_c_abi_host_paths_under_address_sanitizer): life cycle and error paths of every handle kind, on a host without a GPU too.
Prints ASAN_DRIVER_OK when every call behaved; AddressSanitizer aborts the process on a finding."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cdc_compression_amd as cdc  # noqa: E402
from cdc_compression_amd import _lib, synth  # noqa: E402

L = _lib.lib()
assert "asan" in _lib.LIB_PATH, _lib.LIB_PATH


def expect_error(fn):
    try:
        fn()
    except (_lib.CdcError, ValueError, KeyError, RuntimeError) as e:
        return str(e)
    raise AssertionError("expected an error")


# 1. U-Net handle: manifest walk, strict loading, wrong shapes / unknown names, finalize (fails loudly without a GPU), destroy
un = cdc.Unet(dim=16, channels=3, context_channels=8, dim_mults=(1, 2, 3), context_dim_mults=(1, 2))
man = un.manifest()
assert len(man) > 50
sd = synth.unet_state_dict(man, seed=3)
bad = dict(sd)
k0 = man[0][0]
bad[k0] = np.zeros((1,) + tuple(man[0][1]), np.float32)
expect_error(lambda: un.load_state_dict(bad))                     # wrong shape
missing = dict(sd)
del missing[k0]
expect_error(lambda: un.load_state_dict(missing))                 # strict: a missing key
extra = dict(sd)
extra["not.a.parameter"] = np.zeros(3, np.float32)
expect_error(lambda: un.load_state_dict(extra))                   # strict: an unknown key
has_gpu = False
try:
    un.load_state_dict(sd)
    has_gpu = True
except _lib.CdcError as e:
    assert "no HIP device" in str(e), e
h = un._handle()
# raw C-ABI error paths on the same handle
shape = (ctypes.c_int64 * 4)(1, 2, 3, 4)
buf = np.zeros(24, np.float32)
assert L.cdc_load_tensor(h, b"no.such.tensor", buf.ctypes.data, shape, 4) < 0
assert L.cdc_last_error(h)
assert L.cdc_load_tensor(h, k0.encode(), buf.ctypes.data, shape, 9) < 0                      # absurd rank
assert L.cdc_set_schedule(h, 0, None, None, None, None, None, None) < 0                      # zero steps / null tables
assert L.cdc_op_stress(h, -1) < 0
n, d = ctypes.c_int64(), ctypes.c_int64()
assert L.cdc_op_stress_result(h, ctypes.byref(n), ctypes.byref(d)) == 0
assert L.cdc_prof_enable(h, 1) == 0 and L.cdc_prof_reset(h) == 0 and L.cdc_prof_num_ops(h) >= 0
name, ndim = ctypes.c_char_p(), ctypes.c_int()
shp = (ctypes.c_int64 * 8)()
assert L.cdc_tensor_info(h, 10 ** 6, ctypes.byref(name), shp, ctypes.byref(ndim)) < 0        # index out of range
if not has_gpu:
    x = np.zeros((1, 3, 32, 32), np.float32)
    expect_error(lambda: un.forward(x, np.zeros((1,), np.float32), [np.zeros((1, 8, 32, 32), np.float32), np.zeros((1, 16, 16, 16), np.float32)]))
del un

# 2. the single-operator entry points (no GPU: every one must fail loudly, not crash)
from cdc_compression_amd.ops import Ops  # noqa: E402
G = Ops()
if not has_gpu:
    expect_error(lambda: G.chan_layernorm(np.zeros((1, 4, 2, 2), np.float32), np.ones(4), np.zeros(4)))
    expect_error(lambda: G.conv2d(np.zeros((1, 4, 8, 8), np.float32), np.zeros((4, 4, 3, 3), np.float32), None, 1, 1))
del G

# 3. compressor handles (context decoder, hyper decoder, encoder): create, manifests, destroy
comp = cdc.ResnetCompressor(dim=8, dim_mults=[1, 2, 3, 4], reverse_dim_mults=[4, 3, 2, 1], hyper_dims_mults=[4, 4, 4], channels=3, out_channels=8)
for m in (comp.manifest(), comp.hyper_manifest(), comp.encoder_manifest()):
    assert len(m) > 3
csd = synth.unet_state_dict(comp.manifest() + comp.hyper_manifest() + comp.encoder_manifest(), seed=5)
try:
    comp.load_state_dict(csd)
except (_lib.CdcError, KeyError) as e:
    assert has_gpu or "no HIP device" in str(e) or "prior" in str(e), e
del comp

# 4. entropy container parser on hostile bytes (pure host code)
hh, wh, b = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
for blob in (b"", b"\x00" * 10, b"CDC" + b"\xff" * 40, os.urandom(64)):
    arr = (ctypes.c_ubyte * max(1, len(blob))).from_buffer_copy(blob or b"\x00")
    rc = L.cdc_entropy_peek(arr, len(blob), ctypes.byref(hh), ctypes.byref(wh), ctypes.byref(b))
    assert rc != 0 or (hh.value >= 0 and wh.value >= 0)

print("ASAN_DRIVER_OK", "gpu" if has_gpu else "no-gpu")
