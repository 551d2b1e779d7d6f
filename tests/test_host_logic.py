"""CPU tests (no GPU): the C-ABI library loads and exports every declared symbol, the host-side
mirrors reproduce the reference's state_dict manifest and sampling schedules, and compute entry
points fail loudly (no CPU fallback) when no GPU is present."""
import json
import os
import re

import numpy as np
import pytest

import cdc_compression_amd as cdc
from cdc_compression_amd import _lib, schedule
from helpers import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "cdc_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(cdc_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 20
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), f"libcdc_hip.so does not export {name}"
    assert sorted(_lib.EXPORTS) == declared
    assert b"gfx950" in L.cdc_version()


@pytest.mark.parametrize("name", ["small_x", "small_eps", "odd_x", "full_x", "full_eps"])
def test_manifest_matches_reference_state_dict(name):
    mj = json.load(open(os.path.join(GOLDEN, f"manifest_{name}.json")))
    kw = dict(mj["unet_kwargs"])
    kw.pop("embd_type", None)
    un = cdc.Unet(**kw)
    got = [(n, list(s)) for n, s in un.manifest()]
    assert got == [(n, list(s)) for n, s in mj["manifest"]]


def test_schedule_tables_match_reference():
    g = np.load(os.path.join(GOLDEN, "schedules.npz"))
    for tag, T, vs in (("x", 8193, "cosine"), ("eps", 20000, "linear")):
        for steps in (1, 2, 4, 7, 65, 200, 500, 1000):
            s = schedule.SampleSchedule(T, vs, tag, steps)
            np.testing.assert_array_equal(s.alphas_cumprod, g[f"{tag}_{steps}_alphas_cumprod"])
            np.testing.assert_array_equal(s.one_minus_ac_prev,
                                          g[f"{tag}_{steps}_one_minus_alphas_cumprod_prev"])
            # sqrt-derived tables: torch's vectorised CPU sqrt is not correctly rounded (1-ulp misses)
            np.testing.assert_array_max_ulp(s.sqrt_recip, g[f"{tag}_{steps}_sqrt_recip_alphas_cumprod"], 2)
            np.testing.assert_array_max_ulp(s.sqrt_recipm1,
                                            g[f"{tag}_{steps}_sqrt_recipm1_alphas_cumprod"], 2)
            np.testing.assert_array_max_ulp(s.sqrt_ac_prev, g[f"{tag}_{steps}_sqrt_alphas_cumprod_prev"], 2)
            np.testing.assert_array_max_ulp(s.sigma, g[f"{tag}_{steps}_sigma"], 6)
            if tag == "x":
                np.testing.assert_array_equal(s.index, g[f"x_{steps}_index"])
                np.testing.assert_array_equal(
                    s.time_in, (g[f"x_{steps}_index"].astype(np.float32) / np.float32(8193)))


def test_linspace_index_matches_torch():
    torch = pytest.importorskip("torch")
    for T in (8193, 20000):
        for steps in list(range(1, 130)) + [200, 333, 500, 777, 1000]:
            np.testing.assert_array_equal(schedule.linspace_index(T, steps),
                                          torch.linspace(0, T - 1, steps).long().numpy())


def test_load_state_dict_is_strict():
    un = cdc.Unet(dim=16, channels=3, context_channels=8, dim_mults=(1, 2, 3), context_dim_mults=(1, 2))
    man = un.manifest()
    sd = {k: np.zeros(s, np.float32) for k, s in man}
    bad = dict(sd)
    bad.pop("final_conv.1.bias")
    with pytest.raises(RuntimeError, match="missing"):
        un.load_state_dict(bad)
    bad = dict(sd)
    bad["downs.0.0.block1.block.0.weight"] = np.zeros((16, 11, 3, 3), np.float32)
    with pytest.raises(_lib.CdcError, match="size mismatch"):
        un.load_state_dict(bad)


@pytest.mark.skipif(_has_gpu(), reason="only meaningful on a host without a GPU")
def test_compute_fails_loudly_without_gpu():
    un = cdc.Unet(dim=16, channels=3, context_channels=8, dim_mults=(1, 2, 3), context_dim_mults=(1, 2))
    sd = {k: np.zeros(s, np.float32) for k, s in un.manifest()}
    with pytest.raises(_lib.CdcError, match="no HIP device"):
        un.load_state_dict(sd)
    from cdc_compression_amd.ops import Ops
    with pytest.raises(_lib.CdcError, match="no HIP device"):
        Ops().chan_layernorm(np.zeros((1, 4, 2, 2), np.float32), np.ones(4), np.zeros(4))


@pytest.mark.skipif(_has_gpu(), reason="only meaningful on a host without a GPU")
def test_probes_fail_loudly_without_gpu():
    import ctypes
    L = _lib.lib()
    v = ctypes.c_double(-1.0)
    assert L.cdc_probe_mfma_f16(0, 1, 100, ctypes.byref(v)) < 0
    assert L.cdc_probe_hbm_copy(0, 1 << 20, 1, ctypes.byref(v)) < 0


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "cdc_compression_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.lower() or f == "synth.py", f"{f} mentions the oracle"


# ---- context decoder -----------------------------------------------------

def _ctxdec_model(name):
    meta = json.load(open(os.path.join(GOLDEN, f"manifest_{name}.json")))
    cls = getattr(cdc, meta["class"])
    return cls(**meta["kwargs"]), meta


@pytest.mark.parametrize("name", ["ctxdec_small_x", "ctxdec_small_eps", "ctxdec_full_x", "ctxdec_full_eps"])
def test_ctxdec_manifest_matches_reference_state_dict(name):
    m, meta = _ctxdec_model(name)
    assert [(n, list(s)) for n, s in m.manifest()] == [(n, list(s)) for n, s in meta["manifest"]]


def test_ctxdec_load_state_dict_takes_dec_entries_only():
    m, meta = _ctxdec_model("ctxdec_small_x")
    sd = {k: np.zeros(s, np.float32) for k, s in meta["manifest"]}
    bad = dict(sd)
    bad.pop(meta["manifest"][3][0])
    with pytest.raises(RuntimeError, match="missing"):
        m.load_state_dict(bad)
    bad = dict(sd)
    bad["dec.9.0.block1.block.0.weight"] = np.zeros((1,), np.float32)
    with pytest.raises(RuntimeError, match="unexpected"):
        m.load_state_dict(bad)
    with pytest.raises(_lib.CdcError, match="load_encoder_state_dict"):      # encoder weights never loaded
        m.encode(np.zeros((1, 3, 64, 64), np.float32))
    if not _has_gpu():
        full = dict(sd)
        full["enc.0.0.block1.block.0.weight"] = np.zeros((8, 3, 7, 7), np.float32)   # encoder keys are ignored
        with pytest.raises(_lib.CdcError):      # finalize needs the GPU: loud failure, no CPU fallback
            m.load_state_dict(full)


@pytest.mark.parametrize("name", ["hyperdec_small_x", "hyperdec_full_x", "hyperdec_full_eps"])
def test_hyperdec_manifest_matches_reference_state_dict(name):
    meta = json.load(open(os.path.join(GOLDEN, f"manifest_{name}.json")))
    m = getattr(cdc, meta["class"])(**meta["kwargs"])
    assert m.reversed_hyper_dims == meta["dims"]
    assert [(n, list(s)) for n, s in m.hyper_manifest()] == [(n, list(s)) for n, s in meta["manifest"]]


@pytest.mark.parametrize("name", ["encoder_small_x", "encoder_full_x", "encoder_full_eps"])
def test_encoder_manifest_matches_reference_state_dict(name):
    meta = json.load(open(os.path.join(GOLDEN, f"manifest_{name}.json")))
    m = getattr(cdc, meta["class"])(**meta["kwargs"])
    want = [(n, list(s)) for n, s in meta["manifest"] if n.startswith("enc.") or n.startswith("hyper_enc.")]
    assert [(n, list(s)) for n, s in m.encoder_manifest()] == want


# ---- sanitizer build of the C-ABI ------------------------------------------------

def test_c_abi_host_paths_under_address_sanitizer():
    """`make asan` (csrc/Makefile: the host side of the C-ABI under AddressSanitizer, device code untouched) and one pass over the
    life cycle and the error paths of every handle kind through it -- create / manifest / strict load / finalize / destroy, raw
    C-ABI misuse, the entropy container parser on hostile bytes (tests/asan_driver.py).  Runs without a GPU: every compute call
    must fail loudly there, not crash."""
    import shutil
    import subprocess
    import sys
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc) or not shutil.which("make"):
        pytest.skip("needs hipcc + make")
    csrc = os.path.join(ROOT, "cdc_compression_amd", "csrc")
    subprocess.check_call(["make", "-C", csrc, "-j8", "asan"], stdout=subprocess.DEVNULL)
    rt = subprocess.check_output([hipcc, "-print-file-name=libclang_rt.asan-x86_64.so"], text=True).strip()
    assert os.path.exists(rt), rt
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               CDC_HIP_LIB=os.path.join(ROOT, "cdc_compression_amd", "libcdc_hip_asan.so"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asan_driver.py")], env=env, capture_output=True, text=True, timeout=600)
    assert "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0 and "ASAN_DRIVER_OK" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-2000:])


def test_bench_roofline_object_leads_with_its_scalars():
    """The driver's record keeps the scalar fields of `roofline` in order and drops what is nested or late: numbers first,
    `frac_of_measured_sustained` and the measured ceilings among them; tables and prose behind."""
    import bench
    ops = [dict(label="conv 3x3 s1   64->64   out 256x256 MB2 NPW2 WM1 WP4 g1 R5 PF3 LN nof32 +pf", ms=0.41, n=4, flops=154.6e9),
           dict(label="conv 3x3 s1   64->64   out 256x256 MB2 NPW2 WM1 WP4 g1 R5 PF3 LN +resP", ms=0.48, n=4, flops=154.6e9),
           dict(label="conv 1x1 s1   64->64   out 256x256 MB2 NPW2 WM1 WP4 g1 R6 PW pre nof32 +pf +res", ms=0.29, n=4, flops=17.2e9)]
    classes = {k: dict(ms=1.0, launches=4, flops=1e11, bytes=1e9) for k in ("conv3x3", "conv1x1", "layernorm")}
    ceil = {"mfma": {"tflops_random_operands": 1500.0, "tflops_constant_operands": 2400.0}, "hbm": {"gb_per_s": 4900.0}}
    r = bench.roofline_block(classes, ops, 32, 256, 1, 5.3, 500, 1, 6.0, bench.FULL["x"], 125, ceilings=ceil)
    keys = list(r)
    assert keys[:5] == ["bound", "achieved", "peak", "unit", "frac"]
    lead = keys[:keys.index("launches")]
    assert {"traffic", "frac_of_measured_sustained", "mfma_sustained_tflops_measured", "hbm_copy_tb_s_measured", "whole_path_tflops_canonical"} <= set(lead)
    assert all(not isinstance(r[k], (dict, list)) for k in lead)
    assert abs(r["frac_of_measured_sustained"] - r["mfma_tflops_executed"] / 1500.0) < 1e-9
    assert r["kernel"] == "conv_pf3_kernel" and r["launches_per_iteration"] == 2
