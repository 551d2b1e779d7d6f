"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C-ABI, against
(1) golden vectors produced by the real reference and (2) the CPU oracle on the same seeded inputs.

Tolerance: float32 path, max-abs error <= 1e-4 * max(1, max|ref|) (north_star: fp32 tolerance;
observed reference-vs-fp64 round-off is ~7e-6, see tests/test_oracle.py)."""
import ctypes
import json
import os

import numpy as np
import pytest

import cdc_compression_amd as cdc
from cdc_compression_amd import _lib, synth
from cdc_compression_amd.ops import Ops
from oracle import model as om
from oracle import ops as oops
from helpers import GOLDEN, digest_idx, load_case, oracle_cfg

pytestmark = pytest.mark.gpu
TOL = 1e-4         # north_star's statement.  The test bounds below are ~3x what is MEASURED on the MI355X:
TOL_FWD = 1e-5     #   one U-Net / compressor forward (or a stage of it) against the reference golden: measured <= 3.0e-6
TOL_DEC = 5e-5     #   a few-step / full-length decode chain against the reference golden: measured <= 2.1e-5
# (CDC_TEST_OBS=<file>: every relerr() of a run is appended there with its test id -- how the bounds were measured;
#  tools/gpu_profiles_r06.sh, summary under profiles/parity_obs_r06.txt)


def relerr(a, ref):
    e = float(np.abs(a - ref).max()) / max(1.0, float(np.abs(ref).max()))
    obs = os.environ.get("CDC_TEST_OBS")
    if obs:
        with open(obs, "a") as f:
            f.write(json.dumps({"test": os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], "relerr": e}) + "\n")
    return e


@pytest.fixture(scope="module")
def O():
    return oops.OrcOps("f32")


@pytest.fixture(scope="module")
def G():
    return Ops(0)


def test_native_library_loaded():
    L = _lib.lib()
    assert os.path.samefile(L._name, _lib.LIB_PATH)


CONV_CASES = [
    # B, Cin, H, W, Cout, k, stride, pad, fused_ln
    (2, 5, 9, 11, 7, 3, 1, 1, False),
    (1, 4, 12, 10, 6, 3, 2, 1, False),
    (1, 3, 10, 9, 4, 7, 1, 3, False),
    (2, 6, 5, 5, 9, 1, 1, 0, False),
    (2, 64, 32, 32, 64, 3, 1, 1, True),
    (1, 67, 40, 48, 64, 7, 1, 3, True),
    (2, 128, 16, 16, 128, 3, 1, 1, True),
    (1, 256, 16, 32, 192, 3, 1, 1, True),
    (1, 384, 8, 8, 384, 3, 1, 1, True),
    (4, 320, 16, 16, 320, 3, 1, 1, True),
    (1, 64, 64, 64, 64, 3, 2, 1, False),
    (1, 64, 36, 20, 192, 1, 1, 0, False),
    (1, 64, 48, 48, 3, 7, 1, 3, False),
    (1, 24, 24, 40, 24, 3, 1, 1, True),     # Cout % 32 != 0 -> unfused LN path
    (2, 256, 32, 32, 256, 3, 1, 1, True),   # 256 channels: eight channel blocks per wave, fused LayerNorm
    (1, 192, 16, 32, 256, 3, 1, 1, True),
    (2, 320, 16, 16, 320, 3, 2, 1, False),  # stride 2 down to 8x8: two-unit patches + split-K
    (1, 256, 32, 32, 256, 3, 2, 1, False),
]


PF_CASES = [
    # pre-split operand kernel (conv_pf_kernel: W >= 32, Cin % 16 == 0, Cout % 32 == 0), one case per tile shape
    (2, 128, 20, 64, 128, 3, 1, 1, True),   # 128 channels: two channel parts per workgroup, cross-wave LayerNorm
    (1, 192, 36, 32, 192, 3, 1, 1, True),   # 192 channels, eight waves, ragged rows
    (1, 256, 32, 32, 256, 3, 1, 1, True),   # 256 channels, four channel parts
    (2, 384, 32, 32, 128, 1, 1, 0, False),  # 1x1 res_conv: one tap per chunk, three patch buffers
    (1, 64, 33, 48, 64, 3, 1, 1, True),     # ragged in both directions
    (1, 64, 40, 96, 96, 3, 1, 1, False),    # three channel blocks per wave
    (1, 16, 32, 32, 32, 3, 1, 1, False),    # a single chunk: weight ring longer than the tile
]

PF3_CASES = [
    # persistent ping-ponged kernel (conv_pf3_kernel: 64 / 128 output channels, tiles divisible by 2 x workgroups, >= 2 per group)
    (4, 64, 128, 256, 64, 3, 1, 1, True),   # 512 tiles of 8 x 32 on 128 workgroups
    (4, 128, 64, 256, 128, 3, 1, 1, True),  # 128 channels: cross-wave LayerNorm, 4-row tiles
    (4, 128, 128, 256, 64, 3, 1, 1, True),  # Cin != Cout: eight chunks
    (6, 64, 64, 256, 64, 3, 1, 1, True),    # 384 tiles: 96 workgroups is below half the chip -> stays on conv_pf_kernel
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_matches_oracle(O, G, case):
    B, Ci, H, W, Co, k, s, p, fused = case
    x = synth.normal("cx", (B, Ci, H, W), 21)
    w = synth.normal("cw", (Co, Ci, k, k), 21, 1.0 / np.sqrt(Ci * k * k))
    b = synth.normal("cb", (Co,), 21, 0.1)
    ref = O.conv2d(x, w, b, s, p)
    got = G.conv2d(x, w, b, s, p)
    assert got.shape == ref.shape
    assert relerr(got, ref) < 1e-5, relerr(got, ref)
    if fused:
        g = synth.normal("cg", (Co,), 21, 0.2, 1.0)
        bb = synth.normal("cbb", (Co,), 21, 0.2)
        shift = synth.normal("cs", (B, Co), 21, 0.3)
        resid = synth.normal("cr", ref.shape, 21)
        r2 = np.maximum(O.chan_layernorm(ref, g, bb), 0) + shift[:, :, None, None] + resid
        g2 = G.conv2d(x, w, b, s, p, ln_g=g, ln_b=bb, relu=True, shift=shift, resid=resid)
        assert relerr(g2, r2) < 1e-5, relerr(g2, r2)


def _conv_check(O, G, case, tol_plain=1e-5, tol_fused=1e-5):
    B, Ci, H, W, Co, k, s, p, fused = case
    x = synth.normal("cx", (B, Ci, H, W), 21)
    w = synth.normal("cw", (Co, Ci, k, k), 21, 1.0 / np.sqrt(Ci * k * k))
    b = synth.normal("cb", (Co,), 21, 0.1)
    ref = O.conv2d(x, w, b, s, p)
    got = G.conv2d(x, w, b, s, p)
    assert relerr(got, ref) < tol_plain, relerr(got, ref)
    if fused:
        g = synth.normal("cg", (Co,), 21, 0.2, 1.0)
        bb = synth.normal("cbb", (Co,), 21, 0.2)
        shift = synth.normal("cs", (B, Co), 21, 0.3)
        resid = synth.normal("cr", ref.shape, 21)
        r2 = np.maximum(O.chan_layernorm(ref, g, bb), 0) + shift[:, :, None, None] + resid
        g2 = G.conv2d(x, w, b, s, p, ln_g=g, ln_b=bb, relu=True, shift=shift, resid=resid)
        assert relerr(g2, r2) < tol_fused, relerr(g2, r2)


@pytest.mark.parametrize("walk", ["column-major", "row-major"])
@pytest.mark.parametrize("case", PF3_CASES)
def test_conv2d_persistent_ping_pong_kernel(O, case, walk, monkeypatch):
    """conv_pf3_kernel (CDC_PF=1, large 3x3 layers): plain, and with LayerNorm + ReLU + per-image shift + residual; both tile walk
    orders of a workgroup's range (column-major is the default, CDC_PF3_XMAJOR=1 the row-major one)."""
    monkeypatch.setenv("CDC_PF", "1")
    monkeypatch.setenv("CDC_PF_MAXPIX", "0")
    if walk == "row-major":
        monkeypatch.setenv("CDC_PF3_XMAJOR", "1")
    _conv_check(O, Ops(0), case)


@pytest.mark.parametrize("case", PF_CASES + CONV_CASES[4:6])
def test_conv2d_pre_split_operand_kernel(O, case, monkeypatch):
    """conv_pf_kernel (CDC_PF=1): activations arrive as two fp16 planes by LDS-DMA; every tile shape of its planner."""
    monkeypatch.setenv("CDC_PF", "1")
    monkeypatch.setenv("CDC_PF_MAXPIX", "0")
    _conv_check(O, Ops(0), case)


PF_S2_CASES = [
    # conv_pf_kernel<..., STR = 2> (3x3 / stride 2 / pad 1 Downsample on plane operands), one case per tile shape
    (2, 64, 64, 128, 64, 3, 2, 1, False),    # 64 channels: four waves, one pixel block each
    (1, 128, 36, 64, 128, 3, 2, 1, False),   # 128 channels: two channel parts, ragged rows (18 = 4 * 4 + 2)
    (1, 192, 32, 96, 192, 3, 2, 1, False),   # 192 channels: eight waves, ragged columns (48 = 32 + 16)
    (1, 64, 64, 64, 128, 3, 2, 1, False),    # Cin != Cout
    (1, 32, 64, 64, 192, 3, 2, 1, False),    # two chunks only
]


WS_CASES = [
    # weight-stationary kernel of the few-pixel levels (conv_ws_kernel: maps 8 / 16 / 32 wide, 128-pixel tiles, K sliced over the waves)
    (2, 384, 8, 8, 384, 3, 1, 1, True),     # 8x8: a tile = two whole images, 24 chunks over 8 waves
    (6, 320, 8, 8, 384, 3, 1, 1, True),     # 20 chunks over 5 waves, three tiles
    (1, 320, 16, 16, 320, 3, 1, 1, True),   # 16x16: a tile = 8 rows of one image (neighbour rows loaded, image border rows stay zero)
    (3, 256, 16, 16, 320, 3, 1, 1, True),
    (1, 64, 32, 32, 96, 3, 1, 1, True),     # 32 wide: 4-row bands; 4 chunks over 4 waves; 3 channel groups
    (2, 48, 8, 8, 32, 3, 1, 1, True),       # 3 chunks over 3 waves, fewer waves than pixel blocks
    (4, 64, 16, 8, 64, 3, 1, 1, True),      # non-square map, 8 wide: 128 pixels per image
    (1, 128, 8, 64, 192, 3, 1, 1, True),    # 64 wide (one image per call at the 64 x 64 level): a tile = one or two rows
]


@pytest.mark.parametrize("npb", ["2", "4"])
@pytest.mark.parametrize("case", WS_CASES)
def test_conv2d_weight_stationary_kernel(O, case, npb, monkeypatch):
    """conv + LayerNorm + ReLU + shift + residual of a few-pixel level: conv_ws_kernel (raw result, all of K inside one workgroup) + the
    in-place LayerNorm pass, and the bias-only convolution (the raw result itself), against the oracle; the launch really is that kernel."""
    monkeypatch.setenv("CDC_WS_MIN_WGS", "1")
    monkeypatch.setenv("CDC_WS_NPB", npb)             # 64- / 128-pixel tiles
    monkeypatch.setenv("CDC_OP_REQUIRE_WS", "1")      # fails instead of falling back to the register-staged kernel
    from cdc_compression_amd.ops import Ops
    B, Ci, H, W, Co, k, s, p, fused = case
    x = synth.normal("cx", (B, Ci, H, W), 21)
    w = synth.normal("cw", (Co, Ci, k, k), 21, 1.0 / np.sqrt(Ci * k * k))
    b = synth.normal("cb", (Co,), 21, 0.1)
    g = synth.normal("cg", (Co,), 21, 0.2, 1.0)
    bb = synth.normal("cbb", (Co,), 21, 0.2)
    shift = synth.normal("cs", (B, Co), 21, 0.3)
    ref = O.conv2d(x, w, b, s, p)
    resid = synth.normal("cr", ref.shape, 21)
    r2 = np.maximum(O.chan_layernorm(ref, g, bb), 0) + shift[:, :, None, None] + resid
    G2 = Ops(0)
    g2 = G2.conv2d(x, w, b, s, p, ln_g=g, ln_b=bb, relu=True, shift=shift, resid=resid)
    assert relerr(g2, r2) < 1e-5, relerr(g2, r2)
    assert relerr(G2.conv2d(x, w, b, s, p), ref) < 1e-5


@pytest.mark.parametrize("case", [(2, 320, 16, 16, 320), (1, 256, 32, 32, 256), (1, 64, 64, 64, 64), (3, 192, 32, 64, 192), (1, 128, 64, 32, 96)])
def test_conv2d_stride2_weight_stationary_kernel(O, case, monkeypatch):
    """The Downsample (3x3 / stride 2 / pad 1) of a small launch on conv_ws_kernel's stride-2 form (all of K inside the workgroup) against
    the oracle; the launch really is that kernel."""
    monkeypatch.setenv("CDC_WS_MIN_WGS", "1")
    monkeypatch.setenv("CDC_OP_REQUIRE_WS", "1")
    from cdc_compression_amd.ops import Ops
    B, Ci, H, W, Co = case
    x = synth.normal("cx", (B, Ci, H, W), 41)
    w = synth.normal("cw", (Co, Ci, 3, 3), 41, 1.0 / np.sqrt(Ci * 9))
    b = synth.normal("cb", (Co,), 41, 0.1)
    ref = O.conv2d(x, w, b, 2, 1)
    got = Ops(0).conv2d(x, w, b, 2, 1)
    assert got.shape == ref.shape and relerr(got, ref) < 1e-5, relerr(got, ref)


WS1_CASES = [
    # 1x1 layers of the few-pixel levels on conv_ws1_kernel (all of K inside the workgroup; 32-pixel blocks of the flattened batch)
    (2, 384, 8, 8, 384), (4, 320, 8, 8, 960), (1, 256, 16, 16, 768), (3, 640, 16, 16, 256), (2, 48, 8, 8, 32), (1, 64, 4, 24, 96),
    (1, 192, 32, 32, 576), (1, 128, 16, 64, 128),          # wider maps, small launches (one image per call)
]


@pytest.mark.parametrize("npb", ["2", "4"])
@pytest.mark.parametrize("case", WS1_CASES)
def test_conv2d_pointwise_few_pixel_kernel(O, case, npb, monkeypatch):
    """bias-only and bias + per-image shift + residual 1x1 convolutions on conv_ws1_kernel against the oracle (64- and 128-pixel tiles;
    a case whose batch does not divide into the forced tile stays on the round-4 kernels and is still checked)."""
    monkeypatch.setenv("CDC_WS1_MIN_WGS", "1")
    monkeypatch.setenv("CDC_WS1_NPB", npb)
    from cdc_compression_amd.ops import Ops
    B, Ci, H, W, Co = case
    x = synth.normal("cx", (B, Ci, H, W), 31)
    w = synth.normal("cw", (Co, Ci, 1, 1), 31, 1.0 / np.sqrt(Ci))
    b = synth.normal("cb", (Co,), 31, 0.1)
    shift = synth.normal("cs", (B, Co), 31, 0.3)
    ref = O.conv2d(x, w, b, 1, 0)
    resid = synth.normal("cr", ref.shape, 31)
    G2 = Ops(0)
    assert relerr(G2.conv2d(x, w, b, 1, 0), ref) < 1e-5
    got = G2.conv2d(x, w, b, 1, 0, shift=shift, resid=resid)
    assert relerr(got, ref + shift[:, :, None, None] + resid) < 1e-5


@pytest.mark.parametrize("npb", ["2", "4"])
def test_weight_stationary_trunk_matches_the_round4_program(npb, monkeypatch):
    """The few-pixel trunk of the full-width model on conv_ws_kernel (raw result + in-place LayerNorm pass) with 64- and 128-pixel tiles
    forced, one 256 x 256 image and a batch of 4, against the program without the kernel (CDC_WS_MIN_WGS huge: split-K convolutions +
    LayerNorm over the partial sums), which the reference digests pin; and run-to-run determinism (fixed summation order)."""
    kw, man, sd, _, _, _, _ = load_case("full_x")
    S = 256
    for B in (1, 4):
        x = synth.normal("x", (B, 3, S, S), seed=51, std=0.8)
        t = np.full((B, 1), 0.3, np.float32)
        ctx = [synth.normal(f"c{l}", (B, c, S >> l, S >> l), seed=52, std=0.5) for l, c in enumerate([64, 64, 128, 192])]
        monkeypatch.setenv("CDC_WS_MIN_WGS", "1000000")
        monkeypatch.setenv("CDC_WS1_MIN_WGS", "1000000")
        un = cdc.Unet(**kw)
        un.load_state_dict(sd)
        y = un(x, t, ctx)
        assert not [l for l in _op_labels(un) if " WS" in l]
        monkeypatch.setenv("CDC_WS_MIN_WGS", "1")
        monkeypatch.setenv("CDC_WS1_MIN_WGS", "1")
        monkeypatch.setenv("CDC_WS_NPB", npb)
        monkeypatch.setenv("CDC_WS1_NPB", npb)
        un2 = cdc.Unet(**kw)
        un2.load_state_dict(sd)
        y2 = un2(x, t, ctx)
        ws = [l for l in _op_labels(un2) if " WS " in l or l.endswith(" WS")]
        ws1 = [l for l in _op_labels(un2) if " WS1" in l]
        assert len(ws) >= 8 and all(f"NPB{npb}" in l for l in ws), ws
        # the 1x1 family of the trunk: folded-PreNorm projections, per-image attention products, to_out with the residual, res_convs
        # (per-image weight sets exist in the 64-pixel form only: one set per pixel block)
        assert any(" pre" in l for l in ws1) and (npb == "4" or any("perimg" in l for l in ws1)) and any("+res" in l for l in ws1) and len(ws1) >= 4, ws1
        assert relerr(y2, y) < 5e-6, (B, relerr(y2, y))
        for _ in range(5):
            np.testing.assert_array_equal(un2(x, t, ctx), y2)
        monkeypatch.delenv("CDC_WS_NPB")
        monkeypatch.delenv("CDC_WS1_NPB")


@pytest.mark.parametrize("case", PF_S2_CASES)
def test_conv2d_stride2_on_plane_operands(O, case, monkeypatch):
    """Downsample convolutions on conv_pf_kernel (STR = 2): de-interleaved patch columns, 4-row tiles."""
    monkeypatch.setenv("CDC_PF", "1")
    monkeypatch.setenv("CDC_PF_MAXPIX", "0")
    monkeypatch.setenv("CDC_PF_S2_MIN_WGS", "1")
    monkeypatch.setenv("CDC_OP_REQUIRE_PF", "1")      # fails instead of falling back to the register-staged kernel
    _conv_check(O, Ops(0), case)


PW_CASES = [  # B, Cin, H, W, Cout: every tile shape of conv_pw_kernel's planner, ragged rows, linear (narrow-map) tiles
    (2, 64, 36, 64, 64), (1, 128, 20, 32, 128), (2, 256, 32, 32, 256), (1, 96, 64, 64, 192), (1, 64, 40, 96, 384),
    (3, 320, 16, 16, 128), (4, 384, 8, 8, 256), (1, 48, 12, 16, 64)]


def test_conv2d_pointwise_kernel_4_byte_activation_pieces(O, monkeypatch):
    """conv_pw_kernel's older activation path (eight 4-byte LDS-DMA pieces per chunk and block; the default is two 16-byte ones)."""
    monkeypatch.setenv("CDC_PW_MIN_WAVES", "1")
    monkeypatch.setenv("CDC_NO_PW_X16", "1")
    G2 = Ops(0)
    for (B, Ci, H, W, Co) in [(2, 64, 36, 64, 64), (3, 320, 16, 16, 128), (1, 64, 40, 96, 384)]:
        x = synth.normal("px", (B, Ci, H, W), 25)
        w = synth.normal("pw", (Co, Ci, 1, 1), 25, 1.0 / np.sqrt(Ci))
        b = synth.normal("pb", (Co,), 25, 0.1)
        assert relerr(G2.conv2d(x, w, b, 1, 0), O.conv2d(x, w, b, 1, 0)) < 1e-5


@pytest.mark.parametrize("case", PW_CASES)
def test_conv2d_pointwise_kernel(O, case, monkeypatch):
    """conv_pw_kernel (1x1, activations staged per wave straight from the fp32 tensor), forced also for small launches;
    plain, and with ReLU + per-image shift + residual."""
    monkeypatch.setenv("CDC_PW_MIN_WAVES", "1")
    G2 = Ops(0)
    B, Ci, H, W, Co = case
    x = synth.normal("px", (B, Ci, H, W), 27)
    w = synth.normal("pw", (Co, Ci, 1, 1), 27, 1.0 / np.sqrt(Ci))
    b = synth.normal("pb", (Co,), 27, 0.1)
    ref = O.conv2d(x, w, b, 1, 0)
    assert relerr(G2.conv2d(x, w, b, 1, 0), ref) < 1e-5
    resid = synth.normal("pr", ref.shape, 27)
    r2 = ref + resid
    assert relerr(G2.conv2d(x, w, b, 1, 0, resid=resid), r2) < 1e-5


@pytest.mark.parametrize("case", [CONV_CASES[4], CONV_CASES[6], CONV_CASES[7], CONV_CASES[10], CONV_CASES[11]] + PF_CASES[:2])
def test_conv2d_three_plane_bf16_arithmetic(O, case, monkeypatch):
    """CDC_ARITH=0: the exact three-way bf16 split (six MFMA products) stays available as the full-range path."""
    monkeypatch.setenv("CDC_ARITH", "0")
    _conv_check(O, Ops(0), case)


@pytest.mark.parametrize("case", [(2, 5, 6, 7, 4), (1, 64, 16, 16, 64), (1, 320, 8, 8, 320),
                                  (1, 24, 12, 20, 24), (2, 64, 24, 32, 64), (1, 192, 32, 32, 192), (1, 128, 33, 64, 128)])
def test_conv_transpose2d_matches_oracle(O, G, case):
    B, Ci, H, W, Co = case
    x = synth.normal("tx", (B, Ci, H, W), 22)
    w = synth.normal("tw", (Ci, Co, 4, 4), 22, 1.0 / np.sqrt(Ci * 4))
    b = synth.normal("tb", (Co,), 22, 0.1)
    ref = O.conv_transpose2d(x, w, b, 2, 1)
    got = G.conv_transpose2d(x, w, b)
    assert relerr(got, ref) < 1e-5, relerr(got, ref)


PF_TZ_CASES = [  # B, Cin, H, W, Cout: ConvTranspose2d 4x4 / stride 2 / pad 1 with its four phases fused on conv_pf_kernel (TZ = 4)
    (2, 64, 32, 64, 64),     # 64 channels: two channel parts x two row pairs
    (1, 128, 18, 32, 128),   # 128 channels, eight waves, ragged rows (18 = 4 * 4 + 2)
    (1, 64, 16, 48, 64),     # ragged columns (48 = 32 + 16)
    (1, 32, 8, 32, 64),      # two chunks only
]


@pytest.mark.parametrize("case,plan", [(c, None) for c in PF_TZ_CASES] + [(PF_TZ_CASES[0], "1,2,2,2"), (PF_TZ_CASES[2], "1,2,2,2")])
def test_conv_transpose2d_fused_phases_on_plane_operands(O, case, plan, monkeypatch):
    if plan:
        monkeypatch.setenv("CDC_PF_PLAN", plan)           # the other 64-channel wave shape (two channel parts x two row pairs)
    monkeypatch.setenv("CDC_PF", "1")
    monkeypatch.setenv("CDC_PF_MAXPIX", "0")
    monkeypatch.setenv("CDC_PF_TZ_MIN_WGS", "1")
    monkeypatch.setenv("CDC_OP_REQUIRE_PF", "1")
    B, Ci, H, W, Co = case
    x = synth.normal("tx", (B, Ci, H, W), 22)
    w = synth.normal("tw", (Ci, Co, 4, 4), 22, 1.0 / np.sqrt(Ci * 4))
    b = synth.normal("tb", (Co,), 22, 0.1)
    ref = O.conv_transpose2d(x, w, b, 2, 1)
    got = Ops(0).conv_transpose2d(x, w, b)
    assert relerr(got, ref) < 1e-5, relerr(got, ref)


def test_layernorm_matches_oracle(O, G):
    x = synth.normal("lx", (2, 48, 9, 7), 23, 2.0, 0.5)
    g = synth.normal("lg", (48,), 23, 0.2, 1.0)
    b = synth.normal("lb", (48,), 23, 0.2)
    assert relerr(G.chan_layernorm(x, g, b), O.chan_layernorm(x, g, b)) < 1e-6


@pytest.mark.parametrize("env", [{}, {"CDC_LN_VEC_COLS": "8"}, {"CDC_LN_VEC_COLS": "4"}, {"CDC_NO_LN_VEC": "1"}])
@pytest.mark.parametrize("shape", [(2, 48, 8, 8), (3, 320, 16, 16), (2, 384, 8, 8), (1, 256, 12, 20), (33, 40, 4, 4)])
def test_layernorm_on_16_byte_accesses(O, shape, env, monkeypatch):
    """ln_kernel_vec (pixel counts that are multiples of 4): every column count, whole and partial workgroups, against the
    oracle and against the 4-byte kernel it replaces."""
    monkeypatch.setenv("CDC_DEV", "1")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    B, C, H, W = shape
    x = synth.normal("lvx", shape, 29, 2.0, 0.5)
    g = synth.normal("lvg", (C,), 29, 0.2, 1.0)
    b = synth.normal("lvb", (C,), 29, 0.2)
    assert relerr(Ops(0).chan_layernorm(x, g, b), O.chan_layernorm(x, g, b)) < 1e-6


@pytest.mark.parametrize("case", [(2, 16, 8, 8), (1, 64, 32, 32), (2, 64, 64, 64), (1, 128, 64, 64), (2, 128, 16, 16), (1, 384, 8, 8),
                                  (1, 24, 12, 20), (1, 64, 64, 64)])
def test_linear_attention_matches_oracle(O, G, case):
    B, C, H, W = case
    x = synth.normal("ax", (B, C, H, W), 24)
    sd = {"a.fn.norm.g": synth.normal("ag", (1, C, 1, 1), 24, 0.2, 1.0),
          "a.fn.norm.b": synth.normal("ab", (1, C, 1, 1), 24, 0.2),
          "a.fn.fn.to_qkv.weight": synth.normal("aq", (3 * C, C, 1, 1), 24, 2.0 / np.sqrt(C)),
          "a.fn.fn.to_out.weight": synth.normal("ao", (C, C, 1, 1), 24, 1.0 / np.sqrt(C)),
          "a.fn.fn.to_out.bias": synth.normal("aob", (C,), 24, 0.1)}
    ref = om.attention(O, sd, "a", x)
    got = G.linear_attention(x, sd["a.fn.norm.g"], sd["a.fn.norm.b"], sd["a.fn.fn.to_qkv.weight"],
                             sd["a.fn.fn.to_out.weight"], sd["a.fn.fn.to_out.bias"])
    assert relerr(got, ref) < 5e-6, relerr(got, ref)


@pytest.mark.parametrize("case,env", [((2, 128, 16, 16), {"CDC_NO_CTXQ_SPLIT": "1"}),
                                      ((1, 192, 32, 32), {}), ((1, 192, 32, 32), {"CDC_NO_CTXQ_SPLIT": "1"}),
                                      ((2, 64, 64, 64), {"CDC_ARITH": "0"}), ((1, 128, 64, 64), {"CDC_ARITH": "0"}),    # bf16x3: kvctx_kernel, f32 partial context
                                      # few-pixel levels: row maxima + context + reduction in ONE launch (default) / the three-launch chain
                                      ((2, 128, 16, 16), {"CDC_NO_CTX_ONE": "1"}), ((1, 384, 8, 8), {"CDC_NO_CTX_ONE": "1"}), ((3, 320, 16, 16), {}),
                                      ((3, 320, 16, 16), {"CDC_ARITH": "0"}), ((2, 256, 32, 32), {}), ((2, 320, 8, 8), {})])
def test_linear_attention_alternate_kernels(O, case, env, monkeypatch):
    """The non-default attention kernels (register-staged ctx^T q product, the three-launch context chain, and the bf16x3 arithmetic
    with its own fused front half and f32-MFMA partial context) against the same oracle."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    from cdc_compression_amd.ops import Ops
    G2 = Ops(0)                               # (the arithmetic is read when the handle is created)
    B, C, H, W = case
    x = synth.normal("ax", (B, C, H, W), 24)
    sd = {"a.fn.norm.g": synth.normal("ag", (1, C, 1, 1), 24, 0.2, 1.0),
          "a.fn.norm.b": synth.normal("ab", (1, C, 1, 1), 24, 0.2),
          "a.fn.fn.to_qkv.weight": synth.normal("aq", (3 * C, C, 1, 1), 24, 2.0 / np.sqrt(C)),
          "a.fn.fn.to_out.weight": synth.normal("ao", (C, C, 1, 1), 24, 1.0 / np.sqrt(C)),
          "a.fn.fn.to_out.bias": synth.normal("aob", (C,), 24, 0.1)}
    ref = om.attention(O, sd, "a", x)
    got = G2.linear_attention(x, sd["a.fn.norm.g"], sd["a.fn.norm.b"], sd["a.fn.fn.to_qkv.weight"],
                              sd["a.fn.fn.to_out.weight"], sd["a.fn.fn.to_out.bias"])
    assert relerr(got, ref) < 5e-6, relerr(got, ref)


@pytest.mark.parametrize("case", [(2, 64, 64, 64), (1, 128, 64, 64), (1, 192, 64, 64), (3, 64, 32, 64)])
def test_attention_fold_in_one_launch_equals_the_two_launch_fold(O, case, monkeypatch):
    """Folded attention levels (N >= 16 C): the two products of the fold (M' = Wq^T (ctx^T Wo^T)) in ONE launch (fold_r12_mfma_kernel: the
    T1 slab of a column block stays in LDS) against the oracle, and bit-identical to the two launches of round 4 (same products, same order)."""
    from cdc_compression_amd.ops import Ops
    B, C, H, W = case
    x = synth.normal("ax", (B, C, H, W), 24)
    sd = {"a.fn.norm.g": synth.normal("ag", (1, C, 1, 1), 24, 0.2, 1.0),
          "a.fn.norm.b": synth.normal("ab", (1, C, 1, 1), 24, 0.2),
          "a.fn.fn.to_qkv.weight": synth.normal("aq", (3 * C, C, 1, 1), 24, 2.0 / np.sqrt(C)),
          "a.fn.fn.to_out.weight": synth.normal("ao", (C, C, 1, 1), 24, 1.0 / np.sqrt(C)),
          "a.fn.fn.to_out.bias": synth.normal("aob", (C,), 24, 0.1)}
    args = (x, sd["a.fn.norm.g"], sd["a.fn.norm.b"], sd["a.fn.fn.to_qkv.weight"], sd["a.fn.fn.to_out.weight"], sd["a.fn.fn.to_out.bias"])
    ref = om.attention(O, sd, "a", x)
    one = Ops(0).linear_attention(*args)
    monkeypatch.setenv("CDC_DEV", "1")
    monkeypatch.setenv("CDC_FOLD_TWO_LAUNCHES", "1")
    two = Ops(0).linear_attention(*args)
    assert relerr(one, ref) < 5e-6, relerr(one, ref)
    assert np.array_equal(one, two)


def make_unet(name):
    kw, man, sd, x, time, ctx, g = load_case(name)
    un = cdc.Unet(**kw)
    assert [(n, tuple(s)) for n, s in un.manifest()] == man
    un.load_state_dict(sd)
    return un, kw, sd, x, time, ctx, g


@pytest.mark.parametrize("name", ["small_x", "small_eps", "odd_x", "full_x", "full_eps"])
def test_unet_forward_matches_reference_golden(name):
    un, kw, sd, x, time, ctx, g = make_unet(name)
    y = un(x, time, ctx)
    assert y.shape == g["y"].shape
    assert relerr(y, g["y"]) < TOL_FWD, relerr(y, g["y"])


@pytest.mark.parametrize("name,env", [("full_x", {"CDC_PF": "1", "CDC_PF_MAXPIX": "0"}), ("small_x", {"CDC_PF": "1", "CDC_PF_MAXPIX": "0"}),
                                      ("full_eps", {"CDC_PF": "1", "CDC_PF_MAXPIX": "16384"}),
                                      ("full_x", {"CDC_PF": "2"}), ("full_x", {"CDC_PF": "0"}), ("full_eps", {"CDC_PF": "2"}),
                                      ("full_x", {"CDC_PF_S2_MIN_WGS": "1"}),
                                      ("full_eps", {"CDC_PF_S2_MIN_WGS": "1"}), ("full_x", {"CDC_NO_PF_S2": "1"}),
                                      ("full_x", {"CDC_PF_TZ_MIN_WGS": "1"}), ("full_eps", {"CDC_PF_TZ_MIN_WGS": "1", "CDC_PF_S2_MIN_WGS": "1"}),
                                      ("full_x", {"CDC_NO_PF_TZ": "1"}),
                                      ("full_x", {"CDC_PF_17_MIN_WGS": "1"}), ("full_eps", {"CDC_PF_17_MIN_WGS": "1", "CDC_PF_TZ_MIN_WGS": "1"}),
                                      ("small_x", {"CDC_PF_17_MIN_WGS": "1"}), ("full_x", {"CDC_NO_PF_17": "1"}),
                                      # planes-only ResnetBlock-chain outputs, residual read from planes (small launches forced onto conv_pf_kernel)
                                      ("full_x", {"CDC_PF_MIN_WAVES": "1"}), ("full_eps", {"CDC_PF_MIN_WAVES": "1"}), ("small_x", {"CDC_PF_MIN_WAVES": "1"}),
                                      ("full_x", {"CDC_NO_RESID_PF": "1"}), ("full_x", {"CDC_NO_PF_SKIP_PLANES": "1"}),
                                      # the first layer on conv_pf_kernel's UF form (patch buffers built from the 3-channel image)
                                      ("full_x", {"CDC_PF_UF_MIN_WGS": "1"}), ("full_eps", {"CDC_PF_UF_MIN_WGS": "1"}), ("small_x", {"CDC_PF_UF_MIN_WGS": "1"}),
                                      ("odd_x", {"CDC_PF_UF_MIN_WGS": "1"}), ("full_x", {"CDC_NO_PF_UF": "1"}),
                                      # hoisted partial sums in accumulator order (16-byte loads in conv_pf_kernel's epilogue) on / off
                                      ("full_x", {"CDC_PF_MIN_WAVES": "1", "CDC_PF_UF_MIN_WGS": "1"}), ("full_x", {"CDC_PF_MIN_WAVES": "1", "CDC_NO_PRE_C4": "1"}),
                                      ("full_x", {"CDC_PF_MIN_WAVES": "1", "CDC_NO_RESID2": "1"}),     # downs.1.0: concatenated residual materialised again
                                      # ... and planes-only skips (Downsample and decoder join both on plane operands)
                                      ("full_x", {"CDC_PF_MIN_WAVES": "1", "CDC_PF_S2_MIN_WGS": "1", "CDC_PF_TZ_MIN_WGS": "1"}),
                                      ("full_eps", {"CDC_PF_MIN_WAVES": "1", "CDC_PF_S2_MIN_WGS": "1", "CDC_PF_17_MIN_WGS": "1"}),
                                      ("full_x", {"CDC_NO_SPLIT": "1"}), ("full_x", {"CDC_NO_HOIST": "1"}),
                                      ("full_x", {"CDC_NO_KVCTX": "1"}), ("full_x", {"CDC_NO_ATTN_FOLD": "1"}),
                                      ("full_eps", {"CDC_NO_PERIMAGE_SPLIT": "1"}), ("small_x", {"CDC_NO_SPLIT2": "1"}),
                                      ("full_x", {"CDC_NO_PW": "1"}), ("full_x", {"CDC_PW_MIN_WAVES": "1"}), ("full_eps", {"CDC_PW_MIN_WAVES": "1"}),
                                      ("full_x", {"CDC_ARITH": "0"}), ("odd_x", {"CDC_ARITH": "0"})])
def test_unet_forward_alternate_kernel_modes(name, env, monkeypatch):
    """The same goldens through the non-default kernel selections: plane policies, bf16x3 arithmetic, the f32-MFMA convolution path, no context hoisting, unfused / unfolded attention."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    un, kw, sd, x, time, ctx, g = make_unet(name)
    y = un(x, time, ctx)
    assert relerr(y, g["y"]) < TOL_FWD, relerr(y, g["y"])


def test_plane_operand_forms_agree_with_the_register_staged_program_on_ragged_frames():
    """Round-4 forms of conv_pf_kernel (stride 2, fused transposed phases, first / final layer, planes-only tensors, residual from
    planes) and the round-5 few-pixel kernels against the same network with all of them switched off, on frame sizes whose tiles are ragged in both directions (separate
    processes: some of the switches are read once)."""
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "ab_forward.py"), "4x160x224", "3x320x192"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-2000:]


def test_unet_forward_matches_oracle_other_batch(O):
    """Seeded inputs not in the goldens (B=3, different time per row) against the CPU oracle."""
    un, kw, sd, _, _, _, _ = make_unet("small_x")
    B, H, W = 3, 64, 32
    x = synth.normal("x2", (B, 3, H, W), 31)
    ctx = [synth.normal("c0", (B, 8, H, W), 31, 0.5), synth.normal("c1", (B, 16, H // 2, W // 2), 31, 0.5)]
    time = np.array([[0.05], [0.5], [0.93]], np.float32)
    ref = om.unet_forward(O, oracle_cfg(kw), sd, x, time, ctx)
    assert relerr(un(x, time, ctx), ref) < TOL_FWD


@pytest.mark.parametrize("name,param,T,vs", [("small_x", "x", 8193, "cosine"),
                                             ("small_eps", "eps", 20000, "linear"),
                                             ("full_x", "x", 8193, "cosine"),
                                             ("full_eps", "eps", 20000, "linear")])
def test_decode_matches_reference_golden(name, param, T, vs):
    un, kw, sd, x, time, ctx, _ = make_unet(name)
    g = np.load(os.path.join(GOLDEN, f"decode_{name}.npz"))
    init = synth.normal("init", x.shape, seed=1, std=0.8)
    if param == "x":
        diff = cdc.GaussianDiffusionX(un, None, None, num_timesteps=T, pred_mode="x", var_schedule=vs)
    else:
        diff = cdc.GaussianDiffusionEps(un, None, num_timesteps=T, clip_noise="none", pred_mode="noise",
                                        var_schedule=vs)
    for key in [k for k in g.files if k.startswith("decode_")]:
        steps = int(key.split("_")[1])
        rec = diff.decompress(ctx, x.shape, sample_steps=steps, init=init)
        assert relerr(rec, g[key]) < TOL_DEC, (key, relerr(rec, g[key]))
    if "eta_steps" in g.files:
        # eta != 0 with the reference's recorded torch.randn_like draws, fed step by step
        steps = int(g["eta_steps"])
        diff.set_sample_schedule(steps)
        L, h = _lib.lib(), un._handle()
        import ctypes
        img = init.copy()
        ptrs = (ctypes.c_void_p * len(ctx))(*[c.ctypes.data for c in ctx])
        out = np.empty_like(img)
        for cnt, i in enumerate(reversed(range(steps))):
            nz = np.ascontiguousarray(g["eta_noises"][cnt])
            _lib.check(h, L.cdc_ddim_step(h, img.ctypes.data, i, ptrs, len(ctx), nz.ctypes.data, 0.5,
                                          out.ctypes.data, x.shape[0], x.shape[2], x.shape[3],
                                          0 if param == "x" else 1, 1 if param == "x" else 0, 0, None))
            img = out.copy()
        assert relerr(img, g["eta_decode"]) < TOL_DEC


def test_compress_api_with_torch_cuda_tensors():
    torch = pytest.importorskip("torch")
    un, kw, sd, x, time, ctx, g = make_unet("small_x")
    dev = torch.device("cuda:0")
    tctx = [torch.from_numpy(c).to(dev) for c in ctx]

    class Ctx:
        def __call__(self, images):
            return {"output": tctx, "bpp": torch.zeros(images.shape[0], device=dev)}

    diff = cdc.GaussianDiffusionX(un, Ctx(), None, num_timesteps=8193, pred_mode="x",
                                  var_schedule="cosine").to(0).eval()
    gd = np.load(os.path.join(GOLDEN, "decode_small_x.npz"))
    init = torch.from_numpy(synth.normal("init", x.shape, seed=1, std=0.8)).to(dev)
    rec, bpp = diff.compress(torch.zeros(x.shape, device=dev), sample_steps=4, init=init)
    assert rec.is_cuda and rec.shape == tuple(x.shape)
    assert relerr(rec.cpu().numpy(), gd["decode_4"]) < TOL_DEC
    y = un(torch.from_numpy(x).to(dev), torch.from_numpy(time).to(dev), tctx)
    assert relerr(y.cpu().numpy(), g["y"]) < TOL_FWD


def test_full_resolution_256_digest_and_properties():
    """BASELINE-size frame (256x256, full-width x-param model): sampled pixels of the real
    reference's forward + 4-step decode, plus size-independent properties (batch rows are
    independent and identical inputs give identical rows; decode output clamps to [-1,1])."""
    kw, man, sd, _, _, _, _ = load_case("full_x")
    un = cdc.Unet(**kw)
    un.load_state_dict(sd)
    g = np.load(os.path.join(GOLDEN, "full_res_x_256.npz"))
    B, H, W = 1, 256, 256
    x = synth.normal("x", (B, 3, H, W), seed=1, std=0.8)
    ctx = synth.context_pyramid([64, 64, 128, 192], B, H, W, seed=3)
    y = un(x, g["time"], ctx)
    assert relerr(y.reshape(-1)[g["y_idx"]], g["y_val"]) < TOL_FWD
    assert abs(float(y.astype(np.float64).sum()) - float(g["y_sum"])) < 1e-4 * y.size
    diff = cdc.GaussianDiffusionX(un, None, None, num_timesteps=8193, pred_mode="x", var_schedule="cosine")
    init = synth.normal("init", (B, 3, H, W), seed=1, std=0.8)
    rec = diff.decompress(ctx, (B, 3, H, W), sample_steps=4, init=init)
    assert relerr(rec.reshape(-1)[g["dec4_idx"]], g["dec4_val"]) < TOL_DEC
    assert np.abs(rec).max() <= 1.0 + 1e-6
    # batch independence: duplicate the image -> both rows equal the B=1 result bit-for-bit
    x2 = np.concatenate([x, x]); ctx2 = [np.concatenate([c, c]) for c in ctx]
    t2 = np.concatenate([g["time"], g["time"]])
    y2 = un(x2, t2, ctx2)
    np.testing.assert_array_equal(y2[0], y2[1])
    assert relerr(y2[0], y[0]) < 5e-6


def test_run_to_run_determinism(G):
    """Race screen: identical inputs must give bit-identical outputs on every run.  (Guards the
    LDS-DMA staging: with a missing M0 wait state whole workgroup tiles came out wrong ~1/500.)"""
    B, C, H, W = 4, 384, 8, 8
    x = synth.normal("ax", (B, C, H, W), 24)
    ng = synth.normal("ag", (1, C, 1, 1), 24, 0.2, 1.0)
    nb = synth.normal("ab", (1, C, 1, 1), 24, 0.2)
    wq = synth.normal("aq", (3 * C, C, 1, 1), 24, 2.0 / np.sqrt(C))
    wo = synth.normal("ao", (C, C, 1, 1), 24, 1.0 / np.sqrt(C))
    bo = synth.normal("aob", (C,), 24, 0.1)
    ref = G.linear_attention(x, ng, nb, wq, wo, bo)
    for _ in range(60):
        np.testing.assert_array_equal(G.linear_attention(x, ng, nb, wq, wo, bo), ref)
    un, kw, sd, x, time, ctx, g = make_unet("full_x")
    y0 = un(x, time, ctx)
    for _ in range(12):
        np.testing.assert_array_equal(un(x, time, ctx), y0)


# ---- context decoder: Compressor.decode ----------------------------------

def _ctxdec(name):
    meta = json.load(open(os.path.join(GOLDEN, f"manifest_{name}.json")))
    kw = meta["kwargs"]
    man = [(k, tuple(v)) for k, v in meta["manifest"]]
    sd = synth.unet_state_dict(man, seed=5)
    m = getattr(cdc, meta["class"])(**kw)
    m.load_state_dict(sd)
    rev = kw["reverse_dim_mults"] if "reverse_dim_mults" in kw else list(reversed(kw["dim_mults"]))
    cfg = om.CompressorConfig(dim=kw["dim"], rev_mults=rev, out_channels=kw["out_channels"],
                              up_index=meta["up_index"])
    return m, cfg, sd, np.load(os.path.join(GOLDEN, f"{name}.npz"))


@pytest.mark.parametrize("name", ["ctxdec_small_x", "ctxdec_small_eps", "ctxdec_full_x", "ctxdec_full_eps"])
def test_context_decoder_matches_reference_golden(name):
    m, cfg, sd, g = _ctxdec(name)
    outs = m.decode(g["q_latent"])
    assert len(outs) == len(cfg.rev_mults)
    for i, o in enumerate(outs):
        assert list(o.shape) == list(g[f"out{i}_shape"])
        flat = o.reshape(-1)
        assert relerr(flat[g[f"out{i}_idx"]], g[f"out{i}_val"]) < TOL_FWD
        assert abs(float(flat.astype(np.float64).sum()) - float(g[f"out{i}_sum"])) < 1e-4 * flat.size
        if f"out{i}" in g.files:
            assert relerr(o, g[f"out{i}"]) < TOL_FWD


def test_context_decoder_matches_oracle_other_shape_and_feeds_the_unet(O):
    """Another batch / latent size against the CPU restatement, then the decoded pyramid drives the
    denoising U-Net exactly like a host-made context list (decompress(q_latent, ...))."""
    m, cfg, sd, _ = _ctxdec("ctxdec_full_x")
    q = np.round(synth.normal("q2", (2, 256, 2, 4), seed=9, std=2.0)).astype(np.float32)
    outs = m.decode(q)
    ref = om.compressor_decode(O, cfg, sd, q)
    for o, r in zip(outs, ref):
        assert o.shape == r.shape
        assert relerr(o, r) < TOL_FWD
    un, kw, usd, x, time, ctx, g = make_unet("full_x")
    diff = cdc.GaussianDiffusionX(un, m, None, num_timesteps=8193, pred_mode="x", var_schedule="cosine")
    shape = (2, 3, 32, 64)
    init = synth.normal("init", shape, seed=1, std=0.8)
    a = diff.decompress(q, shape, sample_steps=2, init=init)
    b = diff.decompress(outs, shape, sample_steps=2, init=init)
    np.testing.assert_array_equal(a, b)
    assert np.isfinite(a).all() and np.abs(a).max() <= 1.0 + 1e-6


def test_batch32_launch_plans_match_batch1():
    """The launch plans depend on the batch (images per workgroup, split-K slices, attention splits): the
    BASELINE batch of 32 at 256x256 must reproduce the batch-1 result (itself pinned to the reference's
    golden digest above) for every row."""
    kw, man, sd, _, _, _, _ = load_case("full_x")
    un = cdc.Unet(**kw)
    un.load_state_dict(sd)
    H = W = 256
    x = synth.normal("x", (1, 3, H, W), seed=1, std=0.8)
    ctx = synth.context_pyramid([64, 64, 128, 192], 1, H, W, seed=3)
    t = np.full((1, 1), 0.37, np.float32)
    y1 = un(x, t, ctx)
    B = 32
    y32 = un(np.repeat(x, B, 0), np.repeat(t, B, 0), [np.repeat(c, B, 0) for c in ctx])
    for k in (0, 1, 17, 31):
        assert relerr(y32[k], y1[0]) < 5e-6, (k, relerr(y32[k], y1[0]))
    np.testing.assert_array_equal(y32[5], y32[26])


# ---- hyperprior decoder (decode side) ---------------------------------------

@pytest.mark.parametrize("name", ["hyperdec_small_x", "hyperdec_full_x", "hyperdec_full_eps"])
def test_hyper_decoder_matches_reference_golden(name):
    meta = json.load(open(os.path.join(GOLDEN, f"manifest_{name}.json")))
    man = [(k, tuple(v)) for k, v in meta["manifest"]]
    sd = synth.unet_state_dict(man, seed=7)
    m = getattr(cdc, meta["class"])(**meta["kwargs"])
    m.load_hyper_state_dict(sd)
    g = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    mean, scale = m.hyper_decode(g["q_hyper_latent"])
    for key, a in (("mean", mean), ("scale", scale)):
        assert list(a.shape) == list(g[f"{key}_shape"])
        flat = a.reshape(-1)
        assert relerr(flat[g[f"{key}_idx"]], g[f"{key}_val"]) < TOL_FWD
        assert abs(float(flat.astype(np.float64).sum()) - float(g[f"{key}_sum"])) < 1e-4 * flat.size
        if key in g.files:
            assert relerr(a, g[key]) < TOL_FWD
    assert float(scale.min()) >= 0.1
    if "mean" in g.files and "q_latent" in g.files:
        # dequantize is discrete: feed the reference's own mean -> the reference's q_latent, bit for bit
        latent = synth.normal("latent", tuple(g["mean"].shape), seed=9, std=3.0)
        np.testing.assert_array_equal(m.dequantize(latent, g["mean"]), g["q_latent"])


def test_latents_to_image_chain(O):
    """q_hyper_latent -> (mean, scale); symbols + mean -> q_latent; q_latent -> context pyramid -> 2-step
    decode: the whole decoder side on the GPU, checked stage by stage against the CPU restatement."""
    meta = json.load(open(os.path.join(GOLDEN, "manifest_hyperdec_full_x.json")))
    hman = [(k, tuple(v)) for k, v in meta["manifest"]]
    dmeta = json.load(open(os.path.join(GOLDEN, "manifest_ctxdec_full_x.json")))
    dman = [(k, tuple(v)) for k, v in dmeta["manifest"]]
    sd = {**synth.unet_state_dict(hman, seed=7), **synth.unet_state_dict(dman, seed=5)}
    m = cdc.ResnetCompressor(**meta["kwargs"])
    m.load_state_dict(sd)                                  # takes dec.* and hyper_dec.*
    qh = np.round(synth.normal("qh", (1, 256, 1, 2), seed=12, std=2.0)).astype(np.float32)
    mean, scale = m.hyper_decode(qh)
    rmean, rscale = om.hyper_decode(O, meta["dims"], sd, qh)
    assert relerr(mean, rmean) < TOL_FWD and relerr(scale, rscale) < TOL_FWD
    sym = np.round(synth.normal("sym", mean.shape, seed=13, std=2.0)).astype(np.float32)
    q_latent = m.dequantize(sym + rmean, rmean)            # integers + offset survive the round trip
    np.testing.assert_array_equal(q_latent, om.dequantize(sym + rmean, rmean))
    ctx = m.decode(q_latent)
    cfg = om.CompressorConfig(64, (4, 3, 2, 1), 64, 1)
    for a, r in zip(ctx, om.compressor_decode(O, cfg, sd, q_latent)):
        assert relerr(a, r) < TOL_FWD


def test_non_square_frame_against_double_accumulating_oracle():
    """A non-square frame (192 x 320, batch 3) of the full-width model against the restatement built with
    double accumulators: the HIP path sits an order of magnitude closer to it than fp32 CPU summation does
    (measured at 512 x 768: HIP 2.9e-6, fp32 restatement 1.6e-4), so this pins the kernels, not round-off."""
    kw, man, sd, _, _, _, _ = load_case("full_x")
    un = cdc.Unet(**kw)
    un.load_state_dict(sd)
    B, H, W = 3, 192, 320
    x = synth.normal("x", (B, 3, H, W), seed=1, std=0.8)
    ctx = synth.context_pyramid([64, 64, 128, 192], B, H, W, seed=3)
    t = np.full((B, 1), 0.41, np.float32)
    y = un(x, t, ctx)
    ref64 = om.unet_forward(oops.OrcOps("f64"), oracle_cfg(kw), sd, x, t, ctx)
    assert relerr(y, ref64) < 1e-5, relerr(y, ref64)


@pytest.mark.parametrize("name", ["hyperdec_small_x", "hyperdec_full_x", "hyperdec_full_eps"])
def test_rate_estimate_matches_reference_bpp(name):
    """cdc_bpp (FlexiblePrior + NormalDistribution likelihoods, -log2, sums) against the reference's own bpp()."""
    meta = json.load(open(os.path.join(GOLDEN, f"manifest_{name}.json")))
    g = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    if "mean" not in g.files:
        pytest.skip("digest-only fixture")
    man = [(k, tuple(v)) for k, v in meta["manifest"]]
    pman = [(k, tuple(v)) for k, v in meta["prior_manifest"]]
    psd = synth.unet_state_dict(pman, seed=11)
    sd = {**synth.unet_state_dict(man, seed=7),
          **{k: ((v * 2.0).astype(np.float32) if ".weight" in k else v) for k, v in psd.items()}}
    m = getattr(cdc, meta["class"])(**meta["kwargs"])
    m.load_hyper_state_dict(sd)
    b = m.rate(g["q_hyper_for_bpp"], g["q_latent_for_bpp"], g["mean"], g["scale"], tuple(g["img_hw"]))
    assert b.shape == g["bpp"].shape
    assert np.abs(b - g["bpp"]).max() <= 1e-4 * max(1.0, float(np.abs(g["bpp"]).max())), (b, g["bpp"])


# ---- encoder and the whole compressor forward ---------------------------------

def _symbols_close(a, ref, max_flip_frac=1e-4):
    """Quantised tensors: equal up to fp32 round-off except where a value sat within round-off of a rounding
    boundary (then it differs by exactly one quantisation step): allow a small fraction of such flips."""
    d = np.abs(a - ref)
    near = d <= 1.5e-5 * max(1.0, float(np.abs(ref).max()))
    flip = np.abs(d - 1.0) <= 1e-3
    assert (near | flip).all()
    assert flip.sum() <= max(1, int(max_flip_frac * flip.size)), (int(flip.sum()), flip.size)     # measured: 1 of 196 608 on the Kodak crops


@pytest.mark.parametrize("name", ["encoder_small_x", "encoder_full_x", "encoder_full_eps"])
def test_compressor_forward_matches_reference_golden(name):
    """Compressor.forward of the real reference -- analysis transform, hyper encoder, both quantisers, hyper
    decoder, rate estimate and the synthesis transform -- entirely through the C-ABI."""
    meta = json.load(open(os.path.join(GOLDEN, f"manifest_{name}.json")))
    man = [(k, tuple(v)) for k, v in meta["manifest"]]
    sd = synth.unet_state_dict(man, seed=15)
    m = getattr(cdc, meta["class"])(**meta["kwargs"])
    m.load_state_dict(sd)
    g = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    x = synth.normal("image", tuple(g["image_shape"]), seed=16, std=0.5).clip(-1, 1).astype(np.float32)
    latent, hyper = m.analysis(x)
    for key, a in (("latent", latent), ("hyper_latent", hyper)):
        assert list(a.shape) == list(g[f"{key}_shape"])
        assert relerr(a.reshape(-1)[g[f"{key}_idx"]], g[f"{key}_val"]) < TOL_FWD
        if key in g.files:
            assert relerr(a, g[key]) < TOL_FWD
    out = m(x)
    assert set(out) == {"output", "bpp", "q_latent", "q_hyper_latent"}
    for key in ("q_latent", "q_hyper_latent"):
        if key in g.files:
            _symbols_close(out[key], g[key])
    assert np.abs(out["bpp"] - g["bpp"]).max() <= 2e-3 * max(1.0, float(np.abs(g["bpp"]).max())), (out["bpp"], g["bpp"])
    assert len(out["output"]) == 4 and list(out["output"][0].shape) == list(g["ctx0_shape"])
    # the pyramid depends on q_latent: where no symbol flipped it matches the reference
    if "q_latent" in g.files and np.array_equal(out["q_latent"], g["q_latent"]):
        assert relerr(out["output"][3].reshape(-1)[g["ctx3_idx"]], g["ctx3_val"]) < TOL_FWD
        assert relerr(out["output"][0].reshape(-1)[g["ctx0_idx"]], g["ctx0_val"]) < TOL_FWD


def test_compress_end_to_end_without_reference_module():
    """GaussianDiffusion.compress with the GPU compressor as context_fn: images -> (reconstruction, bpp)."""
    meta = json.load(open(os.path.join(GOLDEN, "manifest_encoder_full_x.json")))
    man = [(k, tuple(v)) for k, v in meta["manifest"]]
    comp = cdc.ResnetCompressor(**meta["kwargs"])
    comp.load_state_dict(synth.unet_state_dict(man, seed=15))
    un, kw, usd, _, _, _, _ = make_unet("full_x")
    diff = cdc.GaussianDiffusionX(un, comp, None, num_timesteps=8193, pred_mode="x", var_schedule="cosine")
    x = synth.normal("image", (2, 3, 64, 128), seed=16, std=0.5).clip(-1, 1).astype(np.float32)
    init = synth.normal("init", x.shape, seed=1, std=0.8)
    rec, bpp = diff.compress(x, sample_steps=2, bpp_return_mean=False, init=init)
    assert rec.shape == x.shape and np.isfinite(rec).all() and np.abs(rec).max() <= 1.0 + 1e-6
    assert bpp.shape == (2,) and np.isfinite(bpp).all() and (bpp > 0).all()
    ctx = comp(x)["output"]
    np.testing.assert_array_equal(rec, diff.decompress(ctx, x.shape, sample_steps=2, init=init))


def test_kodak_crops_500_steps_match_reference():
    """BASELINE configs[0]: the reference's own CPU run of compress() on the three Kodak images it ships (256x256
    centre crops, 500 DDIM steps, synthetic parameters; tests/golden/make_golden.py::gen_kodak).
    (1) decode path alone -- the reference's q_latent -> context decoder -> 500-step decode: measured 8.4e-6 max
        abs difference on 256 sampled pixels;
    (2) whole compress() with the GPU compressor: bpp to 1e-5, at most a handful of symbols on a rounding
        boundary flip (measured: 1 of 196 608), the reconstruction follows except around a flipped symbol."""
    g = np.load(os.path.join(GOLDEN, "kodak_x_500.npz"))
    un, kw, usd, _, _, _, _ = make_unet("full_x")
    meta = json.load(open(os.path.join(GOLDEN, "manifest_encoder_full_x.json")))
    comp = cdc.ResnetCompressor(**meta["kwargs"])
    comp.load_state_dict(synth.unet_state_dict([(k, tuple(v)) for k, v in meta["manifest"]], seed=15))
    diff = cdc.GaussianDiffusionX(un, comp, None, num_timesteps=8193, pred_mode="x", var_schedule="cosine")
    x = (g["crops"].astype(np.float32).transpose(0, 3, 1, 2) / 255.0 * 2.0 - 1.0).astype(np.float32)
    init = synth.normal("init", x.shape, seed=1, std=0.8)
    steps = int(g["steps"])
    rec = diff.decompress(comp.decode(g["q_latent"]), x.shape, sample_steps=steps, init=init)
    d = np.abs(rec.reshape(-1)[g["rec_idx"]] - g["rec_val"])
    assert d.max() < 3e-5, float(d.max())          # measured 8.4e-6 (bf16x3) / ~1e-5 (fp16x2)
    assert abs(float(rec.astype(np.float64).sum()) - float(g["rec_sum"])) < 1e-5 * rec.size
    # north_star: "outputs within 1e-4 PSNR of the reference" -- the reference's own per-image PSNR (same definition:
    # 10 log10(4 / mse) on [-1, 1]) is part of the fixture
    psnr = np.array([10 * np.log10(4.0 / np.mean((rec[i].astype(np.float64) - x[i]) ** 2)) for i in range(3)])
    assert np.abs(psnr - g["psnr"]).max() <= 1e-4, (psnr, g["psnr"])
    # the default two-plane fp16 arithmetic did this on its own: no range fault, no silent fall-back behind a green test
    assert un.status() == {"arith": 1, "range_faults": 0, "nonfinite_results": 0} and comp.range_faults == 0
    rec2, bpp = diff.compress(x, sample_steps=steps, bpp_return_mean=False, init=init)
    assert np.abs(bpp - g["bpp"]).max() <= 1e-5 * float(np.abs(g["bpp"]).max())
    flipped = int((np.abs(comp(x)["q_latent"] - g["q_latent"]) > 0.5).sum())
    assert flipped <= 2, flipped          # measured: 1 of 196 608 (a latent within round-off of a rounding boundary)
    d2 = np.abs(rec2.reshape(-1)[g["rec_idx"]] - g["rec_val"])
    assert d2.mean() < (1e-5 if flipped == 0 else 5e-3), (flipped, float(d2.mean()))


def test_kodak_crops_eps_1000_steps_match_reference():
    """epsilon-param counterpart (BASELINE configs[2] step count): the reference's CPU run of
    compress(sample_mode="ddim") with BigCompressor on the same three crops, 1000 steps, no clipping.  The decode
    path alone (reference q_latent -> context decoder -> 1000 DDIM steps) must stay within 1e-3 of it."""
    path = os.path.join(GOLDEN, "kodak_eps_1000.npz")
    if not os.path.exists(path):
        pytest.skip("fixture not generated")
    g = np.load(path)
    crops = np.load(os.path.join(GOLDEN, "kodak_x_500.npz"))["crops"]
    un, kw, usd, _, _, _, _ = make_unet("full_eps")
    meta = json.load(open(os.path.join(GOLDEN, "manifest_encoder_full_eps.json")))
    comp = cdc.BigCompressor(**meta["kwargs"])
    comp.load_state_dict(synth.unet_state_dict([(k, tuple(v)) for k, v in meta["manifest"]], seed=15))
    diff = cdc.GaussianDiffusionEps(un, comp, num_timesteps=20000, clip_noise="none", pred_mode="noise",
                                    var_schedule="linear")
    x = (crops.astype(np.float32).transpose(0, 3, 1, 2) / 255.0 * 2.0 - 1.0).astype(np.float32)
    init = synth.normal("init", x.shape, seed=1, std=0.8)
    steps = int(g["steps"])
    rec = diff.decompress(comp.decode(g["q_latent"]), x.shape, sample_steps=steps, init=init)
    scale = max(1.0, float(np.abs(g["rec_val"]).max()))
    d = np.abs(rec.reshape(-1)[g["rec_idx"]] - g["rec_val"]) / scale
    assert d.max() < 1.5e-5, float(d.max())        # relative to |x| ~ 480; measured 4.4e-6
    out = comp(x)
    assert np.abs(out["bpp"] - g["bpp"]).max() <= 1e-5 * float(np.abs(g["bpp"]).max())
    assert int((np.abs(out["q_latent"] - g["q_latent"]) > 0.5).sum()) <= 2


def test_graph_replay_is_bit_identical_to_eager_launches(monkeypatch):
    """Small batches replay one captured DDIM iteration as a hipGraph (device-side step index): same kernels,
    same order -> the same bits as the eager loop."""
    un, kw, sd, x, time, ctx, g = make_unet("full_x")
    diff = cdc.GaussianDiffusionX(un, None, None, num_timesteps=8193, pred_mode="x", var_schedule="cosine")
    shape = x.shape
    init = synth.normal("init", shape, seed=1, std=0.8)
    monkeypatch.setenv("CDC_GRAPH", "0")
    eager = diff.decompress(ctx, shape, sample_steps=7, init=init)
    monkeypatch.setenv("CDC_GRAPH", "1")
    a = diff.decompress(ctx, shape, sample_steps=7, init=init)
    b = diff.decompress(ctx, shape, sample_steps=5, init=init)      # new schedule -> new capture
    monkeypatch.setenv("CDC_GRAPH", "0")
    np.testing.assert_array_equal(a, eager)
    np.testing.assert_array_equal(b, diff.decompress(ctx, shape, sample_steps=5, init=init))


# ---- round 2: the remaining BASELINE shapes, per-stage taps, sampler variants ---------------------------------

def _digest_check(a, g, key, tol=TOL_FWD):
    flat = a.reshape(-1)
    ref = g[key + "_val"]
    assert relerr(flat[g[key + "_idx"]], ref) <= tol
    assert abs(float(a.astype(np.float64).sum()) - float(g[key + "_sum"])) <= tol * a.size * max(1.0, float(np.abs(ref).max()))


def test_x_param_512_matches_reference_digest_and_batch16_rows():
    """BASELINE configs[4] (x-param, batch 16, 512x512): the real reference's B=1 forward and 4-step decode as
    digests (tests/golden/make_golden.py::gen_full_res_other), then every row of the batch-16 launch plans must
    reproduce the B=1 result."""
    g = np.load(os.path.join(GOLDEN, "full_res_x_512.npz"))
    kw, man, sd, _, _, _, _ = load_case("full_x")
    un = cdc.Unet(**kw)
    un.load_state_dict(sd)
    H = W = 512
    x = synth.normal("x", (1, 3, H, W), seed=1, std=0.8)
    ctx = synth.context_pyramid([64, 64, 128, 192], 1, H, W, seed=3)
    y1 = un(x, g["time"], ctx)
    _digest_check(y1, g, "y")
    diff = cdc.GaussianDiffusionX(un, None, None, num_timesteps=8193, pred_mode="x", var_schedule="cosine")
    init = synth.normal("init", (1, 3, H, W), seed=1, std=0.8)
    steps = int(g["steps"])
    rec1 = diff.decompress(ctx, (1, 3, H, W), sample_steps=steps, init=init)
    _digest_check(rec1, g, "dec", TOL_DEC)
    B = 16
    rec16 = diff.decompress([np.repeat(c, B, 0) for c in ctx], (B, 3, H, W), sample_steps=steps, init=np.repeat(init, B, 0))
    for k in (0, 7, 15):      # (different attention / split-K partitions: summation order differs; measured 1.1e-5)
        assert relerr(rec16[k], rec1[0]) < 3e-5, (k, relerr(rec16[k], rec1[0]))
    np.testing.assert_array_equal(rec16[3], rec16[12])


def test_eps_param_256_matches_reference_digest_and_batch32_rows():
    """BASELINE configs[2] (eps-param, batch 32, 256x256): reference digests at B=1 (forward + 4 DDIM steps, no
    clipping), then the batch-32 launch plans row by row (the eps model's level 0 has 6 -> 64 channels and a
    3-channel context: plans differ from the x-param ones)."""
    g = np.load(os.path.join(GOLDEN, "full_res_eps_256.npz"))
    kw, man, sd, _, _, _, _ = load_case("full_eps")
    un = cdc.Unet(**kw)
    un.load_state_dict(sd)
    H = W = 256
    x = synth.normal("x", (1, 3, H, W), seed=1, std=0.8)
    ctx = synth.context_pyramid([3, 64, 128, 192], 1, H, W, seed=3)
    y1 = un(x, g["time"], ctx)
    _digest_check(y1, g, "y")
    diff = cdc.GaussianDiffusionEps(un, None, num_timesteps=20000, clip_noise="none", pred_mode="noise", var_schedule="linear")
    init = synth.normal("init", (1, 3, H, W), seed=1, std=0.8)
    steps = int(g["steps"])
    rec1 = diff.decompress(ctx, (1, 3, H, W), sample_steps=steps, init=init)
    _digest_check(rec1, g, "dec", TOL_DEC)
    B = 32
    y32 = un(np.repeat(x, B, 0), np.repeat(g["time"], B, 0), [np.repeat(c, B, 0) for c in ctx])
    for k in (0, 13, 31):
        assert relerr(y32[k], y1[0]) < 5e-6, (k, relerr(y32[k], y1[0]))
    rec32 = diff.decompress([np.repeat(c, B, 0) for c in ctx], (B, 3, H, W), sample_steps=steps, init=np.repeat(init, B, 0))
    for k in (0, 19, 31):
        assert relerr(rec32[k], rec1[0]) < 3e-5, (k, relerr(rec32[k], rec1[0]))


def test_batch32_ddim_step_is_bit_reproducible():
    """The same DDIM step of BASELINE configs[1] (batch 32, 256x256) executed 200 times gives the same bits: the model path has no atomics
    and every summation order is fixed by the launch geometry.  (Round 5 found one launch that broke this about once in 10 000 executions on
    some boxes -- conv_pw_kernel's counted LDS-DMA wait, profiles/determinism_r05.txt; tools/determinism_stress_steps.py is the long form of
    this test with the localisation by taps.)"""
    import ctypes
    import torch
    kw, man, sd, _, _, _, _ = load_case("full_x")
    un = cdc.Unet(**kw)
    un.load_state_dict(sd)
    diff = cdc.GaussianDiffusionX(un, None, None, num_timesteps=8193, pred_mode="x", var_schedule="cosine")
    B, S, steps = 32, 256, 500
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(78)
    x = torch.randn((B, 3, S, S), generator=gen, device=dev) * 0.8
    ctx = [torch.randn((B, c, S >> l, S >> l), generator=gen, device=dev) * 0.5 for l, c in enumerate([64, 64, 128, 192])]
    diff.set_sample_schedule(steps)
    L, h = _lib.lib(), un._handle()
    ptrs = (ctypes.c_void_p * len(ctx))(*[c.data_ptr() for c in ctx])
    def step(i, out, with_ctx):
        _lib.check(h, L.cdc_ddim_step(h, x.data_ptr(), i, ptrs if with_ctx else None, len(ctx) if with_ctx else 0, None, 0.0, out.data_ptr(),
                                      B, S, S, diff._pred_flag(), diff._clip_flag(True), _lib.CDC_MEM_DEVICE, None))
    ref, out = torch.empty_like(x), torch.empty_like(x)
    step(250, ref, True)
    for k in range(200):
        step(250, out, False)
        assert torch.equal(out, ref), k


def test_configs1_full_length_batch32_rows_match_batch1_decodes():
    """BASELINE configs[1] at full length and full batch INSIDE pytest (bench.py's `verify` does the same
    outside it): x-param, batch 32 of DISTINCT images, 256x256, all 500 DDIM iterations on the batch-32 launch program
    (persistent kernels, fused plans), then rows 0 / 17 / 31 decoded on their own with the batch-1 program (other kernels,
    K splits and summation orders).  The chain is contractive (x0 is clamped each step): measured 8e-6."""
    import torch
    kw, man, sd, _, _, _, _ = load_case("full_x")
    un = cdc.Unet(**kw)
    un.load_state_dict(sd)
    diff = cdc.GaussianDiffusionX(un, None, None, num_timesteps=8193, pred_mode="x", var_schedule="cosine")
    B, S, steps = 32, 256, 500
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(77)
    init = torch.randn((B, 3, S, S), generator=gen, device=dev) * 0.8
    ctx = [torch.randn((B, c, S >> l, S >> l), generator=gen, device=dev) * 0.5 for l, c in enumerate([64, 64, 128, 192])]
    rec = diff.decompress(ctx, (B, 3, S, S), sample_steps=steps, init=init)
    assert bool(torch.isfinite(rec).all().item()) and float(rec.abs().max().item()) <= 1.0
    assert float((rec[0] - rec[17]).abs().max().item()) > 0.1            # the rows really are different images
    for k in (0, 17, 31):
        r1 = diff.decompress([c[k:k + 1] for c in ctx], (1, 3, S, S), sample_steps=steps, init=init[k:k + 1])
        e = relerr(r1[0].cpu().numpy(), rec[k].cpu().numpy())
        assert e < TOL_DEC, (k, e)
    assert un.status() == {"arith": 1, "range_faults": 0, "nonfinite_results": 0}


def _full_length_rows_against_batch1(diff, un, ctx_channels, B, S, steps, rows, seed):
    import torch
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(seed)
    init = torch.randn((B, 3, S, S), generator=gen, device=dev) * 0.8
    ctx = [torch.randn((B, c, S >> l, S >> l), generator=gen, device=dev) * 0.5 for l, c in enumerate(ctx_channels)]
    rec = diff.decompress(ctx, (B, 3, S, S), sample_steps=steps, init=init)
    assert bool(torch.isfinite(rec).all().item())
    assert float((rec[rows[0]] - rec[rows[1]]).abs().max().item()) > 0.05          # the rows really are different images
    for k in rows:
        r1 = diff.decompress([c[k:k + 1] for c in ctx], (1, 3, S, S), sample_steps=steps, init=init[k:k + 1])
        e = relerr(r1[0].cpu().numpy(), rec[k].cpu().numpy())
        assert e < TOL_DEC, (k, e)
    assert un.status() == {"arith": 1, "range_faults": 0, "nonfinite_results": 0}


def test_configs2_full_length_eps_batch32_rows_match_batch1_decodes():
    """BASELINE configs[2] at full length and full batch inside pytest (bench.py's other_configs does the same outside
    it): eps-param, batch 32 of distinct images, 256x256, all 1000 DDIM iterations without clipping on the batch-32 launch program, then
    rows 0 / 13 / 31 on their own with the batch-1 program.  Measured by bench.py's verify: 5.8e-6."""
    kw, man, sd, _, _, _, _ = load_case("full_eps")
    un = cdc.Unet(**kw)
    un.load_state_dict(sd)
    diff = cdc.GaussianDiffusionEps(un, None, num_timesteps=20000, clip_noise="none", pred_mode="noise", var_schedule="linear")
    _full_length_rows_against_batch1(diff, un, [3, 64, 128, 192], 32, 256, 1000, (0, 13, 31), 79)


def test_configs4_full_length_512_batch16_rows_match_batch1_decodes():
    """BASELINE configs[4] at full length and full batch inside pytest: x-param, batch 16 of distinct images, 512x512, all 500 DDIM
    iterations, rows 0 / 9 / 15 against batch-1 decodes.  Measured by bench.py's verify: 8.6e-6."""
    kw, man, sd, _, _, _, _ = load_case("full_x")
    un = cdc.Unet(**kw)
    un.load_state_dict(sd)
    diff = cdc.GaussianDiffusionX(un, None, None, num_timesteps=8193, pred_mode="x", var_schedule="cosine")
    _full_length_rows_against_batch1(diff, un, [64, 64, 128, 192], 16, 512, 500, (0, 9, 15), 80)


TAPS = ["downs.0.0", "downs.0.2", "downs.1.3", "mid_block1", "ups.0"]


def _op_labels(un):
    L, h = _lib.lib(), un._handle()
    labels = []
    for i in range(L.cdc_prof_num_ops(h)):
        lab, ms, n, fl = ctypes.c_char_p(), ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
        L.cdc_prof_op(h, i, ctypes.byref(lab), ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl))
        labels.append(lab.value.decode())
    return labels


def test_planes_only_tensor_reaching_an_fp32_reader_is_unpacked_not_a_build_error(monkeypatch):
    """a skip / Upsample output is made planes-only on SHAPE predictions of its readers' launch plans, taken
    before those readers are planned.  When a prediction misses -- here every decoder join is forced off the plane-operand kernels
    AFTER its two halves were made planes-only -- the program used to fail to build (CDC_ERR_UNSUPPORTED); it now unpacks
    h + l 2^-11 into the tensor's fp32 buffer once, ahead of that reader.  One 256 x 256 image through the full-width model (the
    128^2 and 64^2 skips and the Upsample halves of their joins are planes-only there): the forward with the missed predictions must
    equal the default program's (which the reference digests pin) to the rounding of the planes."""
    for k, v in {"CDC_PF_MIN_WAVES": "1", "CDC_PF_S2_MIN_WGS": "1", "CDC_PF_TZ_MIN_WGS": "1"}.items():
        monkeypatch.setenv(k, v)
    kw, man, sd, _, _, _, _ = load_case("full_x")
    S = 256
    x = synth.normal("x", (1, 3, S, S), seed=41, std=0.8)
    t = np.full((1, 1), 0.4, np.float32)
    ctx = [synth.normal(f"c{l}", (1, c, S >> l, S >> l), seed=42, std=0.5) for l, c in enumerate([64, 64, 128, 192])]
    un = cdc.Unet(**kw)
    un.load_state_dict(sd)
    y = un(x, t, ctx)
    assert not [l for l in _op_labels(un) if l == "pfunpack"]          # the predictions hold on the default plans: nothing to unpack
    monkeypatch.setenv("CDC_TEST_JOIN_MISS", "1")
    un2 = cdc.Unet(**kw)
    un2.load_state_dict(sd)
    y2 = un2(x, t, ctx)
    n_unpack = len([l for l in _op_labels(un2) if l == "pfunpack"])
    assert n_unpack >= 2, n_unpack                                     # skip halves and Upsample halves of the joins
    assert relerr(y2, y) < 5e-6, relerr(y2, y)
    got = un2.tap("ups.2")                                             # a planes-only Upsample output keeps its tap (unpacked on demand)
    assert got.ndim == 4 and np.isfinite(got).all()


@pytest.mark.parametrize("name", ["small_x", "small_eps", "odd_x", "full_x", "full_eps"])
def test_unet_stage_taps_match_reference(name):
    """Per-stage activations (forward hooks on the reference modules, stored by gen_unet as arrays for the small
    configurations and as digests for the full-width ones) against cdc_unet_tap of the same forward: a wrong
    ResnetBlock / attention / resampler shows up at its own stage, not only in the final output."""
    _check_taps(name)


def test_unet_stage_taps_of_a_planes_only_tensor(monkeypatch):
    """With the level-0 Downsample on the plane-operand kernel its input (downs.0.2, the attention output) exists as planes only:
    cdc_unet_tap unpacks h + l 2^-11."""
    monkeypatch.setenv("CDC_PF_S2_MIN_WGS", "1")
    monkeypatch.setenv("CDC_PF_MIN_WAVES", "1")       # ... and downs.0.0 (the first ResnetBlock's output, read by the second one only)
    _check_taps("full_x")
    _check_taps("full_eps")


def _check_taps(name):
    un, kw, sd, x, time, ctx, g = make_unet(name)
    un(x, time, ctx)
    checked = 0
    for k in TAPS:
        key = k
        if k == "downs.1.3" and f"tap_{k}" not in g.files and f"tap_{k}_val" not in g.files:
            continue
        if k == "downs.1.3" and len(kw["dim_mults"]) <= 2:
            key = "downs.0.3"                                  # (make_golden hooks downs[0][3] for two-level nets)
        if f"tap_{k}" in g.files:
            ref = g[f"tap_{k}"]
            got = un.tap(key)
            assert got.shape == ref.shape, (k, got.shape, ref.shape)
            assert relerr(got, ref) < TOL_FWD, (k, relerr(got, ref))
            checked += 1
        elif f"tap_{k}_val" in g.files:
            got = un.tap(key)
            ref = g[f"tap_{k}_val"]
            err = float(np.abs(got.reshape(-1)[g[f"tap_{k}_idx"]] - ref).max()) / max(1.0, float(np.abs(ref).max()))
            assert err < TOL_FWD, (k, err)
            assert abs(float(got.astype(np.float64).sum()) - float(g[f"tap_{k}_sum"])) < TOL_FWD * got.size * max(1.0, float(np.abs(ref).max()))
            checked += 1
    assert checked >= 4, checked


def test_sampler_variants_match_reference_golden():
    """x-tree pred_mode="noise" (the reference constructor's default) and eps-tree clip_noise="half" (ditto)."""
    g = np.load(os.path.join(GOLDEN, "decode_variants.npz"))
    un, kw, sd, x, time, ctx, _ = make_unet("small_x")
    init = synth.normal("init", x.shape, seed=1, std=0.8)
    diff = cdc.GaussianDiffusionX(un, None, None, num_timesteps=8193, pred_mode="noise", var_schedule="cosine")
    rec = diff.decompress(ctx, x.shape, sample_steps=3, init=init)
    assert relerr(rec, g["small_x"]) < TOL_DEC, relerr(rec, g["small_x"])
    un, kw, sd, x, time, ctx, _ = make_unet("small_eps")
    diff = cdc.GaussianDiffusionEps(un, None, num_timesteps=20000, clip_noise="half", pred_mode="noise", var_schedule="linear")
    rec = diff.decompress(ctx, x.shape, sample_steps=3, init=init)
    assert relerr(rec, g["small_eps"]) < TOL_DEC, relerr(rec, g["small_eps"])
    assert np.abs(rec[: x.shape[0] // 2]).max() <= np.abs(rec).max()
    # x-tree pred_mode="v" (xparam :128-139,161-162: predict_start_from_v), through cdc_decode and through cdc_ddim_step
    un, kw, sd, x, time, ctx, _ = make_unet("small_x")
    diff = cdc.GaussianDiffusionX(un, None, None, num_timesteps=8193, pred_mode="v", var_schedule="cosine")
    rec = diff.decompress(ctx, x.shape, sample_steps=3, init=init)
    assert relerr(rec, g["small_x_v"]) < TOL_DEC, relerr(rec, g["small_x_v"])
    h = un._handle()
    img = init.copy()
    out = np.empty_like(img)
    ptrs = (ctypes.c_void_p * len(ctx))(*[c.ctypes.data for c in ctx])
    for i in reversed(range(3)):
        _lib.check(h, _lib.lib().cdc_ddim_step(h, img.ctypes.data, i, ptrs, len(ctx), None, 0.0, out.ctypes.data, *x.shape[:1], *x.shape[2:],
                                               _lib.CDC_PRED_V, _lib.CDC_CLIP_ALL, _lib.CDC_MEM_HOST, None))
        img = out.copy()
    np.testing.assert_array_equal(img, rec)


@pytest.mark.parametrize("case", ["in_range", "overflow"])
def test_heavy_tailed_parameters_match_reference_in_both_arithmetics(case):
    """parity at a trained-weight-like dynamic range (per-channel gains x50, LayerNorm gains up to 10 with
    either sign, context magnitudes 1e-6 .. 300 / 3e4), against the REAL reference (tests/golden/make_golden.py::
    gen_heavy_tail).  "in_range": the default two-plane fp16 arithmetic has to carry it alone (no range fault);
    "overflow" (convolution inputs up to 7e11): the range guard has to hand the call to the three-plane bf16 arithmetic.
    Both are also run in CDC_ARITH_BF16X3 from the start."""
    from test_oracle import heavy_tail_case
    kw, sd, x, time, ctx, y_ref, rec_ref, biggest = heavy_tail_case(case)
    init = synth.normal("init", x.shape, seed=1, std=0.8)
    for arith in (1, 0):
        un = cdc.Unet(**kw)
        un.load_state_dict(sd)
        _lib.check(un._handle(), _lib.lib().cdc_set_arith(un._handle(), arith))
        y = un(x, time, ctx)
        assert relerr(y, y_ref) < 1e-5, (case, arith, relerr(y, y_ref))
        st = un.status()
        if arith == 1 and case == "in_range":
            assert biggest < 65504 and st == {"arith": 1, "range_faults": 0, "nonfinite_results": 0}, st
        if arith == 1 and case == "overflow":
            assert biggest > 65504 and st == {"arith": 0, "range_faults": 1, "nonfinite_results": 0}, st
        diff = cdc.GaussianDiffusionX(un, None, None, num_timesteps=8193, pred_mode="x", var_schedule="cosine")
        rec = diff.decompress(ctx, x.shape, sample_steps=3, init=init)
        # (|rec| <= 1: absolute = relative.  "overflow": activations up to 7e11 run through three DDIM steps -- measured 4.3e-5)
        assert relerr(rec, rec_ref) < (1.5e-4 if case == "overflow" else TOL_DEC), (case, arith, relerr(rec, rec_ref))


def test_fp16_range_overflow_falls_back_to_bf16_planes():
    """CDC_ARITH_F16X2 cannot represent |activation| >= 65504: the decode must notice (non-finite U-Net output flagged
    by the sampler kernel) and repeat itself in the three-plane bf16 arithmetic instead of returning garbage."""
    L = _lib.lib()
    un, kw, sd, x, time, ctx, _ = make_unet("small_eps")
    diff = cdc.GaussianDiffusionEps(un, None, num_timesteps=20000, clip_noise="none", pred_mode="noise", var_schedule="linear")
    init = synth.normal("init", x.shape, seed=1, std=0.8)
    big = [c * np.float32(3.0e5) for c in ctx]                  # context far outside the fp16 range
    assert L.cdc_get_arith(un._handle()) == 1
    rec = diff.decompress(big, x.shape, sample_steps=2, init=init)
    assert np.isfinite(rec).all()
    assert L.cdc_get_arith(un._handle()) == 0                   # the handle switched itself
    un2, *_ = make_unet("small_eps")
    _lib.check(un2._handle(), L.cdc_set_arith(un2._handle(), 0))
    diff2 = cdc.GaussianDiffusionEps(un2, None, num_timesteps=20000, clip_noise="none", pred_mode="noise", var_schedule="linear")
    ref = diff2.decompress(big, x.shape, sample_steps=2, init=init)
    np.testing.assert_array_equal(rec, ref)


OVERFLOW_BLOCK_CASES = [
    # (case, env) -- one per kernel family that can fuse / follow a Block convolution with LayerNorm + ReLU
    ((2, 64, 32, 32, 64, 3, 1, 1, True), {}),                                        # conv_split2_kernel, LayerNorm in the epilogue
    ((2, 64, 32, 32, 64, 3, 1, 1, True), {"CDC_PF": "1", "CDC_PF_MAXPIX": "0"}),     # conv_pf_kernel
    ((4, 64, 128, 256, 64, 3, 1, 1, True), {"CDC_PF": "1", "CDC_PF_MAXPIX": "0"}),   # conv_pf3_kernel (persistent)
    ((4, 320, 16, 16, 328, 3, 1, 1, True), {}),                                      # Cout % 32 != 0, C <= 384: convolution + ln_kernel_sliced
    ((1, 24, 24, 40, 24, 3, 1, 1, True), {}),                                        # Cout % 32 != 0: convolution + ln_kernel
    ((1, 64, 36, 32, 192, 1, 1, 0, True), {}),                                       # 1x1 (conv_pw_kernel / split2) + LayerNorm
]


@pytest.mark.parametrize("case,env", OVERFLOW_BLOCK_CASES)
def test_block_conv_reports_fp16_overflow_where_layernorm_relu_would_hide_it(O, case, env, monkeypatch):
    """ONE activation outside the fp16 range feeding conv -> LayerNorm -> ReLU (+ residual).  In F16X2 the
    accumulators of that pixel neighbourhood become inf / NaN, the LayerNorm turns them into NaN and a ReLU written as
    max(NaN, 0) returns 0 -- finite and wrong, invisible to a check of the result.  Every epilogue therefore reports
    non-finite accumulators BEFORE its LayerNorm / ReLU (ConvArgs::fault); the call must come back as a range fault,
    repeated in BF16X3, and match the oracle."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    B, Ci, H, W, Co, k, s, p, _ = case
    x = synth.normal("cx", (B, Ci, H, W), 21)
    x[B - 1, Ci // 2, H // 2, W // 3] = np.float32(1.0e5)            # > 65504
    w = synth.normal("cw", (Co, Ci, k, k), 21, 1.0 / np.sqrt(Ci * k * k))
    b = synth.normal("cb", (Co,), 21, 0.1)
    g = synth.normal("cg", (Co,), 21, 0.2, 1.0)
    bb = synth.normal("cbb", (Co,), 21, 0.2)
    conv = O.conv2d(x, w, b, s, p)
    resid = synth.normal("cr", conv.shape, 21)
    ref = np.maximum(O.chan_layernorm(conv, g, bb), 0) + resid
    G = Ops(0)
    assert G.status() == {"arith": 1, "range_faults": 0, "nonfinite_results": 0}
    got = G.conv2d(x, w, b, s, p, ln_g=g, ln_b=bb, relu=True, resid=resid)
    assert G.status() == {"arith": 0, "range_faults": 1, "nonfinite_results": 0}, G.status()
    assert np.isfinite(got).all()
    assert relerr(got, ref) < 1e-5, relerr(got, ref)
    # the same handle, in range again: stays in the full-range arithmetic, no further fault
    x2 = synth.normal("cx", (B, Ci, H, W), 21)
    got2 = G.conv2d(x2, w, b, s, p, ln_g=g, ln_b=bb, relu=True, resid=resid)
    ref2 = np.maximum(O.chan_layernorm(O.conv2d(x2, w, b, s, p), g, bb), 0) + resid
    assert relerr(got2, ref2) < 1e-5 and G.status()["range_faults"] == 1


def test_overflow_of_a_block_input_only_is_detected_inside_the_network(O):
    """The in-network form of the case above: a time-embedding bias of 2e5 on one channel makes h1 = Block1(x) + mlp(t) leave
    the fp16 range; h1 feeds block2 and nothing else (network_components.py:107-114), so the overflow sits ONLY in the
    input of a conv -> LayerNorm -> ReLU and the residual path keeps every tensor finite.  The forward must report a
    range fault and agree with the oracle (before round 4 this relied on a later linear layer meeting the value)."""
    un, kw, sd, x, time, ctx, _ = make_unet("small_x")
    sd = {k: v.copy() for k, v in sd.items()}
    key = "downs.1.1.mlp.1.bias"
    sd[key][3] = np.float32(2.0e5)
    un = cdc.Unet(**kw)
    un.load_state_dict(sd)
    ref = om.unet_forward(O, oracle_cfg(kw), sd, x, time, ctx)
    assert np.isfinite(ref).all()
    y = un(x, time, ctx)
    assert un.status() == {"arith": 0, "range_faults": 1, "nonfinite_results": 0}, un.status()
    assert relerr(y, ref) < 1e-5, relerr(y, ref)


def _small_compressor():
    meta = json.load(open(os.path.join(GOLDEN, "manifest_encoder_small_x.json")))
    comp = cdc.ResnetCompressor(**meta["kwargs"])
    comp.load_state_dict(synth.unet_state_dict([(k, tuple(v)) for k, v in meta["manifest"]], seed=15))
    return comp


def test_range_guard_in_unet_forward_and_ddim_step():
    """VERDICT r2 / ADVICE r2: the fp16 range guard lives in EVERY entry point, not only in cdc_decode.  Inputs far outside
    the fp16 range: the F16X2 call notices its non-finite result, repeats itself in BF16X3, says so through the counters,
    and returns exactly what a handle that was in BF16X3 from the start returns."""
    L = _lib.lib()
    un, kw, sd, x, time, ctx, _ = make_unet("small_x")
    big = [c * np.float32(3.0e5) for c in ctx]
    assert un.status() == {"arith": 1, "range_faults": 0, "nonfinite_results": 0}
    y = un(x, time, big)
    assert np.isfinite(y).all()
    assert un.status() == {"arith": 0, "range_faults": 1, "nonfinite_results": 0}
    un2, *_ = make_unet("small_x")
    _lib.check(un2._handle(), L.cdc_set_arith(un2._handle(), 0))
    np.testing.assert_array_equal(y, un2(x, time, big))
    assert un2.range_faults == 0
    # a non-finite INPUT stays non-finite in the full-range arithmetic: returned as the reference would, and counted
    bad = [c.copy() for c in ctx]
    bad[0][0, 0, 0, 0] = np.nan
    un3, *_ = make_unet("small_x")
    y3 = un3(x, time, bad)
    assert not np.isfinite(y3).all()
    # (and since the repetition was non-finite too, the range was not the cause: the handle keeps its fast arithmetic)
    assert un3.status() == {"arith": 1, "range_faults": 0, "nonfinite_results": 1}
    y3b = un3(x, time, ctx)                                     # ... and still works
    assert np.isfinite(y3b).all() and un3.status()["arith"] == 1
    # per-step sampler entry point (eta != 0 takes cdc_ddim_step)
    un4, *_ = make_unet("small_x")
    diff = cdc.GaussianDiffusionX(un4, None, None, num_timesteps=8193, pred_mode="x", var_schedule="cosine")
    init = synth.normal("init", x.shape, seed=1, std=0.8)
    np.random.seed(3)
    rec = diff.decompress(big, x.shape, sample_steps=2, init=init, eta=0.5)
    assert np.isfinite(rec).all() and un4.status()["arith"] == 0 and un4.range_faults == 1


def test_range_guard_in_the_compressor_entry_points():
    """cdc_encoder_encode, cdc_hyperdec_decode, cdc_ctxdec_decode (and through them compress_to_bytes): same contract."""
    L = _lib.lib()
    comp, ref = _small_compressor(), _small_compressor()
    img = synth.normal("img", (1, 3, 64, 64), seed=9, std=0.5)
    for c in (ref,):
        c(img)                                                   # create the handles ...
        for hnd in (c._h, c._hh, c._eh):
            if hnd is not None:
                _lib.check(hnd, L.cdc_set_arith(hnd, 0))         # ... and put them in the full-range arithmetic
    huge = img * np.float32(2.0e6)
    lat, hyp = comp.analysis(huge)
    assert np.isfinite(lat).all() and np.isfinite(hyp).all()
    assert comp.status()["enc"] == {"arith": 0, "range_faults": 1, "nonfinite_results": 0}
    lat_r, hyp_r = ref.analysis(huge)
    np.testing.assert_array_equal(lat, lat_r)
    np.testing.assert_array_equal(hyp, hyp_r)
    qh = np.rint(hyp_r) * np.float32(1.0e4)
    mean, scale = comp.hyper_decode(qh)
    assert np.isfinite(mean).all() and np.isfinite(scale).all()
    assert comp.status()["hyper_dec"]["arith"] == 0 and comp.status()["hyper_dec"]["range_faults"] == 1
    mr, sr = ref.hyper_decode(qh)
    np.testing.assert_array_equal(mean, mr)
    np.testing.assert_array_equal(scale, sr)
    ql = np.rint(lat_r) * np.float32(1.0e3)
    outs = comp.decode(ql)
    assert all(np.isfinite(o).all() for o in outs)
    assert comp.status()["dec"]["arith"] == 0 and comp.status()["dec"]["range_faults"] == 1
    for a, b in zip(outs, ref.decode(ql)):
        np.testing.assert_array_equal(a, b)
    assert ref.range_faults == 0
    # the entropy encoder refuses what it cannot code instead of converting NaN to an integer
    comp2 = _small_compressor()
    with pytest.raises(_lib.CdcError, match="non-finite"):
        comp2.compress_to_bytes(img * np.float32(np.nan))
    # ... and garbage input is refused BEFORE hyper_dec runs: it must not cost the hyper-decoder handle its fast arithmetic
    assert comp2.status()["hyper_dec"] == {"arith": 1, "range_faults": 0, "nonfinite_results": 0}


def test_model_follows_device_change_after_load():
    """load_state_dict() followed by .to(device) must leave a usable model (both reference test
    scripts use that order): the handle is re-created, the parameters replayed AND finalized again."""
    un, kw, sd, x, time, ctx, g = make_unet("small_x")
    y0 = un(x, time, ctx)
    un.to(0)                          # same device: nothing happens
    un._h and _lib.lib().cdc_destroy(un._h)
    un._h, un._finalized = None, False          # what .to(other device) does
    y1 = un(x, time, ctx) if (un._handle() and un._finalized) else None
    assert y1 is not None
    np.testing.assert_array_equal(y0, y1)
    meta = json.load(open(os.path.join(GOLDEN, "manifest_encoder_small_x.json")))
    comp = cdc.ResnetCompressor(**meta["kwargs"])
    csd = synth.unet_state_dict([(k, tuple(v)) for k, v in meta["manifest"]], seed=15)
    comp.load_state_dict(csd)
    img = synth.normal("img", (1, 3, 64, 64), seed=9, std=0.5)
    a = comp(img)
    comp.device_index = -1            # force the "device changed" branch of .to()
    comp.to(0)
    b = comp(img)
    np.testing.assert_array_equal(a["q_latent"], b["q_latent"])
    np.testing.assert_array_equal(a["bpp"], b["bpp"])


def test_bench_under_torchrun_one_rank_nccl(tmp_path):
    """bench.py's multi-GPU path on the one GPU a test box has: torchrun --nproc-per-node 1 initialises the RCCL
    ("nccl") process group, the decode goes through parallel.sharded_decode (shard of a global batch + all_gather),
    and the line must report the rank count it saw and a verified output."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29611", os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1",
           "--warmup", "0", "--batch", "3", "--sample-steps", "6", "--size", "64", "--no-cpu-baseline", "--prof-every", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["config"]["rccl_ranks_seen"] == 1
    assert d["config"]["global_batch"] == 3 and d["config"]["finite"] is True
    assert d["value"] > 0 and d["roofline"]["achieved"] >= 0


def test_bench_multi_rank_branch_two_ranks_on_one_gpu():
    """the world > 1 branch of bench.py (per-rank seeds, shard of a global batch, all_gather of the decoded
    images, MAX-over-ranks timing, rank count, per-rank verification) on the one GPU of a test box: torchrun starts TWO ranks
    that share cuda:0 and the collectives run over gloo (RCCL refuses two ranks on one device).  The 8-GPU command differs
    only in `--backend nccl` (the default) and one GPU per rank."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != "HSA_ENABLE_IPC_MODE_LEGACY"}    # bench.py sets it itself
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29613", os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2",
           "--warmup", "1", "--batch", "3", "--sample-steps", "6", "--size", "64", "--no-cpu-baseline", "--prof-every", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                       # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["rccl_ranks_seen"] == 2 and d["config"]["backend"] == "gloo"
    assert d["config"]["global_batch"] == 6 and d["config"]["batch_per_gpu"] == 3 and d["steps"] == 2 and d["warmup"] == 1
    assert d["verify"]["ok"] is True and d["config"]["finite"] is True
    assert d["scaling"] == "weak" and abs(d["value"] - 6 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    assert "cpu_baseline" not in d and "other_configs" not in d   # N = 1 legs only


def test_probes_measure_the_parts_own_ceilings():
    """bench.py re-measures what the box sustains (csrc/probe.hip) instead of quoting committed literals.  The
    figures must be physical: the register-only MFMA loop between a fifth of and just above the nominal 2.5 PFLOP/s, the float4 copy between 1 and 8 TB/s."""
    L = _lib.lib()
    tf_r, tf_c, gbs = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    assert L.cdc_probe_mfma_f16(0, 1, 20000, ctypes.byref(tf_r)) == 0
    assert L.cdc_probe_mfma_f16(0, 0, 20000, ctypes.byref(tf_c)) == 0
    assert L.cdc_probe_hbm_copy(0, 1 << 28, 3, ctypes.byref(gbs)) == 0
    assert 500.0 < tf_r.value < 2700.0 and 500.0 < tf_c.value < 2700.0, (tf_r.value, tf_c.value)
    assert 1000.0 < gbs.value < 8200.0, gbs.value
    assert L.cdc_probe_mfma_f16(0, 1, 0, ctypes.byref(tf_r)) != 0 and L.cdc_probe_hbm_copy(0, 16, 1, ctypes.byref(gbs)) != 0   # bad arguments


def test_bench_self_launch_two_ranks_on_one_gpu():
    """the driver's form of a multi-GPU run is the BARE command `python bench.py --gpus N ...` -- no
    torch.distributed.run on the command line, no WORLD_SIZE in the environment.  bench.py then starts its N ranks itself
    (self_launch: one child per GPU, rendezvous on 127.0.0.1) and rank 0 prints the one JSON line.  Two ranks share cuda:0 over
    gloo here; on an 8-GPU node the same command without --backend runs one rank per GPU over RCCL."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("HSA_ENABLE_IPC_MODE_LEGACY", "RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1",
           "--batch", "3", "--sample-steps", "6", "--size", "64", "--no-cpu-baseline", "--prof-every", "2"]
    assert "torch.distributed.run" not in " ".join(cmd)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                       # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["rccl_ranks_seen"] == 2 and d["config"]["launcher"] == "self"
    assert d["config"]["global_batch"] == 6 and d["config"]["batch_per_gpu"] == 3 and d["steps"] == 2 and d["warmup"] == 1
    assert d["verify"]["ok"] is True and d["config"]["finite"] is True
    assert d["scaling"] == "weak" and abs(d["value"] - 6 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]


def test_inference_script_counterpart_on_kodak_crops(tmp_path):
    """examples/test_xparam.py = the reference's test_xparam.py on this path: argument set, EMA checkpoint layout
    ("ema_model." prefix, wrapper entries ignored), uint8/255*2-1 scaling, printed bpp.  Driven on two of the Kodak
    crops with the fixture's parameters stored as a checkpoint file; the printed bpp must be the reference's."""
    import subprocess
    import sys
    import torch
    from PIL import Image
    g = np.load(os.path.join(GOLDEN, "kodak_x_500.npz"))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    kw, man, usd, _, _, _, _ = load_case("full_x")
    meta = json.load(open(os.path.join(GOLDEN, "manifest_encoder_full_x.json")))
    csd = synth.unet_state_dict([(k, tuple(v)) for k, v in meta["manifest"]], seed=15)
    ema = {"initted": torch.tensor(True), "step": torch.tensor(12345)}
    for k, v in usd.items():
        ema["ema_model.denoise_fn." + k] = torch.from_numpy(np.asarray(v))
        ema["online_model.denoise_fn." + k] = torch.zeros(1)                 # wrapper entries: must be ignored
    for k, v in csd.items():
        ema["ema_model.context_fn." + k] = torch.from_numpy(np.asarray(v))
    ema["ema_model.train_betas"] = torch.zeros(8193)                          # derived buffers: ignored
    ckpt = tmp_path / "ckpt.pt"
    torch.save({"step": 1, "ema": ema}, ckpt)
    imgs, outs = tmp_path / "imgs", tmp_path / "out"
    imgs.mkdir()
    for i in range(2):
        Image.fromarray(g["crops"][i]).save(imgs / f"{i}.png")
    cmd = [sys.executable, os.path.join(root, "examples", "test_xparam.py"), "--ckpt", str(ckpt), "--lpips_weight", "0.0",
           "--n_denoise_step", "6", "--img_dir", str(imgs), "--out_dir", str(outs), "--seed", "3"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    bpps = [float(l.split("bpp:")[1].strip().strip("tensor()").split(",")[0].strip("[]() "))
            for l in r.stdout.splitlines() if l.startswith("bpp:")]
    assert len(bpps) == 2
    for i in range(2):
        assert abs(bpps[i] - float(g["bpp"][i])) <= 1e-5 * float(g["bpp"][i]), (i, bpps[i], float(g["bpp"][i]))
        out = np.asarray(Image.open(outs / f"{i}.png"))
        assert out.shape == (256, 256, 3) and out.dtype == np.uint8
    r2 = subprocess.run([sys.executable, os.path.join(root, "examples", "test_epsilonparam.py"), "--ckpt", "synthetic",
                         "--lpips_weight", "0.0", "--n_denoise_step", "4", "--img_dir", str(imgs), "--out_dir",
                         str(tmp_path / "out_eps"), "--seed", "3"], capture_output=True, text=True, timeout=900, cwd=root)
    assert r2.returncode == 0, r2.stderr[-2000:]
    assert len([l for l in r2.stdout.splitlines() if l.startswith("bpp:")]) == 2


@pytest.mark.gpu
def test_persistent_kernel_variants_in_the_network_match_batch1_plans():
    """BASELINE configs[1] size: at batch 32 / 256 x 256 the 64- and 128-channel 3x3 layers run on conv_pf3_kernel (every
    epilogue variant, the 3-channel res_conv of downs.0.0 included); one image at a time none of them does.  Rows of the
    batch-32 forward must equal the batch-1 forwards to fp32 round-off, and the launch program must really contain the
    persistent kernel."""
    import torch
    import cdc_compression_amd as cdc
    from cdc_compression_amd import _lib
    kw, man, sd, _, _, _, _ = load_case("full_x")
    un = cdc.Unet(**kw)
    un.load_state_dict(sd)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(11)
    B, S = 32, 256
    x = torch.randn((B, 3, S, S), generator=g, device=dev) * 0.8
    t = torch.full((B, 1), 0.37, device=dev)
    ctx = [torch.randn((B, c, S >> l, S >> l), generator=g, device=dev) * 0.5 for l, c in enumerate([64, 64, 128, 192])]
    L, h = _lib.lib(), un._handle()
    L.cdc_prof_reset(h)
    L.cdc_prof_enable(h, 1)
    y = un(x, t, ctx).clone()
    labels = []
    for i in range(L.cdc_prof_num_ops(h)):
        lab, ms, n, fl = ctypes.c_char_p(), ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
        L.cdc_prof_op(h, i, ctypes.byref(lab), ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl))
        labels.append(lab.value.decode())
    L.cdc_prof_enable(h, 0)
    pf3 = [l for l in labels if " PF3 " in l]
    assert len(pf3) >= 10 and any("64->64" in l and "+pf +res" in l and "256x256" in l for l in pf3), pf3
    assert bool(torch.isfinite(y).all())
    for b in (0, 17, 31):
        yb = un(x[b:b + 1], t[b:b + 1], [c[b:b + 1].contiguous() for c in ctx])
        err = float((yb - y[b:b + 1]).abs().max() / y[b:b + 1].abs().max())
        assert err < 2e-5, (b, err)
