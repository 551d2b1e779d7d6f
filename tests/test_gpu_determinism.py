"""Long-form determinism guard of the model path's kernels.

The model path has no atomics and every summation order is fixed by the launch geometry, so a launch program must reproduce its
own bits.  Round 5 found one kernel (conv_pw_kernel with a counted LDS-DMA wait) that broke this about once in 10 000 launches on
some boxes; a 200-repeat test cannot see such a rate.  Here every LDS-DMA kernel family runs >= 20 000 executions of a small layer
through the C-ABI's stress mode (include/cdc_hip.h: cdc_op_stress -- the program is re-launched on the device and every result is
compared bitwise with the first execution's; no host round trip per launch), a few seconds per family.
"""
import numpy as np
import pytest

from cdc_compression_amd import synth

pytestmark = pytest.mark.gpu

REPEATS = 20000

PF_ENV = {"CDC_PF": "1", "CDC_PF_MAXPIX": "0", "CDC_WS_MIN_WGS": "1000000000", "CDC_PF_MIN_WAVES": "1"}     # (never the weight-stationary kernel; small launches stay on the plane kernels)
# (label, env, kind, case)   conv: (B, Cin, H, W, Cout, k, stride, pad, fused)   convT: (B, Cin, H, W, Cout)
FAMILIES = [
    ("conv_pf3_kernel<2,2,1,4> 3x3 64ch persistent ping-pong", PF_ENV, "conv", (4, 64, 128, 256, 64, 3, 1, 1, True)),
    ("conv_pf3_kernel<2,2,2,2> 3x3 128ch persistent ping-pong", PF_ENV, "conv", (4, 128, 64, 256, 128, 3, 1, 1, True)),
    ("conv_pf_kernel 3x3 192ch (eight waves)", {**PF_ENV, "CDC_OP_REQUIRE_PF": "1"}, "conv", (1, 192, 36, 32, 192, 3, 1, 1, True)),
    ("conv_pf_kernel 3x3 256ch", {**PF_ENV, "CDC_OP_REQUIRE_PF": "1"}, "conv", (1, 256, 32, 32, 256, 3, 1, 1, True)),
    ("conv_pf_kernel 1x1 384->128", {**PF_ENV, "CDC_OP_REQUIRE_PF": "1"}, "conv", (2, 384, 32, 32, 128, 1, 1, 0, False)),
    ("conv_pf_kernel STR=2 (Downsample)", {**PF_ENV, "CDC_PF_S2_MIN_WGS": "1", "CDC_OP_REQUIRE_PF": "1"}, "conv", (2, 64, 64, 128, 64, 3, 2, 1, False)),
    ("conv_pf_kernel TZ=4 (Upsample, fused phases)", {**PF_ENV, "CDC_PF_TZ_MIN_WGS": "1", "CDC_OP_REQUIRE_PF": "1"}, "convT", (2, 64, 32, 64, 64)),
    ("conv_pw_kernel 16-byte activation pieces, three channel groups", {"CDC_PW_MIN_WAVES": "1"}, "conv", (2, 192, 64, 64, 384, 1, 1, 0, False)),
    ("conv_pw_kernel 16-byte activation pieces, 256->256", {"CDC_PW_MIN_WAVES": "1"}, "conv", (2, 256, 32, 32, 256, 1, 1, 0, False)),
    ("conv_pw_kernel 4-byte activation pieces", {"CDC_PW_MIN_WAVES": "1", "CDC_NO_PW_X16": "1"}, "conv", (1, 64, 40, 96, 384, 1, 1, 0, False)),
    ("conv_split2_kernel (weights by LDS-DMA)", {}, "conv", (2, 128, 16, 16, 128, 3, 1, 1, True)),
    ("conv_ws_kernel (few-pixel level)", {"CDC_WS_MIN_WGS": "1", "CDC_OP_REQUIRE_WS": "1"}, "conv", (2, 384, 8, 8, 384, 3, 1, 1, True)),
]


@pytest.mark.parametrize("label,env,kind,case", FAMILIES, ids=[f[0].split(" ")[0] + "-" + str(i) for i, f in enumerate(FAMILIES)])
def test_kernel_family_reproduces_its_bits_over_20000_launches(label, env, kind, case, monkeypatch):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    from cdc_compression_amd.ops import Ops
    G = Ops(0)
    G.stress(REPEATS)
    if kind == "conv":
        B, Ci, H, W, Co, k, s, p, fused = case
        x = synth.normal("dx", (B, Ci, H, W), 31)
        w = synth.normal("dw", (Co, Ci, k, k), 31, 1.0 / np.sqrt(Ci * k * k))
        b = synth.normal("db", (Co,), 31, 0.1)
        if fused:
            Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
            y = G.conv2d(x, w, b, s, p, ln_g=synth.normal("dg", (Co,), 31, 0.2, 1.0), ln_b=synth.normal("dbb", (Co,), 31, 0.2), relu=True,
                         shift=synth.normal("ds", (B, Co), 31, 0.3), resid=synth.normal("dr", (B, Co, Ho, Wo), 31))
        else:
            y = G.conv2d(x, w, b, s, p)
    else:
        B, Ci, H, W, Co = case
        y = G.conv_transpose2d(synth.normal("dx", (B, Ci, H, W), 31), synth.normal("dw", (Ci, Co, 4, 4), 31, 1.0 / np.sqrt(Ci * 4)),
                               synth.normal("db", (Co,), 31, 0.1))
    n, differing = G.stress_result()
    assert np.isfinite(y).all()
    assert n == REPEATS, (label, n)
    assert differing == 0, f"{label}: {differing} of {n} executions differ bitwise from the first"
