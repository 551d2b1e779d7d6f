"""Helper of tests/test_gpu_parity.py (separate processes: some switches are read once): one U-Net forward with the round-4 plane-operand
forms and the round-5 few-pixel kernels on (default) against all of them off, on frame sizes whose tiles are ragged -- the two programs
must agree to rounding."""
import os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))     # (the repository root: this file lives in tests/)
OFF = {"CDC_NO_PF_S2": "1", "CDC_NO_PF_TZ": "1", "CDC_NO_PF_17": "1", "CDC_NO_PF_UF": "1", "CDC_NO_RESID_PF": "1", "CDC_NO_PF_SKIP_PLANES": "1",
       "CDC_NO_RESID2": "1", "CDC_NO_CTX_ONE": "1", "CDC_WS_MIN_WGS": "100000000", "CDC_WS1_MIN_WGS": "100000000"}

def run(tag, B, H, W, off):
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, %r)
import cdc_compression_amd as cdc
from cdc_compression_amd import synth
kw = dict(dim=64, channels=3, context_channels=64, dim_mults=(1, 2, 3, 4, 5, 6), context_dim_mults=(1, 2, 3, 4))
un = cdc.Unet(**kw)
un.load_state_dict(synth.unet_state_dict(un.manifest(), seed=3))
B, H, W = %d, %d, %d
x = synth.normal("x", (B, 3, H, W), 5)
t = np.linspace(0.1, 0.9, B).astype(np.float32)
ctx = [synth.normal("ctx%%d" %% l, (B, c, H >> l, W >> l), 3, 0.5) for l, c in enumerate([64, 128, 192, 256])]
y = un(x, t, ctx)
np.save(%r, y)
print(un.status())
''' % (ROOT, B, H, W, "/tmp/ab_%s.npy" % tag)
    env = dict(os.environ, CDC_DEV="1")
    if off:
        env.update(OFF)
    subprocess.run([sys.executable, "-c", code], env=env, check=True)
    return np.load("/tmp/ab_%s.npy" % tag)

worst = 0.0
SHAPES = [(4, 160, 224), (2, 288, 352), (16, 64, 96), (8, 256, 256), (3, 320, 192)]
if len(sys.argv) > 1:                       # e.g. "4x160x224 3x320x192"
    SHAPES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for (B, H, W) in SHAPES:
    a, b = run("on", B, H, W, False), run("off", B, H, W, True)
    d = float(np.abs(a - b).max()) / max(1.0, float(np.abs(b).max()))
    worst = max(worst, d)
    print("B=%d %dx%d: rel diff %.3e, finite %s" % (B, H, W, d, bool(np.isfinite(a).all())))
assert worst < 1e-5, worst
print("ok")
