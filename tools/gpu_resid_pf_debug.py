"""Development aid: where does the planes-residual program first differ from the fp32-residual one?  (taps of one forward, 256x256)"""
import os, sys, subprocess, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def run(flag):
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, %r)
import cdc_compression_amd as cdc
from cdc_compression_amd import synth
kw = dict(dim=64, channels=3, context_channels=64, dim_mults=(1, 2, 3, 4, 5, 6), context_dim_mults=(1, 2, 3, 4))
un = cdc.Unet(**kw)
sd = synth.unet_state_dict(un.manifest(), seed=3)
un.load_state_dict(sd)
B, H = int(os.environ.get("DBG_B", "1")), 256
x = synth.normal("x", (B, 3, H, H), 5)
t = np.full((B,), 0.3, np.float32)
ctx = synth.context_pyramid([64 * m for m in (1, 2, 3, 4)], B, H, H)
y = un(x, t, ctx)
out = {"y": y}
for k in ["downs.0.0", "downs.0.1", "downs.0.2", "downs.0.3", "downs.1.0", "downs.1.1", "downs.2.0", "downs.2.1", "mid_block1", "ups.0", "ups.3"]:
    try: out[k] = un.tap(k)
    except Exception as e: out[k] = np.zeros(1, np.float32)
np.savez(%r, **out)
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "/tmp/dbg_%s.npz" % flag)
    env = dict(os.environ, CDC_DEV="1")
    if flag == "old": env["CDC_NO_RESID_PF"] = "1"
    subprocess.run([sys.executable, "-c", code], env=env, check=True)
    return np.load("/tmp/dbg_%s.npz" % flag)

a, b = run("new"), run("old")
for k in a.files:
    if a[k].shape != b[k].shape: print(k, "shape", a[k].shape, b[k].shape); continue
    d = float(np.abs(a[k] - b[k]).max()) / max(1.0, float(np.abs(b[k]).max()))
    print("%-12s rel diff %.3e   (max |old| %.3g)" % (k, d, float(np.abs(b[k]).max())))
