#!/usr/bin/env python3
"""Development aid: full-width x-param U-Net forward at a Kodak-sized frame (512 x 768, B = 1) and an odd batch
at 320 x 448 (B = 3), checked against the CPU restatement."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cdc_compression_amd as cdc
from cdc_compression_amd import synth
from oracle import model as om, ops as oops

KW = dict(dim=64, channels=3, context_channels=64, dim_mults=(1, 2, 3, 4, 5, 6), context_dim_mults=(1, 2, 3, 4))
un = cdc.Unet(**KW)
sd = synth.unet_state_dict(un.manifest(), seed=0)
un.load_state_dict(sd)
O = oops.OrcOps("f32")
for (B, H, W) in [(1, 512, 768), (3, 320, 448)]:
    x = synth.normal("x", (B, 3, H, W), seed=1, std=0.8)
    ctx = synth.context_pyramid([64, 64, 128, 192], B, H, W, seed=3)
    t = np.full((B, 1), 0.41, np.float32)
    y = un(x, t, ctx)
    t0 = time.time()
    ref = om.unet_forward(O, om.UnetConfig(**KW), sd, x, t, ctx)
    ref64 = om.unet_forward(oops.OrcOps("f64"), om.UnetConfig(**KW), sd, x, t, ctx)
    sc = max(1.0, np.abs(ref64).max())
    err = float(np.abs(y - ref).max() / sc)
    print(f"B={B} {H}x{W}: HIP vs f32 restatement {err:.2e}; HIP vs f64-accumulating {float(np.abs(y - ref64).max() / sc):.2e}; "
          f"f32 restatement vs f64 {float(np.abs(ref - ref64).max() / sc):.2e} (oracle {time.time()-t0:.1f} s)", flush=True)
