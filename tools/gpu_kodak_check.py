#!/usr/bin/env python3
"""Development aid: BASELINE configs[0] on the GPU -- the three Kodak crops through the whole compress() (GPU
compressor + 500-step decode) against the reference's CPU run (tests/golden/kodak_x_500.npz)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cdc_compression_amd as cdc
from cdc_compression_amd import synth
G = os.path.join(ROOT, "tests", "golden")
g = np.load(os.path.join(G, "kodak_x_500.npz"))
kw = dict(dim=64, channels=3, context_channels=64, dim_mults=(1, 2, 3, 4, 5, 6), context_dim_mults=(1, 2, 3, 4))
un = cdc.Unet(**kw); un.load_state_dict(synth.unet_state_dict(un.manifest(), seed=0))
meta = json.load(open(os.path.join(G, "manifest_encoder_full_x.json")))
comp = cdc.ResnetCompressor(**meta["kwargs"])
comp.load_state_dict(synth.unet_state_dict([(k, tuple(v)) for k, v in meta["manifest"]], seed=15))
diff = cdc.GaussianDiffusionX(un, comp, None, num_timesteps=8193, pred_mode="x", var_schedule="cosine")
x = (g["crops"].astype(np.float32).transpose(0, 3, 1, 2) / 255.0 * 2.0 - 1.0).astype(np.float32)
init = synth.normal("init", x.shape, seed=1, std=0.8)
fo = comp(x)
ql = fo["q_latent"].reshape(-1)
print("q_latent: max |diff| on the digest", float(np.abs(ql[g["q_idx"]] - g["q_val"]).max()), "sum", float(ql.astype(np.float64).sum()), float(g["q_sum"]))
rec, bpp = diff.compress(x, sample_steps=int(g["steps"]), bpp_return_mean=False, init=init)
flat = rec.reshape(-1)
d = np.abs(flat[g["rec_idx"]] - g["rec_val"])
print("bpp", bpp, g["bpp"])
print("reconstruction digest: max |diff|", float(d.max()), "mean", float(d.mean()), "sum", float(flat.astype(np.float64).sum()), float(g["rec_sum"]))
psnr = [10 * np.log10(4.0 / np.mean((rec[i] - x[i]) ** 2)) for i in range(3)]
print("psnr vs input (synthetic weights, meaningless but comparable):", psnr, g["psnr"])

# decode path alone: the reference's own q_latent -> GPU context decoder -> 500-step decode
ctx = comp.decode(g["q_latent"])
rec2 = diff.decompress(ctx, x.shape, sample_steps=int(g["steps"]), init=init)
d2 = np.abs(rec2.reshape(-1)[g["rec_idx"]] - g["rec_val"])
print("decode from the reference's q_latent: max |diff|", float(d2.max()), "mean", float(d2.mean()),
      "sum", float(rec2.astype(np.float64).sum()), float(g["rec_sum"]), "flipped symbols:", int((np.abs(fo["q_latent"] - g["q_latent"]) > 0.5).sum()))
