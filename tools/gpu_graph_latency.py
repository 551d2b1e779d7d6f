#!/usr/bin/env python3
"""Development aid: per-iteration latency of small batches, eager launches vs hipGraph replay (CDC_GRAPH)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cdc_compression_amd as cdc
from cdc_compression_amd import synth
KW = dict(dim=64, channels=3, context_channels=64, dim_mults=(1, 2, 3, 4, 5, 6), context_dim_mults=(1, 2, 3, 4))
dev = torch.device("cuda", 0)
un = cdc.Unet(**KW); un.load_state_dict(synth.unet_state_dict(un.manifest(), seed=0))
diff = cdc.GaussianDiffusionX(un, None, None, num_timesteps=8193, pred_mode="x", var_schedule="cosine")
g = torch.Generator(device=dev).manual_seed(1)
for B in (1, 2, 4, 8, 16):
    init = torch.randn((B, 3, 256, 256), generator=g, device=dev) * 0.8
    ctx = [torch.randn((B, c, 256 >> l, 256 >> l), generator=g, device=dev) * 0.5 for l, c in enumerate([64, 64, 128, 192])]
    res = {}
    for mode in ("0", "1"):
        os.environ["CDC_GRAPH"] = mode
        diff.decompress(ctx, (B, 3, 256, 256), sample_steps=6, init=init); torch.cuda.synchronize()
        t = time.perf_counter()
        out = diff.decompress(ctx, (B, 3, 256, 256), sample_steps=100, init=init); torch.cuda.synchronize()
        res[mode] = ((time.perf_counter() - t) / 100 * 1e3, out)
    print(f"B={B:2d}: eager {res['0'][0]:.2f} ms/iter, graph {res['1'][0]:.2f} ms/iter, identical={bool(torch.equal(res['0'][1], res['1'][1]))}")
