#!/usr/bin/env python3
"""Development aid: does running two half-batches on two streams (two handles, two host threads) beat one
full batch?  The few-pixel levels leave most CUs idle; another stream's high-resolution kernels can fill them."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cdc_compression_amd as cdc
from cdc_compression_amd import synth

KW = dict(dim=64, channels=3, context_channels=64, dim_mults=(1, 2, 3, 4, 5, 6), context_dim_mults=(1, 2, 3, 4))
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
NSPLIT = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda", 0)

def make(B, seed):
    un = cdc.Unet(**KW, device=0)
    un.load_state_dict(synth.unet_state_dict(un.manifest(), seed=0))
    diff = cdc.GaussianDiffusionX(un, None, None, num_timesteps=8193, pred_mode="x", var_schedule="cosine")
    g = torch.Generator(device=dev).manual_seed(seed)
    init = torch.randn((B, 3, 256, 256), generator=g, device=dev) * 0.8
    ctx = [torch.randn((B, c, 256 >> l, 256 >> l), generator=g, device=dev) * 0.5 for l, c in enumerate([64, 64, 128, 192])]
    return diff, ctx, init, (B, 3, 256, 256)

def run(m, stream, out, i):
    diff, ctx, init, shape = m
    with torch.cuda.stream(stream):
        out[i] = diff.decompress(ctx, shape, sample_steps=STEPS, init=init)

full = make(32, 1)
full[0].decompress(full[1], full[3], sample_steps=2, init=full[2]); torch.cuda.synchronize()
t0 = time.perf_counter(); full[0].decompress(full[1], full[3], sample_steps=STEPS, init=full[2]); torch.cuda.synchronize()
t_full = time.perf_counter() - t0
parts = [make(32 // NSPLIT, 10 + i) for i in range(NSPLIT)]
streams = [torch.cuda.Stream(device=dev) for _ in range(NSPLIT)]
outs = [None] * NSPLIT
for i, m in enumerate(parts):
    run(m, streams[i], outs, i)
torch.cuda.synchronize()
t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(parts[i], streams[i], outs, i)) for i in range(NSPLIT)]
for t in th: t.start()
for t in th: t.join()
torch.cuda.synchronize()
t_split = time.perf_counter() - t0
print(f"one batch of 32: {t_full*1e3/STEPS:.2f} ms/iter   {NSPLIT} x {32//NSPLIT} on {NSPLIT} streams: {t_split*1e3/STEPS:.2f} ms/iter  finite={all(bool(torch.isfinite(o).all()) for o in outs)}")
