"""Determinism stress (round 5): the same decode again and again must give the same bits (the model path has no atomics, every
summation order is fixed by the launch geometry).  python tools/determinism_stress.py [n32] [n1] -> one line per mismatch + a summary."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import cdc_compression_amd as cdc
from test_gpu_parity import load_case

n32 = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n1 = int(sys.argv[2]) if len(sys.argv) > 2 else 50
kw, man, sd, _, _, _, _ = load_case("full_x")
un = cdc.Unet(**kw)
un.load_state_dict(sd)
diff = cdc.GaussianDiffusionX(un, None, None, num_timesteps=8193, pred_mode="x", var_schedule="cosine")
B, S, steps = 32, 256, 500
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(77)
init = torch.randn((B, 3, S, S), generator=gen, device=dev) * 0.8
ctx = [torch.randn((B, c, S >> l, S >> l), generator=gen, device=dev) * 0.5 for l, c in enumerate([64, 64, 128, 192])]
bad = 0
ref = None
t0 = time.time()
for i in range(n32):
    rec = diff.decompress(ctx, (B, 3, S, S), sample_steps=steps, init=init)
    if ref is None:
        ref = rec.clone()
    elif not torch.equal(rec, ref):
        d = (rec - ref).abs().amax(dim=(1, 2, 3))
        bad += 1
        print("batch-32 decode %d differs: rows %s max %s" % (i, torch.nonzero(d).flatten().tolist(), float(d.max())), flush=True)
print("batch 32: %d decodes, %d differ (%.0f s)" % (n32, bad, time.time() - t0), flush=True)
k = 17
ref1 = None
bad1 = 0
t0 = time.time()
for i in range(n1):
    r1 = diff.decompress([c[k:k + 1] for c in ctx], (1, 3, S, S), sample_steps=steps, init=init[k:k + 1])
    if ref1 is None:
        ref1 = r1.clone()
    elif not torch.equal(r1, ref1):
        bad1 += 1
        print("batch-1 decode %d differs: max %g" % (i, float((r1 - ref1).abs().max())), flush=True)
print("batch 1: %d decodes, %d differ (%.0f s)" % (n1, bad1, time.time() - t0), flush=True)
if ref is not None and ref1 is not None:
    print("row 17, batch 32 against batch 1: %.3e" % float((ref[k] - ref1[0]).abs().max()))
sys.exit(1 if (bad or bad1) else 0)
