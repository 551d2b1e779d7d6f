"""Determinism stress of Unet.forward at batch 32, launches queued back to back (no host synchronisation inside a burst)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import cdc_compression_amd as cdc
from test_gpu_parity import load_case

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 20
BURST = 250
kw, man, sd, _, _, _, _ = load_case("full_x")
un = cdc.Unet(**kw)
un.load_state_dict(sd)
B, S = 32, 256
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(77)
x = torch.randn((B, 3, S, S), generator=gen, device=dev) * 0.8
ctx = [torch.randn((B, c, S >> l, S >> l), generator=gen, device=dev) * 0.5 for l, c in enumerate([64, 64, 128, 192])]
t = torch.full((B,), 0.37, device=dev)
ref = un(x, t, ctx).clone()
bad = 0
t0 = time.time()
for b in range(NB):
    ys = [un(x, t, ctx) for _ in range(BURST)]
    torch.cuda.synchronize()
    for i, y in enumerate(ys):
        if not torch.equal(y, ref):
            bad += 1
            d = (y - ref).abs().amax(dim=(1, 2, 3))
            print("burst %d forward %d differs: images %s max %.3g" % (b, i, torch.nonzero(d).flatten().tolist(), float(d.max())), flush=True)
    del ys
print("%d forwards in bursts of %d, %d differ (%.0f s)" % (NB * BURST, BURST, bad, time.time() - t0), flush=True)
