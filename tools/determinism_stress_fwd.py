"""Determinism stress of Unet.forward at batch 32 (round 5): the same forward again and again must give the same bits; at a mismatch
the intermediate activations (Unet.tap) say which stage differed first."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import cdc_compression_amd as cdc
from test_gpu_parity import load_case

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
kw, man, sd, _, _, _, _ = load_case("full_x")
un = cdc.Unet(**kw)
un.load_state_dict(sd)
B, S = 32, 256
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(77)
x = torch.randn((B, 3, S, S), generator=gen, device=dev) * 0.8
ctx = [torch.randn((B, c, S >> l, S >> l), generator=gen, device=dev) * 0.5 for l, c in enumerate([64, 64, 128, 192])]
t = torch.full((B,), 0.37, device=dev)
names = []
for i in range(6):
    names += ["downs.%d.%d" % (i, j) for j in range(4)]
names += ["mid_block1", "mid_attn", "mid_block2"]
for i in range(6):
    names += ["ups.%d.%d" % (i, j) for j in range(4)]
ref = un(x, t, ctx).clone()
ok_names, ref_taps = [], {}
for n in names:
    try:
        ref_taps[n] = un.tap(n); ok_names.append(n)
    except Exception:
        pass
print("taps:", ok_names, flush=True)
bad = 0
t0 = time.time()
for i in range(N):
    y = un(x, t, ctx)
    if not torch.equal(y, ref):
        bad += 1
        d = (y - ref).abs().amax(dim=(1, 2, 3))
        msg = "forward %d differs: images %s max %.3g;" % (i, torch.nonzero(d).flatten().tolist(), float(d.max()))
        for n in ok_names:
            a = un.tap(n)
            if not np.array_equal(a, ref_taps[n]):
                dd = np.abs(a - ref_taps[n])
                idx = np.argwhere(dd > 0)
                msg += " first differing tap %s %s: %d values, images %s, channels %s.., max %.3g" % (n, a.shape, len(idx), sorted(set(idx[:, 0].tolist())), sorted(set(idx[:, 1].tolist()))[:6], float(dd.max()))
                break
        print(msg, flush=True)
print("%d forwards, %d differ (%.0f s)" % (N, bad, time.time() - t0), flush=True)
