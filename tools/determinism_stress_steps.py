"""Determinism stress, one DDIM step at a time (round 5): every step of a batch-32 decode is repeated from the SAME (reference) input and must
give the same bits; at a mismatch the intermediate activations (Unet.tap) of the bad execution are compared with a repeat of that step."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import cdc_compression_amd as cdc
from cdc_compression_amd import _lib
from test_gpu_parity import load_case

R = int(sys.argv[1]) if len(sys.argv) > 1 else 20
kw, man, sd, _, _, _, _ = load_case("full_x")
un = cdc.Unet(**kw)
un.load_state_dict(sd)
diff = cdc.GaussianDiffusionX(un, None, None, num_timesteps=8193, pred_mode="x", var_schedule="cosine")
B, S, steps = 32, 256, 500
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(77)
init = torch.randn((B, 3, S, S), generator=gen, device=dev) * 0.8
ctx = [torch.randn((B, c, S >> l, S >> l), generator=gen, device=dev) * 0.5 for l, c in enumerate([64, 64, 128, 192])]
diff.set_sample_schedule(steps)
L, h = _lib.lib(), un._handle()
pred, clip = diff._pred_flag(), diff._clip_flag(True)
ptrs = (ctypes.c_void_p * len(ctx))(*[c.data_ptr() for c in ctx])
def step(x, i, out, with_ctx=False):
    _lib.check(h, L.cdc_ddim_step(h, x.data_ptr(), i, ptrs if with_ctx else None, len(ctx) if with_ctx else 0, None, 0.0, out.data_ptr(),
                                  B, S, S, pred, clip, _lib.CDC_MEM_DEVICE, None))
names = []
for i in range(6):
    names += ["downs.%d.0" % i, "downs.%d.1" % i, "downs.%d.1.stat_mean" % i, "downs.%d.1.stat_rstd" % i, "downs.%d.2.kv" % i, "downs.%d.2.kmax" % i,
              "downs.%d.2.S" % i, "downs.%d.2.Z" % i, "downs.%d.2.M" % i, "downs.%d.2" % i, "downs.%d.3" % i]
names += ["mid_block1", "mid_attn", "mid_block2"] + ["ups.%d" % i for i in range(6)]
# reference trajectory
xin = [None] * steps
ref = [None] * steps
x = init.clone()
first = True
for i in reversed(range(steps)):
    out = torch.empty_like(x)
    step(x, i, out, with_ctx=first)
    first = False
    xin[i], ref[i] = x, out
    x = out
torch.cuda.synchronize()
ok_names = []
for n in names:
    try:
        un.tap(n); ok_names.append(n)
    except Exception:
        pass
print("taps:", ok_names, flush=True)
bad = 0
t0 = time.time()
out = torch.empty_like(init)
for rep in range(R):
    for i in reversed(range(steps)):
        step(xin[i], i, out)
        if not torch.equal(out, ref[i]):
            bad += 1
            d = (out - ref[i]).abs().amax(dim=(1, 2, 3))
            msg = "rep %d step %d differs: images %s max %.3g;" % (rep, i, torch.nonzero(d).flatten().tolist(), float(d.max()))
            bad_taps = {n: un.tap(n) for n in ok_names}
            step(xin[i], i, out)
            again = torch.equal(out, ref[i])
            msg += " repeat %s;" % ("clean" if again else "ALSO differs")
            for n in ok_names:
                a = un.tap(n)
                if not np.array_equal(a, bad_taps[n]):
                    dd = np.abs(a - bad_taps[n])
                    idx = np.argwhere(dd > 0)
                    msg += " first differing tap %s %s: %d values, images %s, channels %s (%d distinct), rows %s, max %.3g" % (
                        n, a.shape, len(idx), sorted(set(idx[:, 0].tolist())), sorted(set(idx[:, 1].tolist()))[:8], len(set(idx[:, 1].tolist())),
                        sorted(set(idx[:, 2].tolist()))[:8], float(dd.max()))
                    break
            print(msg, flush=True)
print("%d steps, %d differ (%.0f s)" % (R * steps, bad, time.time() - t0), flush=True)
