#!/usr/bin/env python3
"""Development aid (round 4): ddim_rows4_kernel (sampler update + 7-row combine on 16-byte accesses) against ddim_kernel
(CDC_NO_DDIM_ROWS4=1): 30-step decodes (x-param clip all, eps-param clip half) must be bit-identical.
usage: gpu_ddim_rows4_check.py [child param out.pt]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def child(param, out):
    import torch
    import cdc_compression_amd as cdc
    from cdc_compression_amd import synth
    KW = dict(dim=64, channels=3, context_channels=64, dim_mults=(1, 2, 3, 4, 5, 6), context_dim_mults=(1, 2, 3, 4))
    un = cdc.Unet(**KW); un.load_state_dict(synth.unet_state_dict(un.manifest(), seed=0))
    if param == "x":
        diff = cdc.GaussianDiffusionX(un, None, None, num_timesteps=8193, pred_mode="x", var_schedule="cosine")
    else:
        diff = cdc.GaussianDiffusionEps(un, None, num_timesteps=20000, clip_noise="half", pred_mode="noise", var_schedule="linear")
    B = 4
    dev = torch.device("cuda", 0); g = torch.Generator(device=dev).manual_seed(3)
    init = torch.randn((B, 3, 256, 256), generator=g, device=dev) * 0.8
    ctx = [torch.randn((B, c, 256 >> l, 256 >> l), generator=g, device=dev) * 0.5 for l, c in enumerate([64, 64, 128, 192])]
    a = diff.decompress(ctx, (B, 3, 256, 256), sample_steps=30, init=init)
    torch.save(a.cpu(), out)

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(sys.argv[2], sys.argv[3]); sys.exit(0)
    import torch
    ok = True
    for param in ("x", "eps"):
        res = []
        for new in (1, 0):
            env = dict(os.environ, CDC_DEV="1")
            if not new: env["CDC_NO_DDIM_ROWS4"] = "1"
            out = f"/tmp/ddim_{param}_{new}.pt"
            r = subprocess.run([sys.executable, __file__, "child", param, out], env=env, capture_output=True, text=True)
            if r.returncode: print(r.stderr[-600:]); sys.exit(1)
            res.append(torch.load(out))
        same = bool(torch.equal(res[0], res[1]))
        print(f"{param}-param: bit-identical {same}, max abs diff {(res[0] - res[1]).abs().max().item():.3e}, finite {bool(torch.isfinite(res[0]).all())}", flush=True)
        ok = ok and same
    print("DDIM_ROWS4_CHECK", "OK" if ok else "DIFFERS")
