#!/usr/bin/env python3
"""Generate golden vectors from the REAL reference (buggyyang/CDC_compression at /root/reference).

Runs only in the build container (the reference cannot travel to the GPU box).  It imports the
reference's own modules (with a stub `lpips`, which the decode path never calls), injects the
deterministic synthetic parameters of cdc_compression_amd.synth, runs the reference's PyTorch CPU
path, and stores inputs + expected outputs as small .npz / .json fixtures in this directory.

    python tests/golden/make_golden.py            # regenerate everything (~2-3 min, 8 cores)

Fixtures are data only: no reference source text is stored.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from cdc_compression_amd import synth  # noqa: E402

REF = "/root/reference"


def import_reference(tree):
    """Import `modules.*` of one reference tree (xparam | epsilonparam) under a private name."""
    sys.modules["lpips"] = types.SimpleNamespace(LPIPS=lambda **k: None)
    for k in [k for k in sys.modules if k == "modules" or k.startswith("modules.")]:
        del sys.modules[k]
    sys.path.insert(0, os.path.join(REF, tree))
    try:
        import modules.unet as unet
        import modules.denoising_diffusion as dd
        import modules.compress_modules as cm
        import modules.network_components as nc
    finally:
        sys.path.pop(0)
    return types.SimpleNamespace(unet=unet, dd=dd, cm=cm, nc=nc)


def manifest_of(module):
    return [(k, list(v.shape)) for k, v in module.state_dict().items()]


def load_synth(module, seed, final_gain=1.0):
    man = manifest_of(module)
    sd = synth.unet_state_dict(man, seed=seed, final_gain=final_gain)
    module.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    module.eval()
    return man, sd


def digest(a, nsample=64, seed=11):
    a = np.asarray(a, np.float32)
    flat = a.reshape(-1)
    idx = (synth._splitmix64(np.arange(nsample, dtype=np.uint64) + np.uint64(seed * 1000))
           % np.uint64(flat.size)).astype(np.int64)
    return {"shape": list(a.shape), "sum": float(flat.astype(np.float64).sum()),
            "sumsq": float((flat.astype(np.float64) ** 2).sum()),
            "idx": idx, "val": flat[idx].copy()}


CONFIGS = {
    # name: (tree, unet kwargs, ctx channel list, H, W, B)
    "small_x": ("xparam", dict(dim=16, channels=3, context_channels=8, dim_mults=(1, 2, 3),
                               context_dim_mults=(1, 2), embd_type="01"), [8, 16], 32, 32, 2),
    "small_eps": ("epsilonparam", dict(dim=16, channels=3, context_channels=3,
                                       dim_mults=(1, 2, 3), context_dim_mults=(1, 2)),
                  [3, 16], 32, 32, 2),
    "odd_x": ("xparam", dict(dim=24, channels=3, context_channels=5, dim_mults=(1, 3),
                             context_dim_mults=(1,), embd_type="01"), [5], 24, 40, 1),
    "full_x": ("xparam", dict(dim=64, channels=3, context_channels=64,
                              dim_mults=(1, 2, 3, 4, 5, 6), context_dim_mults=(1, 2, 3, 4),
                              embd_type="01"), [64, 64, 128, 192], 64, 64, 1),
    "full_eps": ("epsilonparam", dict(dim=64, channels=3, context_channels=3,
                                      dim_mults=(1, 2, 3, 4, 5, 6),
                                      context_dim_mults=(1, 2, 3, 4)), [3, 64, 128, 192], 64, 64, 1),
}

DIFF = {
    "xparam": dict(num_timesteps=8193, loss_type="l2", lagrangian=0.0032, pred_mode="x",
                   aux_loss_weight=0, aux_loss_type="lpips", var_schedule="cosine",
                   use_loss_weight=True, loss_weight_min=5),
    "epsilonparam": dict(num_timesteps=20000, loss_type="l1", clip_noise="none", vbr=False,
                         lagrangian=0.9, pred_mode="noise", var_schedule="linear",
                         aux_loss_weight=0, aux_loss_type="lpips"),
}


class FixedContext(torch.nn.Module):
    """Stand-in context_fn returning a fixed pyramid (the decode path only reads ["output"])."""

    def __init__(self, ctx):
        super().__init__()
        self.ctx = ctx

    def forward(self, images, *a):
        return {"output": self.ctx, "bpp": torch.zeros(images.shape[0])}


def gen_unet(name, taps=True):
    tree, kw, ctxc, H, W, B = CONFIGS[name]
    ref = import_reference(tree)
    torch.manual_seed(0)
    net = ref.unet.Unet(**kw)
    man, sd = load_synth(net, seed=0, final_gain=0.2 if tree == "epsilonparam" else 1.0)
    x = synth.normal("x", (B, 3, H, W), seed=1, std=0.8)
    ctx = synth.context_pyramid(ctxc, B, H, W, seed=3)
    time = np.linspace(0.1, 0.7, B, dtype=np.float32).reshape(B, 1)
    acts = {}
    hooks = []
    if taps:
        def mk(key):
            def fn(m, i, o):
                acts[key] = o.detach().numpy().copy()
            return fn
        for key, mod in [("downs.0.0", net.downs[0][0]), ("downs.0.2", net.downs[0][2]),
                         ("downs.1.3", net.downs[1][3] if len(net.downs) > 2 else net.downs[0][3]),
                         ("mid_block1", net.mid_block1), ("ups.0", net.ups[0][3])]:
            hooks.append(mod.register_forward_hook(mk(key)))
    with torch.no_grad():
        y = net(torch.from_numpy(x), torch.from_numpy(time),
                [torch.from_numpy(c) for c in ctx]).numpy()
    for h in hooks:
        h.remove()
    out = {"x": x, "time": time, "y": y, "B": B, "H": H, "W": W}
    for i, c in enumerate(ctx):
        out[f"ctx{i}"] = c
    if name.startswith("full"):
        # full-width model: keep the file small (inputs are regenerated from synth; store digests)
        out = {"time": time, "y": y, "B": B, "H": H, "W": W}
        for k, v in acts.items():
            d = digest(v)
            out[f"tap_{k}_idx"], out[f"tap_{k}_val"] = d["idx"], d["val"]
            out[f"tap_{k}_sum"] = d["sum"]
    else:
        for k, v in acts.items():
            out[f"tap_{k}"] = v
    if taps:          # (taps=False: a caller that only wants the loaded reference model must not overwrite the fixture)
        np.savez_compressed(os.path.join(HERE, f"unet_{name}.npz"), **out)
        with open(os.path.join(HERE, f"manifest_{name}.json"), "w") as f:
            json.dump({"unet_kwargs": {k: (list(v) if isinstance(v, tuple) else v)
                                       for k, v in kw.items()},
                       "context_channels_per_level": ctxc, "manifest": man}, f)
    print(name, "unet ok", y.shape, float(np.abs(y).max()))
    return ref, net, kw, ctxc


def gen_decode(name, steps_list, eta_case=False):
    tree, kw, ctxc, H, W, B = CONFIGS[name]
    ref, net, _, _ = gen_unet(name, taps=not name.startswith("full") or True)
    ctx = synth.context_pyramid(ctxc, B, H, W, seed=3)
    tctx = [torch.from_numpy(c) for c in ctx]
    diff = ref.dd.GaussianDiffusion(denoise_fn=net, context_fn=FixedContext(tctx),
                                    **({"ae_fn": None} if tree == "xparam" else {}), **DIFF[tree])
    diff.eval()
    init = synth.normal("init", (B, 3, H, W), seed=1, std=0.8)
    images = np.zeros((B, 3, H, W), np.float32)
    out = {}
    for steps in steps_list:
        with torch.no_grad():
            if tree == "xparam":
                rec, _ = diff.compress(torch.from_numpy(images), sample_steps=steps,
                                       bpp_return_mean=True, init=torch.from_numpy(init.copy()))
            else:
                rec, _ = diff.compress(torch.from_numpy(images), sample_steps=steps,
                                       sample_mode="ddim", bpp_return_mean=False,
                                       init=torch.from_numpy(init.copy()))
        out[f"decode_{steps}"] = rec.numpy()
        print(name, "decode", steps, float(rec.abs().max()))
    if eta_case:
        # eta != 0: the reference draws torch.randn_like every step from the global CPU RNG;
        # record the draws so the same noise can be fed to the build (xparam :172 / eps :150).
        steps = steps_list[0]
        torch.manual_seed(1234)
        noises = np.stack([torch.randn((B, 3, H, W)).numpy() for _ in range(steps)])
        torch.manual_seed(1234)
        with torch.no_grad():
            if tree == "xparam":
                rec, _ = diff.compress(torch.from_numpy(images), sample_steps=steps,
                                       init=torch.from_numpy(init.copy()), eta=0.5)
            else:
                rec, _ = diff.compress(torch.from_numpy(images), sample_steps=steps,
                                       sample_mode="ddim", init=torch.from_numpy(init.copy()),
                                       eta=0.5)
        out["eta_steps"] = steps
        out["eta_noises"] = noises
        out["eta_decode"] = rec.numpy()
    np.savez_compressed(os.path.join(HERE, f"decode_{name}.npz"), **out)


def gen_decode_variants():
    """The sampler branches the test scripts do not take: x-tree pred_mode="noise" (xparam :155-156,165; it is the
    reference constructor's default) and eps-tree clip_noise="half" (epsilonparam :142-143, the constructor's default)."""
    out = {}
    for key, name, over in (("small_x", "small_x", {"pred_mode": "noise"}), ("small_eps", "small_eps", {"clip_noise": "half"}),
                            ("small_x_v", "small_x", {"pred_mode": "v"})):       # xparam :128-139,161-162
        tree, kw, ctxc, H, W, B = CONFIGS[name]
        ref, net, _, _ = gen_unet(name, taps=False)
        ctx = synth.context_pyramid(ctxc, B, H, W, seed=3)
        tctx = [torch.from_numpy(c) for c in ctx]
        args = dict(DIFF[tree]); args.update(over)
        diff = ref.dd.GaussianDiffusion(denoise_fn=net, context_fn=FixedContext(tctx),
                                        **({"ae_fn": None} if tree == "xparam" else {}), **args)
        diff.eval()
        init = synth.normal("init", (B, 3, H, W), seed=1, std=0.8)
        images = np.zeros((B, 3, H, W), np.float32)
        with torch.no_grad():
            if tree == "xparam":
                rec, _ = diff.compress(torch.from_numpy(images), sample_steps=3, init=torch.from_numpy(init.copy()))
            else:
                rec, _ = diff.compress(torch.from_numpy(images), sample_steps=3, sample_mode="ddim",
                                       bpp_return_mean=False, init=torch.from_numpy(init.copy()))
        out[key] = rec.numpy()
        print("variant", key, over, float(rec.abs().max()))
    np.savez_compressed(os.path.join(HERE, "decode_variants.npz"), **out)


def heavy_tail_state(man, sd, gain_max, g_max, seed=31, only_normalised=False):
    """Heavy-tailed parameters on top of the synthetic ones: the output channels of the convolutions get log-uniform gains
    in [1 / gain_max, gain_max] (only_normalised: just those whose output a LayerNorm rescales, i.e. Block convolutions),
    every LayerNorm gain is log-uniform in [1 / g_max, g_max] with a random sign."""
    rng = np.random.default_rng(seed)
    out = {}
    for k, shape in man:
        v = np.array(sd[k], np.float32, copy=True)
        if k.endswith(".weight") and v.ndim == 4 and "final_conv" not in k and (not only_normalised or ".block." in k):
            co = v.shape[1] if ".up." in k or "upsample" in k.lower() else v.shape[0]
            gains = np.exp(rng.uniform(-np.log(gain_max), np.log(gain_max), co)).astype(np.float32)
            if v.shape[0] == co:
                v *= gains[:, None, None, None]
            else:
                v *= gains[None, :, None, None]
        elif k.endswith(".g"):
            mag = np.exp(rng.uniform(-np.log(g_max), np.log(g_max), v.shape)).astype(np.float32)
            v = mag * np.where(rng.random(v.shape) < 0.5, -1.0, 1.0).astype(np.float32)
        out[k] = v
    return out


def heavy_tail_context(ctxc, B, H, W, lo, hi, seed=33):
    """Context pyramid with log-uniform magnitudes in [lo, hi] and random signs."""
    rng = np.random.default_rng(seed)
    out = []
    for i, c in enumerate(ctxc):
        shape = (B, c, H >> i, W >> i)
        mag = np.exp(rng.uniform(np.log(lo), np.log(hi), shape))
        out.append((mag * np.where(rng.random(shape) < 0.5, -1.0, 1.0)).astype(np.float32))
    return out


def gen_heavy_tail():
    """evidence at a trained-weight-like dynamic range.  Two cases of the small x-param model from the
    REAL reference: "in_range" (gains x50 on every LayerNorm-ed convolution, LN gains up to 10, context 1e-6 .. 300: a
    large dynamic range with every convolution INPUT inside the fp16 range of CDC_ARITH_F16X2) and "overflow" (gains x50
    on every convolution, context up to 3e4: convolution inputs of 1e11, the range guard has to act).  One U-Net forward +
    a 3-step decode each."""
    out = {}
    for case, gain_max, g_max, lo, hi in (("in_range", 50.0, 10.0, 1e-6, 300.0), ("overflow", 50.0, 10.0, 1e-6, 3.0e4)):
        tree, kw, ctxc, H, W, B = CONFIGS["small_x"]
        ref = import_reference(tree)
        net = ref.unet.Unet(**kw)
        man, sd = load_synth(net, seed=0)
        sd2 = heavy_tail_state(man, sd, gain_max, g_max, only_normalised=case == "in_range")
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd2.items()})
        net.eval()
        ctx = heavy_tail_context(ctxc, B, H, W, lo, hi)
        tctx = [torch.from_numpy(c) for c in ctx]
        x = synth.normal("x", (B, 3, H, W), seed=1, std=0.8)
        time = np.linspace(0.1, 0.7, B, dtype=np.float32).reshape(B, 1)
        biggest = [0.0]
        # what the fp16 planes of CDC_ARITH_F16X2 have to hold: the INPUT of every convolution
        hooks = [m.register_forward_hook(lambda m_, i_, o_: biggest.__setitem__(0, max(biggest[0], float(i_[0].detach().abs().max()))))
                 for m in net.modules() if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d))]
        with torch.no_grad():
            y = net(torch.from_numpy(x), torch.from_numpy(time), tctx).numpy()
        for hk in hooks:
            hk.remove()
        diff = ref.dd.GaussianDiffusion(denoise_fn=net, context_fn=FixedContext(tctx), ae_fn=None, **DIFF[tree])
        diff.eval()
        init = synth.normal("init", (B, 3, H, W), seed=1, std=0.8)
        with torch.no_grad():
            rec, _ = diff.compress(torch.zeros(B, 3, H, W), sample_steps=3, init=torch.from_numpy(init.copy()))
        out[f"{case}_y"] = y
        out[f"{case}_rec"] = rec.numpy()
        out[f"{case}_max_conv_input"] = np.array(biggest[0], np.float64)
        for i, c in enumerate(ctx):
            out[f"{case}_ctx{i}"] = c
        for k, v in sd2.items():
            if k.endswith(".g") or (k.endswith(".weight") and v.ndim == 4 and "final_conv" not in k):
                out[f"{case}_sd_{k}"] = v
        print("heavy tail", case, "largest convolution input", biggest[0], "| y max", float(np.abs(y).max()), "| rec max", float(np.abs(rec.numpy()).max()))
    np.savez_compressed(os.path.join(HERE, "heavy_tail_small_x.npz"), **out)


def gen_schedules():
    out = {}
    for tree in ("xparam", "epsilonparam"):
        ref = import_reference(tree)
        net = ref.unet.Unet(dim=8, dim_mults=(1,), context_dim_mults=(1,))
        diff = ref.dd.GaussianDiffusion(denoise_fn=net, context_fn=None,
                                        **({"ae_fn": None} if tree == "xparam" else {}),
                                        **DIFF[tree])
        tag = "x" if tree == "xparam" else "eps"
        out[f"{tag}_train_alphas_cumprod_digest"] = np.array(
            [float(diff.train_alphas_cumprod.double().sum()),
             float(diff.train_alphas_cumprod[-1]), float(diff.train_alphas_cumprod[0])])
        for steps in (1, 2, 4, 7, 65, 200, 500, 1000):
            diff.set_sample_schedule(steps, torch.device("cpu"))
            for nm in ("alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod_prev",
                       "one_minus_alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
                       "sqrt_recipm1_alphas_cumprod", "sigma", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod"):
                out[f"{tag}_{steps}_{nm}"] = getattr(diff, nm).numpy().copy()
            if tree == "xparam":
                out[f"{tag}_{steps}_index"] = diff.index.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "schedules.npz"), **out)
    print("schedules ok")


def gen_full_res():
    """BASELINE-size (256x256) x-param U-Net forward + 4-step decode of the real reference, B=1,
    stored as digests (sums + 64 sampled pixels)."""
    tree, kw, ctxc, _, _, _ = CONFIGS["full_x"]
    ref = import_reference(tree)
    net = ref.unet.Unet(**kw)
    load_synth(net, seed=0)
    B, H, W = 1, 256, 256
    x = synth.normal("x", (B, 3, H, W), seed=1, std=0.8)
    ctx = synth.context_pyramid(ctxc, B, H, W, seed=3)
    tctx = [torch.from_numpy(c) for c in ctx]
    time = np.full((B, 1), 0.37, np.float32)
    with torch.no_grad():
        y = net(torch.from_numpy(x), torch.from_numpy(time), tctx).numpy()
    d = digest(y)
    out = {"time": time, "y_idx": d["idx"], "y_val": d["val"], "y_sum": d["sum"],
           "y_sumsq": d["sumsq"]}
    diff = ref.dd.GaussianDiffusion(denoise_fn=net, context_fn=FixedContext(tctx), ae_fn=None,
                                    **DIFF[tree])
    diff.eval()
    init = synth.normal("init", (B, 3, H, W), seed=1, std=0.8)
    with torch.no_grad():
        rec, _ = diff.compress(torch.zeros(B, 3, H, W), sample_steps=4,
                               init=torch.from_numpy(init.copy()))
    d = digest(rec.numpy())
    out.update({"dec4_idx": d["idx"], "dec4_val": d["val"], "dec4_sum": d["sum"],
                "dec4_sumsq": d["sumsq"]})
    np.savez_compressed(os.path.join(HERE, "full_res_x_256.npz"), **out)
    print("full res ok", d["sum"])


def gen_full_res_other(name, cfg, H, W, steps=4):
    """Digests of the real reference at the other BASELINE shapes: x-param 512x512 (configs[4]) and eps-param
    256x256 (configs[2]), B=1: one U-Net forward + a `steps`-step decode."""
    tree, kw, ctxc, _, _, _ = CONFIGS[cfg]
    ref = import_reference(tree)
    net = ref.unet.Unet(**kw)
    load_synth(net, seed=0, final_gain=0.2 if tree == "epsilonparam" else 1.0)
    B = 1
    x = synth.normal("x", (B, 3, H, W), seed=1, std=0.8)
    ctx = synth.context_pyramid(ctxc, B, H, W, seed=3)
    tctx = [torch.from_numpy(c) for c in ctx]
    time = np.full((B, 1), 0.37, np.float32)
    with torch.no_grad():
        y = net(torch.from_numpy(x), torch.from_numpy(time), tctx).numpy()
    d = digest(y)
    out = {"time": time, "y_idx": d["idx"], "y_val": d["val"], "y_sum": d["sum"], "y_sumsq": d["sumsq"]}
    extra = {"ae_fn": None} if tree == "xparam" else {}
    diff = ref.dd.GaussianDiffusion(denoise_fn=net, context_fn=FixedContext(tctx), **extra, **DIFF[tree])
    diff.eval()
    init = synth.normal("init", (B, 3, H, W), seed=1, std=0.8)
    with torch.no_grad():
        if tree == "xparam":
            rec, _ = diff.compress(torch.zeros(B, 3, H, W), sample_steps=steps, init=torch.from_numpy(init.copy()))
        else:
            rec, _ = diff.compress(torch.zeros(B, 3, H, W), sample_steps=steps, sample_mode="ddim",
                                   init=torch.from_numpy(init.copy()))
    d = digest(rec.numpy())
    out.update({"dec_idx": d["idx"], "dec_val": d["val"], "dec_sum": d["sum"], "dec_sumsq": d["sumsq"], "steps": steps})
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "ok", d["sum"])


CTXDEC = {
    # name: (tree, class name, ctor kwargs, up_index, latent h, w, B)
    "ctxdec_small_x": ("xparam", "ResnetCompressor",
                       dict(dim=8, dim_mults=[1, 2, 3, 4], reverse_dim_mults=[4, 3, 2, 1],
                            hyper_dims_mults=[4, 4, 4], channels=3, out_channels=8), 1, 3, 5, 2),
    "ctxdec_small_eps": ("epsilonparam", "BigCompressor",
                         dict(dim=8, dim_mults=(1, 2, 3, 4), hyper_dims_mults=(4, 4, 4), channels=3,
                              out_channels=3, vbr=False), 2, 4, 4, 2),
    "ctxdec_full_x": ("xparam", "ResnetCompressor",
                      dict(dim=64, dim_mults=[1, 2, 3, 4], reverse_dim_mults=[4, 3, 2, 1],
                           hyper_dims_mults=[4, 4, 4], channels=3, out_channels=64), 1, 4, 4, 1),
    "ctxdec_full_eps": ("epsilonparam", "BigCompressor",
                        dict(dim=64, dim_mults=(1, 2, 3, 4), hyper_dims_mults=(4, 4, 4), channels=3,
                             out_channels=3, vbr=False), 2, 4, 4, 1),
}


def gen_ctxdec(name):
    """Compressor.decode of the real reference on a synthetic q_latent: the `dec.*` parameters come
    from synth (the rest of the module is untouched and unused), outputs stored in full for the small
    configurations and as digests for the full-width ones."""
    tree, cls, kw, up_index, hl, wl, B = CTXDEC[name]
    ref = import_reference(tree)
    net = getattr(ref.cm, cls)(**kw)
    man = [(k, list(v.shape)) for k, v in net.state_dict().items() if k.startswith("dec.")]
    sd = synth.unet_state_dict(man, seed=5)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    net.eval()
    c0 = man[0][1][1]                                   # dec.0.0.block1.block.0.weight: [mid][Cin][3][3]
    q = np.round(synth.normal("q_latent", (B, c0, hl, wl), seed=6, std=2.0)).astype(np.float32)
    with torch.no_grad():
        outs = [o.numpy() for o in net.decode(torch.from_numpy(q))]
    json.dump({"kwargs": {k: (list(v) if isinstance(v, (list, tuple)) else v) for k, v in kw.items()},
               "class": cls, "tree": tree, "up_index": up_index, "manifest": man},
              open(os.path.join(HERE, f"manifest_{name}.json"), "w"))
    rec = {"q_latent": q}
    for i, o in enumerate(outs):
        if o.size <= 70000:
            rec[f"out{i}"] = o
        d = digest(o)
        rec.update({f"out{i}_shape": np.array(o.shape), f"out{i}_idx": d["idx"], f"out{i}_val": d["val"],
                    f"out{i}_sum": d["sum"], f"out{i}_sumsq": d["sumsq"]})
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **rec)
    print(name, "ok", [o.shape for o in outs])


HYPERDEC = {
    # name: (tree, class, ctor kwargs, hyper-latent h, w, B)
    "hyperdec_small_x": ("xparam", "ResnetCompressor",
                         dict(dim=8, dim_mults=[1, 2, 3, 4], reverse_dim_mults=[4, 3, 2, 1],
                              hyper_dims_mults=[4, 4, 4], channels=3, out_channels=8), 2, 3, 2),
    "hyperdec_full_x": ("xparam", "ResnetCompressor",
                        dict(dim=64, dim_mults=[1, 2, 3, 4], reverse_dim_mults=[4, 3, 2, 1],
                             hyper_dims_mults=[4, 4, 4], channels=3, out_channels=64), 4, 4, 1),
    "hyperdec_full_eps": ("epsilonparam", "BigCompressor",
                          dict(dim=64, dim_mults=(1, 2, 3, 4), hyper_dims_mults=(4, 4, 4), channels=3,
                               out_channels=3, vbr=False), 3, 5, 1),
}


def gen_hyperdec(name):
    """hyper_dec of the real reference (compress_modules.py:54-59) on a synthetic q_hyper_latent, plus the
    reference's own dequantize (utils.py quantize(..., "dequantize", mean)) of a synthetic latent."""
    tree, cls, kw, hh, wh, B = HYPERDEC[name]
    ref = import_reference(tree)
    net = getattr(ref.cm, cls)(**kw)
    man = [(k, list(v.shape)) for k, v in net.state_dict().items() if k.startswith("hyper_dec.")]
    sd = synth.unet_state_dict(man, seed=7)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    net.eval()
    c0 = man[0][1][0]                                   # hyper_dec.0.0.weight: ConvTranspose2d [Cin][Cout][5][5]
    q = np.round(synth.normal("q_hyper", (B, c0, hh, wh), seed=8, std=2.0)).astype(np.float32) + np.float32(0.25)
    import modules.utils as ut
    with torch.no_grad():
        x = torch.from_numpy(q)
        for layer in net.hyper_dec:
            for m in layer:
                x = m(x)
        mean, scale = x.chunk(2, 1)
        scale = scale.clamp(min=0.1)
        latent = torch.from_numpy(synth.normal("latent", tuple(mean.shape), seed=9, std=3.0))
        ql = ut.quantize(latent, "dequantize", mean)
    # rate estimate: the reference's own bpp() (eval mode) on synthetic latents, prior parameters from synth
    pman = [(k, list(v.shape)) for k, v in net.state_dict().items()
            if k.startswith("prior.affine") or k.startswith("prior.a.")]
    psd = synth.unet_state_dict(pman, seed=11)
    for k in psd:
        if ".weight" in k:
            psd[k] = (psd[k] * 2.0).astype(np.float32)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in psd.items()}, strict=False)
    with torch.no_grad():
        hyper_latent = torch.from_numpy(synth.normal("hyper_latent", (B, c0, hh, wh), seed=10, std=2.0))
        latent_b = mean + torch.from_numpy(synth.normal("latent_b", tuple(mean.shape), seed=14, std=1.0)) * scale
        state = {"latent": latent_b, "hyper_latent": hyper_latent,
                 "latent_distribution": ut.NormalDistribution(mean, scale)}
        img_hw = (hh * 64, wh * 64)
        bpp = net.bpp((B, 3) + img_hw, state)
        q_hyper_for_bpp = ut.quantize(hyper_latent, "dequantize", net.prior.medians)
        q_latent_for_bpp = ut.quantize(latent_b, "dequantize", mean)
    json.dump({"kwargs": {k: (list(v) if isinstance(v, (list, tuple)) else v) for k, v in kw.items()},
               "class": cls, "tree": tree, "dims": list(net.reversed_hyper_dims), "manifest": man,
               "prior_manifest": pman},
              open(os.path.join(HERE, f"manifest_{name}.json"), "w"))
    rec = {"q_hyper_latent": q, "bpp": bpp.numpy(), "q_hyper_for_bpp": q_hyper_for_bpp.numpy(), "q_latent_for_bpp": q_latent_for_bpp.numpy(),
           "img_hw": np.array(img_hw)}
    for key, t in (("mean", mean), ("scale", scale), ("q_latent", ql)):
        a = t.numpy()
        if a.size <= 70000:
            rec[key] = a
        d = digest(a)
        rec.update({f"{key}_shape": np.array(a.shape), f"{key}_idx": d["idx"], f"{key}_val": d["val"],
                    f"{key}_sum": d["sum"]})
    if "q_latent" not in rec:
        rec["latent_seed"] = np.array(9)
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **rec)
    print(name, "ok", tuple(mean.shape), float(scale.min()))


ENCODER = {
    # name: (tree, class, ctor kwargs, down_index, H, W, B)
    "encoder_small_x": ("xparam", "ResnetCompressor",
                        dict(dim=8, dim_mults=[1, 2, 3, 4], reverse_dim_mults=[4, 3, 2, 1],
                             hyper_dims_mults=[4, 4, 4], channels=3, out_channels=8), 1, 64, 128, 2),
    "encoder_full_x": ("xparam", "ResnetCompressor",
                       dict(dim=64, dim_mults=[1, 2, 3, 4], reverse_dim_mults=[4, 3, 2, 1],
                            hyper_dims_mults=[4, 4, 4], channels=3, out_channels=64), 1, 128, 128, 1),
    "encoder_full_eps": ("epsilonparam", "BigCompressor",
                         dict(dim=64, dim_mults=(1, 2, 3, 4), hyper_dims_mults=(4, 4, 4), channels=3,
                              out_channels=3, vbr=False), 2, 64, 128, 1),
}


def gen_encoder(name):
    """encode() of the real reference on a synthetic image: the unquantised latent / hyper_latent (the quantisers
    are covered by the hyperdec fixtures), plus the whole forward() (bpp and the context pyramid) for the
    end-to-end chain."""
    tree, cls, kw, down_index, H, W, B = ENCODER[name]
    ref = import_reference(tree)
    net = getattr(ref.cm, cls)(**kw)
    keep = ("enc.", "hyper_enc.", "hyper_dec.", "dec.", "prior.affine", "prior.a.")
    man = [(k, list(v.shape)) for k, v in net.state_dict().items() if k.startswith(keep)]
    sd = synth.unet_state_dict(man, seed=15)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    net.eval()
    x = (synth.normal("image", (B, 3, H, W), seed=16, std=0.5)).clip(-1, 1).astype(np.float32)
    with torch.no_grad():
        q_latent, q_hyper, st = net.encode(torch.from_numpy(x))
        out = net(torch.from_numpy(x))
    json.dump({"kwargs": {k: (list(v) if isinstance(v, (list, tuple)) else v) for k, v in kw.items()},
               "class": cls, "tree": tree, "down_index": down_index, "manifest": man,
               "medians_shape": list(net.prior.medians.shape)},
              open(os.path.join(HERE, f"manifest_{name}.json"), "w"))
    rec = {"image_shape": np.array(x.shape), "bpp": out["bpp"].numpy()}
    for key, t in (("latent", st["latent"]), ("hyper_latent", st["hyper_latent"]), ("q_latent", out["q_latent"]),
                   ("q_hyper_latent", out["q_hyper_latent"]), ("ctx0", out["output"][0]), ("ctx3", out["output"][3])):
        a = t.numpy()
        if a.size <= 40000:
            rec[key] = a
        d = digest(a)
        rec.update({f"{key}_shape": np.array(a.shape), f"{key}_idx": d["idx"], f"{key}_val": d["val"],
                    f"{key}_sum": d["sum"]})
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **rec)
    print(name, "ok", tuple(st["latent"].shape), tuple(st["hyper_latent"].shape), out["bpp"].numpy())


def gen_kodak(steps=500):
    """BASELINE configs[0]: the reference's full x-param compress() on the three Kodak images it ships
    (imgs/1.png..3.png, centre crop rows 128:384 / cols 256:512 as in SURVEY 8(d)), synthetic parameters, 500 DDIM
    steps on the CPU.  Stored: the crops (data), the reference's bpp, q_latent / reconstruction digests."""
    from PIL import Image
    tree = "xparam"
    ref = import_reference(tree)
    _, kw, _, _, _, _ = CONFIGS["full_x"]
    net = ref.unet.Unet(**kw)
    uman, usd = load_synth(net, seed=0)
    ckw = ENCODER["encoder_full_x"][2]
    comp = ref.cm.ResnetCompressor(**ckw)
    keep = ("enc.", "hyper_enc.", "hyper_dec.", "dec.", "prior.affine", "prior.a.")
    cman = [(k, list(v.shape)) for k, v in comp.state_dict().items() if k.startswith(keep)]
    csd = synth.unet_state_dict(cman, seed=15)
    comp.load_state_dict({k: torch.from_numpy(v) for k, v in csd.items()}, strict=False)
    comp.eval()
    crops = []
    for i in (1, 2, 3):
        im = np.asarray(Image.open(os.path.join(REF, "imgs", f"{i}.png")).convert("RGB"))
        if im.shape[0] > im.shape[1]:
            im = np.transpose(im, (1, 0, 2))
        crops.append(im[128:384, 256:512].copy())
    crops = np.stack(crops)                                         # [3, 256, 256, 3] uint8
    x = torch.from_numpy(crops).permute(0, 3, 1, 2).float() / 255.0 * 2.0 - 1.0
    diff = ref.dd.GaussianDiffusion(denoise_fn=net, context_fn=comp, ae_fn=None, **DIFF[tree])
    diff.eval()
    init = synth.normal("init", tuple(x.shape), seed=1, std=0.8)
    with torch.no_grad():
        cd = comp(x)
        rec, bpp = diff.compress(x, sample_steps=steps, bpp_return_mean=False, init=torch.from_numpy(init.copy()))
    rec = rec.numpy()
    d = digest(rec, nsample=256)
    dq = digest(cd["q_latent"].numpy(), nsample=256)
    np.savez_compressed(os.path.join(HERE, f"kodak_x_{steps}.npz"), crops=crops, steps=np.array(steps),
                        bpp=bpp.numpy(), rec_idx=d["idx"], rec_val=d["val"], rec_sum=d["sum"], rec_sumsq=d["sumsq"],
                        q_idx=dq["idx"], q_val=dq["val"], q_sum=dq["sum"], q_latent=cd["q_latent"].numpy(),
                        psnr=np.array([10 * np.log10(4.0 / np.mean((rec[i] - x[i].numpy()) ** 2)) for i in range(3)]))
    print("kodak ok", bpp.numpy(), d["sum"])


def gen_kodak_eps(steps=1000):
    """epsilon-param counterpart of gen_kodak (BASELINE configs[2] step count on the reference's own images):
    BigCompressor + eps U-Net, compress(sample_mode="ddim"), 1000 steps, no clipping."""
    from PIL import Image
    tree = "epsilonparam"
    ref = import_reference(tree)
    _, kw, _, _, _, _ = CONFIGS["full_eps"]
    net = ref.unet.Unet(**kw)
    load_synth(net, seed=0, final_gain=0.2)
    ckw = ENCODER["encoder_full_eps"][2]
    comp = ref.cm.BigCompressor(**ckw)
    keep = ("enc.", "hyper_enc.", "hyper_dec.", "dec.", "prior.affine", "prior.a.")
    cman = [(k, list(v.shape)) for k, v in comp.state_dict().items() if k.startswith(keep)]
    csd = synth.unet_state_dict(cman, seed=15)
    comp.load_state_dict({k: torch.from_numpy(v) for k, v in csd.items()}, strict=False)
    comp.eval()
    crops = np.load(os.path.join(HERE, "kodak_x_500.npz"))["crops"]
    x = torch.from_numpy(crops).permute(0, 3, 1, 2).float() / 255.0 * 2.0 - 1.0
    diff = ref.dd.GaussianDiffusion(denoise_fn=net, context_fn=comp, **DIFF[tree])
    diff.eval()
    init = synth.normal("init", tuple(x.shape), seed=1, std=0.8)
    with torch.no_grad():
        cd = comp(x)
        rec, bpp = diff.compress(x, sample_steps=steps, sample_mode="ddim", bpp_return_mean=False,
                                 init=torch.from_numpy(init.copy()))
    rec = rec.numpy()
    d = digest(rec, nsample=256)
    np.savez_compressed(os.path.join(HERE, f"kodak_eps_{steps}.npz"), steps=np.array(steps), bpp=bpp.numpy(),
                        rec_idx=d["idx"], rec_val=d["val"], rec_sum=d["sum"], rec_sumsq=d["sumsq"],
                        q_latent=cd["q_latent"].numpy())
    print("kodak eps ok", bpp.numpy(), d["sum"], float(np.abs(rec).max()))


if __name__ == "__main__":
    torch.set_num_threads(8)
    if len(sys.argv) > 1:                 # python make_golden.py gen_decode_variants gen_heavy_tail ...: only these generators
        for fn in sys.argv[1:]:
            globals()[fn]()
        sys.exit(0)
    gen_schedules()
    gen_decode("small_x", [4, 1], eta_case=True)
    gen_decode("small_eps", [4], eta_case=True)
    gen_unet("odd_x")
    gen_decode("full_x", [3])
    gen_decode("full_eps", [3])
    gen_decode_variants()
    gen_full_res()
    gen_full_res_other("full_res_x_512", "full_x", 512, 512)
    gen_full_res_other("full_res_eps_256", "full_eps", 256, 256)
    for n in CTXDEC:
        gen_ctxdec(n)
    for n in HYPERDEC:
        gen_hyperdec(n)
    for n in ENCODER:
        gen_encoder(n)
    gen_kodak()          # ~15 min on 8 cores
    gen_kodak_eps()      # ~25 min
