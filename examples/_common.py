"""Shared pieces of the two inference scripts (counterparts of the reference's test_xparam.py / test_epsilonparam.py):
image IO without torchvision, checkpoint unwrapping, the per-image loop."""
import os
import pathlib
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def read_image(path, device):
    """torchvision.io.read_image(path).unsqueeze(0).float() / 255  ->  [1, 3, H, W] in [0, 1]."""
    import torch
    from PIL import Image
    a = np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8)
    return torch.from_numpy(a.transpose(2, 0, 1).copy()).unsqueeze(0).float().to(device) / 255.0


def save_image(t, path):
    """torchvision.utils.save_image for one [1, 3, H, W] tensor in [0, 1]: x*255 + 0.5, clamp, uint8."""
    from PIL import Image
    a = (t[0].detach().float().cpu() * 255.0 + 0.5).clamp(0, 255).to(dtype=__import__("torch").uint8).numpy()
    Image.fromarray(a.transpose(1, 2, 0)).save(path)


def ema_model_state(ema_state):
    """State of `ema.ema_model` from an ema_pytorch.EMA state_dict (test_xparam.py:62-68 loads the EMA wrapper and
    then uses its `ema_model`): keys "ema_model.<name>"; "online_model.*", "initted", "step" are the wrapper's."""
    out = {k[len("ema_model."):]: v for k, v in ema_state.items() if k.startswith("ema_model.")}
    if not out:
        raise KeyError('no "ema_model.*" entries: is this an ema_pytorch.EMA state_dict?')
    return out


def load_checkpoint(path):
    import torch
    return torch.load(path, map_location="cpu", weights_only=False)


def synthetic_state(diffusion, seed_unet=0, seed_ctx=15, eps=False):
    """Deterministic stand-in parameters (there is no network for the published checkpoints): the generator the
    parity fixtures use (cdc_compression_amd.synth)."""
    from cdc_compression_amd import synth
    sd = {}
    for k, v in synth.unet_state_dict(diffusion.denoise_fn.manifest(), seed=seed_unet,
                                      final_gain=0.2 if eps else 1.0).items():
        sd["denoise_fn." + k] = v
    comp = diffusion.context_fn
    man = comp.manifest() + comp.hyper_manifest() + comp.encoder_manifest()
    csd = synth.unet_state_dict(man, seed=seed_ctx)
    C0 = comp.reversed_hyper_dims[0]
    pd = (1, 3, 3, 3, 1)
    for i in range(4):
        csd[f"prior.affine.{i}.weight"] = synth.normal(f"pw{i}", (C0, 1, 1, pd[i], pd[i + 1]), seed_ctx, 1.0)
        csd[f"prior.affine.{i}.bias"] = synth.normal(f"pb{i}", (C0, 1, 1, 1, pd[i + 1]), seed_ctx, 0.1)
        if i < 3:
            csd[f"prior.a.{i}"] = synth.normal(f"pa{i}", (C0, 1, 1, 1, pd[i + 1]), seed_ctx, 0.5)
    csd["prior._medians"] = np.zeros((1, C0, 1, 1), np.float32)
    for k, v in csd.items():
        sd["context_fn." + k] = v
    return sd


def run_folder(diffusion, config, rank, compress_kwargs):
    """The per-image loop of both reference scripts (test_xparam.py:72-84 / test_epsilonparam.py:67-80)."""
    import torch
    if getattr(config, "seed", None) is not None:
        torch.manual_seed(config.seed)
    for img in sorted(os.listdir(config.img_dir)):
        if img.endswith(".png") or img.endswith(".jpg"):
            to_be_compressed = read_image(os.path.join(config.img_dir, img), rank)
            compressed, bpp = diffusion.compress(
                to_be_compressed * 2.0 - 1.0,
                sample_steps=config.n_denoise_step,
                init=torch.randn_like(to_be_compressed) * config.gamma,
                **compress_kwargs,
            )
            compressed = compressed.clamp(-1, 1) / 2.0 + 0.5
            pathlib.Path(config.out_dir).mkdir(parents=True, exist_ok=True)
            save_image(compressed.cpu(), os.path.join(config.out_dir, img))
            print("image:", img)
            print("bpp:", bpp)
