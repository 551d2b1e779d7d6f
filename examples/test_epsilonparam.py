#!/usr/bin/env python3
"""Counterpart of the reference's epsilonparam/test_epsilonparam.py on the MI355X path: same arguments, same model
configuration (:27-56), `compress(..., sample_mode="ddim", bpp_return_mean=False)` per image (:67-80).

`--ckpt synthetic` runs with deterministic stand-in parameters (no network in the build environment)."""
import argparse

from _common import load_checkpoint, run_folder, synthetic_state

import cdc_compression_amd as cdc

parser = argparse.ArgumentParser(description="epsilon-parameterisation: decode a directory of images with the HIP path (reference counterpart: epsilonparam/test_epsilonparam.py)")
parser.add_argument("--ckpt", type=str, required=True)               # ckpt path, or "synthetic"
parser.add_argument("--gamma", type=float, default=0.8)
parser.add_argument("--n_denoise_step", type=int, default=200)
parser.add_argument("--device", type=int, default=0)
parser.add_argument("--img_dir", type=str, default="../imgs")
parser.add_argument("--out_dir", type=str, default="../compressed_imgs")
parser.add_argument("--lpips_weight", type=float, required=True)
parser.add_argument("--seed", type=int, default=None)                # (extension) seed of the init noise


def main(args):
    rank = args.device
    denoise_model = cdc.Unet(dim=64, channels=3, context_channels=3, dim_mults=(1, 2, 3, 4, 5, 6),
                             context_dim_mults=(1, 2, 3, 4), device=rank)
    context_model = cdc.BigCompressor(dim=64, dim_mults=(1, 2, 3, 4), hyper_dims_mults=(4, 4, 4), channels=3,
                                      out_channels=3, vbr=False, device=rank)
    diffusion = cdc.epsilonparam.GaussianDiffusion(
        denoise_fn=denoise_model, context_fn=context_model, num_timesteps=20000, loss_type="l1", clip_noise="none",
        vbr=False, lagrangian=0.9, pred_mode="noise", var_schedule="linear", aux_loss_weight=args.lpips_weight,
        aux_loss_type="lpips")
    if args.ckpt == "synthetic":
        state = synthetic_state(diffusion, eps=True)
    else:
        state = load_checkpoint(args.ckpt)["model"]                  # :63
    diffusion.load_state_dict(state)
    diffusion.to(rank)
    diffusion.eval()
    run_folder(diffusion, args, rank, dict(sample_mode="ddim", bpp_return_mean=False))


if __name__ == "__main__":
    main(parser.parse_args())
