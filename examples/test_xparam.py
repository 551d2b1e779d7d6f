#!/usr/bin/env python3
"""Counterpart of the reference's xparam/test_xparam.py on the MI355X path: same arguments, same model
configuration (:29-61), same per-image loop (uint8/255*2-1, init = randn*gamma, clamp(-1,1)/2+.5, printed bpp).

    python examples/test_xparam.py --ckpt image-l2-use_weight5-vimeo-d64-t8193-b0.0032-x-cosine-01-float32-aux0.0_2.pt \\
        --lpips_weight 0.0 --n_denoise_step 65 --img_dir imgs --out_dir compressed_imgs

`--ckpt synthetic` runs with deterministic stand-in parameters (no network in the build environment).
Image sides must be multiples of 64 (five U-Net and six compressor down-samplings), as in the reference."""
import argparse

from _common import ema_model_state, load_checkpoint, run_folder, synthetic_state

import cdc_compression_amd as cdc

parser = argparse.ArgumentParser(description="x-parameterisation: decode a directory of images with the HIP path (reference counterpart: xparam/test_xparam.py)")
parser.add_argument("--ckpt", type=str, required=True)               # ckpt path, or "synthetic"
parser.add_argument("--gamma", type=float, default=0.8)              # noise intensity for decoding
parser.add_argument("--n_denoise_step", type=int, default=65)        # number of denoising steps
parser.add_argument("--device", type=int, default=0)                 # gpu device index
parser.add_argument("--img_dir", type=str, default="../imgs")
parser.add_argument("--out_dir", type=str, default="../compressed_imgs")
parser.add_argument("--lpips_weight", type=float, required=True)     # must match the ckpt (its LPIPS-VGG entries are skipped)
parser.add_argument("--seed", type=int, default=None)                # (extension) seed of the init noise


def main(config):
    rank = config.device
    denoise_model = cdc.Unet(dim=64, channels=3, context_channels=64, dim_mults=[1, 2, 3, 4, 5, 6],
                             context_dim_mults=[1, 2, 3, 4], embd_type="01", device=rank)
    context_model = cdc.ResnetCompressor(dim=64, dim_mults=[1, 2, 3, 4], reverse_dim_mults=[4, 3, 2, 1],
                                         hyper_dims_mults=[4, 4, 4], channels=3, out_channels=64, device=rank)
    diffusion = cdc.xparam.GaussianDiffusion(
        denoise_fn=denoise_model, context_fn=context_model, ae_fn=None, num_timesteps=8193, loss_type="l2",
        lagrangian=0.0032, pred_mode="x", aux_loss_weight=config.lpips_weight, aux_loss_type="lpips",
        var_schedule="cosine", use_loss_weight=True, loss_weight_min=5, use_aux_loss_weight_schedule=False)
    if config.ckpt == "synthetic":
        state = synthetic_state(diffusion)
    else:
        # the reference wraps the model in ema_pytorch.EMA, loads ckpt["ema"] and takes ema.ema_model (:62-68)
        state = ema_model_state(load_checkpoint(config.ckpt)["ema"])
    diffusion.load_state_dict(state)
    diffusion.to(rank)
    diffusion.eval()
    run_folder(diffusion, config, rank, dict(bpp_return_mean=True))


if __name__ == "__main__":
    main(parser.parse_args())
