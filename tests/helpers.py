"""Shared helpers for the parity tests (oracle-side setup of the golden cases)."""
import json
import os

import numpy as np

from cdc_compression_amd import synth
from oracle import model as om

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    """Golden U-Net case -> (cfg kwargs, manifest, state_dict, x, time, ctx list, golden npz)."""
    mj = json.load(open(os.path.join(GOLDEN, f"manifest_{name}.json")))
    kw = dict(mj["unet_kwargs"])
    kw.pop("embd_type", None)
    man = [(a, tuple(b)) for a, b in mj["manifest"]]
    sd = synth.unet_state_dict(man, seed=0, final_gain=0.2 if "eps" in name else 1.0)
    g = np.load(os.path.join(GOLDEN, f"unet_{name}.npz"))
    B, H, W = int(g["B"]), int(g["H"]), int(g["W"])
    x = synth.normal("x", (B, 3, H, W), seed=1, std=0.8)
    ctx = synth.context_pyramid(mj["context_channels_per_level"], B, H, W, seed=3)
    return kw, man, sd, x, g["time"], ctx, g


def oracle_cfg(kw):
    return om.UnetConfig(**kw)


def digest_idx(nsample, size, seed=11):
    return (synth._splitmix64(np.arange(nsample, dtype=np.uint64) + np.uint64(seed * 1000))
            % np.uint64(size)).astype(np.int64)
