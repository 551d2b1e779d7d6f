"""Deterministic synthetic weights / inputs for tests and bench (no checkpoints offline).

Counter-based generator built only from exact operations (uint64 hashing, integer sums, one
float64 multiply, one cast) so that this container, the GPU box and any future host produce
bit-identical float32 tensors -- unlike torch's or numpy's normal samplers, whose libm paths can
differ by an ulp between CPUs.  Used to inject identical parameters into the real reference
(tests/golden/make_golden.py), the CPU oracle and the HIP path.
"""
import numpy as np

_MASK = (1 << 64) - 1
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _fnv1a(s):
    h = 0xCBF29CE484222325
    for ch in s.encode():
        h ^= ch
        h = (h * 0x100000001B3) & _MASK
    return h


def _splitmix64(z):
    with np.errstate(over="ignore"):
        z = (z + _GOLD)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


_STD4 = float(np.sqrt(4.0 * (65536.0 ** 2 - 1.0) / 12.0))


def normal(name, shape, seed=0, std=1.0, mean=0.0):
    """Approximately N(mean, std^2) float32 tensor (Irwin-Hall n=4), a pure function of
    (name, seed, element index)."""
    n = int(np.prod(shape)) if len(shape) else 1
    base = np.uint64((_fnv1a(name) ^ (seed * 0xD6E8FEB86659FD93)) & _MASK)
    with np.errstate(over="ignore"):
        ctr = base + np.arange(n, dtype=np.uint64) * _GOLD
    r = _splitmix64(ctr)
    m = np.uint64(0xFFFF)
    s = ((r & m) + ((r >> np.uint64(16)) & m) + ((r >> np.uint64(32)) & m)
         + ((r >> np.uint64(48)) & m)).astype(np.int64) - 2 * 65535
    z = s.astype(np.float64) * (float(std) / _STD4) + float(mean)
    return z.astype(np.float32).reshape(shape)


def unet_state_dict(manifest, seed=0, final_gain=1.0):
    """Synthetic parameters for a Unet manifest [(name, shape), ...].

    Gains: conv/linear weights N(0, 1/fan_in); biases N(0, 0.1^2); LayerNorm g = 1 + N(0, 0.2^2),
    b = N(0, 0.2^2); to_qkv weights x2 (so the k-softmax is not flat); final conv x final_gain."""
    sd = {}
    for name, shape in manifest:
        shape = tuple(shape)
        if name.endswith(".g"):
            t = normal(name, shape, seed, 0.2, 1.0)
        elif name.endswith(".b"):
            t = normal(name, shape, seed, 0.2)
        elif name.endswith(".bias"):
            t = normal(name, shape, seed, 0.1)
        else:
            if name.endswith(".conv.weight") and len(shape) == 4 and shape[-1] == 4:
                fan_in = shape[0] * 4          # ConvTranspose2d [Cin][Cout][4][4], 2x2 taps/output
            else:
                fan_in = int(np.prod(shape[1:]))
            std = 1.0 / np.sqrt(max(fan_in, 1))
            if "to_qkv" in name:
                std *= 2.0
            if name.startswith("final_conv.1") or name.endswith("final_conv.1.weight"):
                std *= final_gain
            if name == "time_mlp.0.weight":
                std = 1.0
            t = normal(name, shape, seed, std)
        sd[name] = t
    return sd


def context_pyramid(cfg_context_channels, B, H, W, seed=3, std=0.5):
    """N(0, std^2) stand-in for `context_fn.decode(q_latent)`: list of [B, C_l, H/2^l, W/2^l]."""
    return [normal(f"ctx{l}", (B, c, H >> l, W >> l), seed, std)
            for l, c in enumerate(cfg_context_channels)]
