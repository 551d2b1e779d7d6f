"""Batch sharding of the decode over the GPUs of one node.

Every image is an independent DDIM chain (no cross-image operation anywhere on the path), so the batch
is split into contiguous shards, one process + one libcdc_hip handle per GPU, weights replicated.  There
is NO collective on the data path; the only exchange is the final gather of the decoded images
(`torch.distributed.all_gather`: RCCL over xGMI with the "nccl" backend, gloo on CPU in the tests).
"""


def shard_bounds(batch, world_size, rank):
    """Contiguous, balanced [lo, hi) of `batch` images for `rank` (first `batch % world` ranks get +1)."""
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside world of {world_size}")
    base, extra = divmod(batch, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def sharded_decode(decode_fn, init, context, world_size=1, rank=0, dist=None, global_batch=None):
    """Run `decode_fn(init_shard, context_shard) -> reconstruction shard` on this rank's slice of the
    batch and gather the full batch on every rank.

    init: [B, C, H, W] tensor or None; context: list of [B, C_l, H_l, W_l] tensors.  With
    world_size == 1 (or dist None) this is a plain call.  Shards may be ragged (B % world != 0):
    they are padded to the largest shard for the fixed-size all_gather and trimmed afterwards.
    global_batch: the inputs already ARE this rank's shard of a `global_batch`-image job (each rank
    generated / loaded only its own images); the shard sizes must then follow shard_bounds."""
    if global_batch is None:
        B = context[0].shape[0]
        lo, hi = shard_bounds(B, world_size, rank)
        sl = slice(lo, hi)
        rec = decode_fn(None if init is None else init[sl], [c[sl] for c in context])
    else:
        B = int(global_batch)
        lo, hi = shard_bounds(B, world_size, rank)
        if context[0].shape[0] != hi - lo:
            raise ValueError(f"rank {rank} holds {context[0].shape[0]} images, its shard of {B} is {hi - lo}")
        rec = decode_fn(init, list(context))
    if dist is None:                  # (a one-rank process group still runs the gather: bench.py under torchrun at N=1)
        return rec
    import torch
    maxn = -(-B // world_size)
    pad = torch.zeros((maxn,) + tuple(rec.shape[1:]), dtype=rec.dtype, device=rec.device)
    pad[: hi - lo] = rec
    parts = [torch.empty_like(pad) for _ in range(world_size)]
    dist.all_gather(parts, pad)
    out = []
    for r in range(world_size):
        rlo, rhi = shard_bounds(B, world_size, r)
        out.append(parts[r][: rhi - rlo])
    return torch.cat(out, dim=0)
