#!/usr/bin/env python3
"""ms per DDIM iteration of one decode (unprofiled wall clock), for A/B of process-level switches:
    gpu_b1_ab.py --batch 1 --sample-steps 200 [--side-stream]      (env: CDC_DEV=1 CDC_DEV_REPEAT=n, CDC_GRAPH=0/1, ...)
prints one line: label, batch, ms per iteration (best of --reps decodes)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--sample-steps", type=int, default=200)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--side-stream", action="store_true")
    ap.add_argument("--label", default="")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    un, diff, cfgd = bench.build_model("x", 0)
    init, ctx, gen = bench.make_inputs(cfgd, a.batch, 256, dev, 1000)
    st = torch.cuda.Stream() if a.side_stream else torch.cuda.current_stream()
    best = 1e9
    with torch.cuda.stream(st):
        diff.decompress(ctx, (a.batch, 3, 256, 256), sample_steps=3, init=init)
        torch.cuda.synchronize()
        for _ in range(a.reps):
            t0 = time.perf_counter()
            diff.decompress(ctx, (a.batch, 3, 256, 256), sample_steps=a.sample_steps, init=init)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / a.sample_steps * 1e3)
    env = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("CDC_") and k != "CDC_DEV")
    print(f"batch {a.batch:3d}  {best:8.4f} ms/iteration   stream={'side' if a.side_stream else 'null'}  {env}  {a.label}", flush=True)


if __name__ == "__main__":
    main()
