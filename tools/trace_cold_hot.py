#!/usr/bin/env python3
"""Cold against hot launches of the library's own kernels.

Input: a `rocprofv3 --kernel-trace` CSV of a run with CDC_DEV=1 CDC_DEV_REPEAT=n (every op of the launch program issued n times
in a row).  Inside a hipGraph replay / a deep queue the profiler's start stamp of a dispatch is the end stamp of its predecessor,
so `duration` = the launch's whole cost in place, boundary included.  For every op: duration of the FIRST of its n launches
(cold: code, operands, argument block as the program leaves them) and the mean of launches 2..n (hot).  Aggregated per kernel.

usage: trace_cold_hot.py <kernel_trace.csv> n [--iters-skip K]
"""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("cdc::", "").replace("(anonymous namespace)::", "")
    return re.sub(r"^void ", "", name)[:80]


def main():
    path, n = sys.argv[1], int(sys.argv[2])
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    names = [r[2] for r in rows]
    dur = [(r[1] - r[0]) * 1e-3 for r in rows]
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    i, ops, cold_sum, hot_sum = 0, 0, 0.0, 0.0
    while i < len(rows):
        hit = False
        for p in (1, 2, 3):
            if i + p * n > len(rows):
                continue
            pat = names[i:i + p]
            if p > 1 and len(set(pat)) == 1:
                continue
            if all(names[i + k * p:i + (k + 1) * p] == pat for k in range(n)) and (i + p * n >= len(rows) or names[i + p * n:i + p * n + p] != pat or p == 1):
                for q in range(p):
                    c = dur[i + q]
                    h = sum(dur[i + k * p + q] for k in range(1, n)) / (n - 1)
                    a = agg[pat[q]]
                    a[0] += 1; a[1] += c; a[2] += h
                    cold_sum += c; hot_sum += h
                ops += p
                i += p * n
                hit = True
                break
        if not hit:
            i += 1
    print(f"# {path}: {len(rows)} dispatches, {ops} ops matched as {n} repeats")
    print(f"# sum over ops: cold {cold_sum / 1e3:.3f} ms, hot {hot_sum / 1e3:.3f} ms  (cold - hot = {(cold_sum - hot_sum) / max(ops, 1):.2f} us per op)")
    print(f"{'kernel':80s} {'ops':>6s} {'cold us':>8s} {'hot us':>8s} {'diff':>7s}")
    for nm, (c, a, b) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{nm:80s} {c:6d} {a / c:8.2f} {b / c:8.2f} {(a - b) / c:7.2f}")


if __name__ == "__main__":
    main()
