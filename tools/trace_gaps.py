#!/usr/bin/env python3
"""Dependent-launch gaps of the library's own launches, from a `rocprofv3 --kernel-trace` CSV.

For every pair of consecutive dispatches on the same queue: gap = start[i+1] - end[i].  Prints per kernel name (templates
shortened): launches, mean duration, mean gap BEFORE it (its start - predecessor's end) and mean gap AFTER it, then the totals:
sum of durations, sum of gaps, span.  `--skip-first N` drops the warm-up part of the trace, `--only-steady` keeps the longest
run of dispatches whose gaps stay below 50 us (= inside one decode).   Development aid.

usage: trace_gaps.py <kernel_trace.csv> [--top 40] [--skip-first N]
"""
import argparse
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("cdc::", "").replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    return name[:90]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--top", type=int, default=45)
    ap.add_argument("--skip-first", type=int, default=0)
    ap.add_argument("--break-us", type=float, default=50.0)
    a = ap.parse_args()
    rows = []
    with open(a.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
    rows.sort()
    rows = rows[a.skip_first:]
    # longest steady run
    runs, cur = [], [0]
    for i in range(1, len(rows)):
        if rows[i][3] == rows[i - 1][3] and (rows[i][0] - rows[i - 1][1]) < a.break_us * 1e3:
            cur.append(i)
        else:
            runs.append(cur); cur = [i]
    runs.append(cur)
    runs.sort(key=len, reverse=True)
    total_in_runs = 0
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    dur_sum = gap_sum = span = 0.0
    n_gaps = 0
    gaps_all = []
    for run in runs:
        if len(run) < 20:
            continue
        total_in_runs += len(run)
        span += (rows[run[-1]][1] - rows[run[0]][0]) * 1e-3
        for j, i in enumerate(run):
            s, e, nm, _ = rows[i]
            k = agg[short(nm)]
            k[0] += 1; k[1] += (e - s) * 1e-3
            dur_sum += (e - s) * 1e-3
            if j > 0:
                g = (s - rows[run[j - 1]][1]) * 1e-3
                k[2] += g; gap_sum += g; n_gaps += 1; gaps_all.append(g)
            if j + 1 < len(run):
                k[3] += (rows[run[j + 1]][0] - e) * 1e-3
    print(f"# {a.csv}: {len(rows)} dispatches, {total_in_runs} in steady runs (gap < {a.break_us} us), {n_gaps} gaps")
    if not n_gaps:
        return 1
    gaps_all.sort()
    print(f"# sum of kernel durations {dur_sum / 1e3:.3f} ms, sum of gaps {gap_sum / 1e3:.3f} ms ({100 * gap_sum / span:.1f} % of the span {span / 1e3:.3f} ms)")
    print(f"# gap: mean {gap_sum / n_gaps:.2f} us, p10 {gaps_all[n_gaps // 10]:.2f}, median {gaps_all[n_gaps // 2]:.2f}, p90 {gaps_all[n_gaps * 9 // 10]:.2f}, min {gaps_all[0]:.2f}")
    print(f"{'kernel':90s} {'n':>7s} {'dur us':>8s} {'gap before':>10s} {'gap after':>10s} {'sum ms':>8s}")
    for nm, (n, d, gb, ga) in sorted(agg.items(), key=lambda kv: -kv[1][1])[: a.top]:
        print(f"{nm:90s} {n:7d} {d / n:8.2f} {gb / n:10.2f} {ga / n:10.2f} {d / 1e3:8.3f}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
