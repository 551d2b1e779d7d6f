#!/usr/bin/env python3
"""VERDICT r3 item 8 (the clock lever), from the counter passes of tools/gpu_pmc_path.sh: per kernel the shader clock under its own
load (GRBM_GUI_ACTIVE summed over the 8 XCDs / 8 / the dispatch's duration in the same pass), the matrix-pipe-busy share and their
product relative to the 2.4 GHz the 2.5 PFLOP/s peak assumes.  usage: pmc_clock_table.py gpurun_out/pmc_r04 > profiles/pmc_r04_clock.txt"""
import collections, csv, glob, sys
out = sys.argv[1]


def short(n):
    return n.replace("void cdc::", "").replace("cdc::", "")[:64]


acc = collections.defaultdict(lambda: collections.defaultdict(float))
seen, durs = collections.defaultdict(set), collections.defaultdict(list)
for f in glob.glob(out + "/b/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen[k]:
            seen[k].add(r["Dispatch_Id"])
            durs[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = []
for k, c in acc.items():
    n = len(seen[k])
    if not c.get("GRBM_GUI_ACTIVE") or n == 0:
        continue
    d = sum(durs[k]) / n
    if d < 80e3:          # short launches: the busy counter includes the dispatch's ramp, the ratio says nothing about the clock
        continue
    clk = c["GRBM_GUI_ACTIVE"] / n / 8 / d
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (128.0 * c["GRBM_GUI_ACTIVE"])
    rows.append((busy, k, n, d / 1e3, clk))
print("kernels of one batch-32 iteration longer than 80 us, by matrix-pipe-busy share (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE):")
print(f"{'kernel (instantiation)':<66} {'launches':>8} {'avg us':>8} {'clock GHz':>9} {'MFMA busy':>9} {'busy x clock / 2.4':>18}")
for busy, k, n, d, clk in sorted(rows, reverse=True):
    print(f"{k:<66} {n:8d} {d:8.1f} {clk:9.2f} {busy:9.3f} {busy * clk / 2.4:18.3f}")
xs = [(b, c) for b, _, _, _, c in rows if b > 0.15]
if len(xs) >= 3:
    mb = sum(b for b, _ in xs) / len(xs); mc = sum(c for _, c in xs) / len(xs)
    slope = sum((b - mb) * (c - mc) for b, c in xs) / sum((b - mb) ** 2 for b, _ in xs)
    icpt = mc - slope * mb
    print(f"\nleast-squares over these kernels: clock = {icpt:.2f} {slope:+.2f} x busy GHz -> a pipe that never idles (busy = 1) would run at "
          f"{icpt + slope:.2f} GHz = {(icpt + slope) / 2.4:.2f} of the peak's clock")
