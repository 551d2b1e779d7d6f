#!/usr/bin/env python3
"""Summarise the rocprofv3 passes of tools/gpu_pmc_path.sh: per kernel (text: those above 2 % of the GPU time; json: above 0.2 %) the average
duration, HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md; raw value
beside it), MFMA-busy share of the SIMD-busy cycles, MFMA instructions, LDS bank-conflict share; and the whole-iteration
HBM byte total next to SURVEY 8(d)'s canonical 0.714 GB per image-step."""
import collections
import csv
import glob
import json
import os
import sys

out, steps = sys.argv[1], int(sys.argv[2])
B = 32


def short(name):
    name = name.replace("void cdc::", "").replace("cdc::", "")
    return name[:110]


def counters(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = short(r.get("Kernel_Name", ""))
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (r.get("Dispatch_Id"), k)
            if key not in seen:
                seen.add(key)
                n[k] += 1
    return acc, n


stats = {}
for f in glob.glob(os.path.join(out, "t", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        stats[short(r["Name"])] = dict(calls=int(r["Calls"]), total_ns=float(r["TotalDurationNs"]), avg_ns=float(r["AverageNs"]), pct=float(r["Percentage"]))
A, nA = counters("a")
Wc, nW = counters("w")
Bc, nB = counters("b")
C, nC = counters("c")
tot_ns = sum(v["total_ns"] for v in stats.values())
print(f"rocprofv3 passes over `bench.py --sample-steps {steps}` (batch {B}, 256x256, x-param): {len(stats)} kernels, {tot_ns / 1e6:.1f} ms of GPU time")
print(f"{'kernel':<112} {'calls':>6} {'avg us':>9} {'% time':>7} {'HBM MB/launch (x2 corr | raw)':>32} {'TB/s':>6} {'MFMA busy':>10} {'MFMA insts':>11} {'LDS confl':>9}")
rows = []
for k, s in sorted(stats.items(), key=lambda kv: -kv[1]["total_ns"]):
    if s["pct"] < 0.2:
        continue
    a, b, c = A.get(k, {}), Bc.get(k, {}), C.get(k, {})
    na, nb, nc = max(nA.get(k, 0), 1), max(nB.get(k, 0), 1), max(nC.get(k, 0), 1)
    fetch, write = a.get("FETCH_SIZE", 0) / na * 1024, Wc.get(k, {}).get("WRITE_SIZE", 0) / max(nW.get(k, 0), 1) * 1024
    corr, raw = 2 * fetch + write, fetch + write
    # SQ_VALU_MFMA_BUSY_CYCLES = matrix-pipe cycles summed over the chip's 1024 SIMDs (32 per v_mfma_f32_32x32x16); GRBM_GUI_ACTIVE =
    # busy cycles summed over the 8 XCDs: share of the chip's matrix-pipe time = MFMA_BUSY / (1024 * GUI_ACTIVE / 8)
    busy = b.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (128.0 * b["GRBM_GUI_ACTIVE"]) if b.get("GRBM_GUI_ACTIVE") else float("nan")
    confl = c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"] if c.get("SQ_LDS_IDX_ACTIVE") else float("nan")
    tbs = corr / (s["avg_ns"] * 1e-9) / 1e12
    rows.append(dict(kernel=k, calls=s["calls"], avg_us=s["avg_ns"] / 1e3, pct=s["pct"], hbm_mb_corr=corr / 1e6, hbm_mb_raw=raw / 1e6,
                     tb_per_s=tbs, mfma_busy=busy, mfma_insts=b.get("SQ_INSTS_MFMA", 0) / nb, lds_conflict=confl))
    if s["pct"] >= 2.0:
        print(f"{k:<112} {s['calls']:>6} {s['avg_ns'] / 1e3:>9.1f} {s['pct']:>7.2f} {corr / 1e6:>16.1f} | {raw / 1e6:>11.1f} {tbs:>6.2f} {busy:>10.3f} {b.get('SQ_INSTS_MFMA', 0) / nb:>11.0f} {confl:>9.3f}")
# whole path: every dispatch of the pass / number of DDIM iterations (the 2-iteration build decode and the context pre-pass are in: upper bound)
tot_fetch = sum(v.get("FETCH_SIZE", 0) for v in A.values()) * 1024
tot_write = sum(v.get("WRITE_SIZE", 0) for v in Wc.values()) * 1024
iters = steps + 2
per_iter_corr, per_iter_raw = (2 * tot_fetch + tot_write) / iters, (tot_fetch + tot_write) / iters
print(f"\nwhole path: FETCH {tot_fetch / 1e9:.2f} GB (raw) + WRITE {tot_write / 1e9:.2f} GB over {iters} DDIM iterations of {B} images")
print(f"  per image-iteration: {per_iter_corr / B / 1e9:.3f} GB (FETCH x2) | {per_iter_raw / B / 1e9:.3f} GB (raw)   vs canonical 0.714 GB (SURVEY 8(d): every conv input once + output once, fp32)")
json.dump({"steps": steps, "batch": B, "kernels": rows, "hbm_gb_per_image_iter_corrected": per_iter_corr / B / 1e9,
           "hbm_gb_per_image_iter_raw": per_iter_raw / B / 1e9, "canonical_gb_per_image_iter": 0.714,
           "note": "FETCH_SIZE x2 is the guide's gfx950 correction for wide coalesced reads; 4-byte accesses are uncalibrated, so the "
                   "truth lies between the raw and the corrected figure"},
          open(os.path.join(out, "pmc_r03_path.json"), "w"), indent=1)
