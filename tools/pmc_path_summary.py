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
tag = sys.argv[3] if len(sys.argv) > 3 else "r03"            # round tag of the output files
ops_file = sys.argv[4] if len(sys.argv) > 4 else None        # bench.py --dump-ops: the launch program's op labels in program order
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
# (not the path: bench.py's own ceiling probes, csrc/probe.hip, which run in the same process since round 5)
tot_fetch = sum(v.get("FETCH_SIZE", 0) for k, v in A.items() if "probe_" not in k) * 1024
tot_write = sum(v.get("WRITE_SIZE", 0) for k, v in Wc.items() if "probe_" not in k) * 1024
iters = steps + 2
per_iter_corr, per_iter_raw = (2 * tot_fetch + tot_write) / iters, (tot_fetch + tot_write) / iters
print(f"\nwhole path: FETCH {tot_fetch / 1e9:.2f} GB (raw) + WRITE {tot_write / 1e9:.2f} GB over {iters} DDIM iterations of {B} images")
print(f"  per image-iteration: {per_iter_corr / B / 1e9:.3f} GB (FETCH x2) | {per_iter_raw / B / 1e9:.3f} GB (raw)   vs canonical 0.714 GB (SURVEY 8(d): every conv input once + output once, fp32)")
json.dump({"steps": steps, "batch": B, "kernels": rows, "hbm_gb_per_image_iter_corrected": per_iter_corr / B / 1e9,
           "hbm_gb_per_image_iter_raw": per_iter_raw / B / 1e9, "canonical_gb_per_image_iter": 0.714,
           "note": "FETCH_SIZE x2 is the guide's gfx950 correction for wide coalesced reads; 4-byte accesses are uncalibrated, so the "
                   "truth lies between the raw and the corrected figure"},
          open(os.path.join(out, f"pmc_{tag}_path.json"), "w"), indent=1)


# ---- per-op traffic (round 4): the convolution dispatches of the LAST DDIM iteration of the FETCH_SIZE / WRITE_SIZE passes, matched
# in program order to the launch program's op labels (every convolution op is exactly one dispatch of its kernel family; an iteration
# ends with the sampler kernel).  bench.py averages these over the launches of its dominant (layer shape, kernel) pair, so that
# `roofline.traffic` and `roofline.algorithmic_bytes_per_launch` describe the same launches.
def dispatches(d, counter):
    per = collections.OrderedDict()
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = int(r["Dispatch_Id"])
            e = per.setdefault(k, [short(r["Kernel_Name"]), 0.0])
            e[1] += float(r["Counter_Value"])
    return [per[k] for k in sorted(per)]


def last_iteration(seq):
    ends = [i for i, (k, _) in enumerate(seq) if k.startswith("ddim_")]          # ddim_kernel / ddim_rows4_kernel
    if len(ends) < 2:
        return []
    return seq[ends[-2] + 1: ends[-1] + 1]


FAMILY = {"PF3": "conv_pf3_kernel", "PF": "conv_pf_kernel", "PW": "conv_pw_kernel", "SPLIT2H": "conv_split2_kernel", "SPLIT2": "conv_split2_kernel",
          "SPLIT": "conv_split_kernel", "CONV": "conv_mfma_kernel", "WS": "conv_ws_kernel", "WS1": "conv_ws1_kernel"}


def kern_of(label):
    t = label.split()
    return next((k for k in ("PF3", "PF", "PW", "WS1", "WS", "SPLIT2H", "SPLIT2", "SPLIT") if k in t), "CONV")


if ops_file and os.path.exists(ops_file):
    labels = [l.rstrip("\n") for l in open(ops_file) if l.startswith("conv ") and " HOIST" not in l]
    fa, wr = last_iteration(dispatches("a", "FETCH_SIZE")), last_iteration(dispatches("w", "WRITE_SIZE"))
    conv_a = [x for x in fa if x[0].startswith("conv_")]
    conv_w = [x for x in wr if x[0].startswith("conv_")]
    ops, ok = {}, len(conv_a) == len(labels) == len(conv_w)
    if ok:
        for lab, (ka, f), (kw, w) in zip(labels, conv_a, conv_w):
            if not (ka.startswith(FAMILY[kern_of(lab)]) and kw.startswith(FAMILY[kern_of(lab)])):
                ok = False
                break
            e = ops.setdefault(lab, dict(n=0, fetch=0.0, write=0.0))
            e["n"] += 1; e["fetch"] += f * 1024; e["write"] += w * 1024
    if ok:
        res = {lab: {"launches": e["n"], "fetch_bytes_raw": e["fetch"] / e["n"], "write_bytes": e["write"] / e["n"],
                     "hbm_bytes_corrected": (2 * e["fetch"] + e["write"]) / e["n"], "hbm_bytes_raw": (e["fetch"] + e["write"]) / e["n"]}
               for lab, e in ops.items()}
        arith = "bf16x3" if os.environ.get("CDC_ARITH") == "0" else "f16x2"
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
        from cdc_compression_amd._lib import kernel_source_hash
        json.dump({"batch": B, "arith": arith, "ops": res, "kernel_source_hash": kernel_source_hash(),
                   "source": f"profiles/pmc_{tag}_path.* passes (rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE over bench.py, separate passes, "
                             "FETCH_SIZE x2 gfx950 correction): the convolution dispatches of the last DDIM iteration matched in program order to the op labels"},
                  open(os.path.join(out, f"pmc_{tag}_traffic.json"), "w"), indent=1)
        print(f"\nper-op traffic: {len(labels)} convolution launches matched -> pmc_{tag}_traffic.json")
    else:
        print(f"\nper-op traffic: could not match {len(labels)} op labels to {len(conv_a)} / {len(conv_w)} convolution dispatches", file=sys.stderr)
