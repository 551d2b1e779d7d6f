#!/usr/bin/env python3
"""Per-op and per-level time of one DDIM iteration from a `rocprofv3 --kernel-trace` CSV (no hipEvent pairs in the stream).

The per-op tables of rounds 1 - 5 (CDC_BENCH_OPS: one hipEvent pair around every launch) carry the cost of the pair itself, 3.7 - 4.4 us
per launch (profiles/launch_floor_r06.txt) -- a fifth of a few-pixel launch.  This tool takes the dispatches of a kernel trace instead:
the iterations are cut at the sampler kernel, the dispatches of an iteration are matched in program order to the launch program's op
labels (bench.py --dump-ops), and an op costs  end(op) - end(previous op)  -- its kernels plus the launch gap in front of them, which is
what the iteration pays for it.  Output: per-op table, per-class and per-level sums (mean over the matched iterations).

usage: trace_by_op.py <kernel_trace.csv> <ops.txt> [--batch B]
"""
import argparse
import collections
import csv
import re
import sys

FAMILY = {"PF3": "conv_pf3_kernel", "PF": "conv_pf_kernel", "PW": "conv_pw_kernel", "SPLIT2H": "conv_split2_kernel", "SPLIT2": "conv_split2_kernel",
          "SPLIT": "conv_split_kernel", "CONV": "conv_mfma_kernel", "WS": "conv_ws_kernel", "WS1": "conv_ws1_kernel"}


def short(name):
    name = re.sub(r"^void ", "", name).replace("cdc::", "").replace("(anonymous namespace)::", "")
    return name


def expect(label):
    t = label.split()
    k = t[0]
    if k == "conv":
        kern = next((x for x in ("PF3", "PF", "PW", "WS1", "WS", "SPLIT2H", "SPLIT2", "SPLIT") if x in t), "CONV")
        return (FAMILY[kern],), 1
    return {"ln": (("ln_kernel",), 1), "temb": (("copy_kernel", "temb_kernel"), 1), "copy": (("copy_kernel",), 1), "kvctx": (("kvctx",), 1),
            "ctxf": (("ctx_r", "fold_"), 0), "ctxp": (("ctx_partial",), 1), "ctx1": (("ctx_partial",), 1), "kstats": (("kmax_kernel",), 1),
            "pfpack": (("pf_pack_kernel", "c4_pack_kernel"), 1), "pfunpack": (("pf_unpack_kernel",), 1), "unfold": (("unfold_x_kernel",), 1),
            "ctxr": (("ctx_reduce_kernel",), 1), "lnconv": (("lnconv_kernel",), 1)}.get(k, ((k,), 1))


def level_of(label):
    m = re.search(r"out\s+(\d+)x(\d+)", label)
    if m:
        return int(m.group(2))
    m = re.search(r"(?:HW|N)=(\d+)", label)
    if m:
        return int(round(int(m.group(1)) ** 0.5))
    return 0


def cls_of(label):
    t = label.split()
    if t[0] != "conv":
        return {"ln": "layernorm", "kvctx": "attention", "ctxf": "attention", "ctxp": "attention", "ctx1": "attention", "kstats": "attention"}.get(t[0], "small")
    k, s = t[1], t[2]
    if "TZ4" in t or k == "2x2":
        return "transposed"
    if s == "s2":
        return "stride2"
    if k == "3x3":
        return "conv3x3"
    if k == "1x1":
        return "conv1x1"
    return "first/last"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("ops")
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    rows = []
    with open(a.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    labels = [l.rstrip("\n") for l in open(a.ops) if l.strip() and " HOIST" not in l and not l.startswith("combine")]
    ends = [i for i, r in enumerate(rows) if r[2].startswith("ddim_")]
    per_op = collections.defaultdict(list)
    n_ok = n_bad = 0
    bad_at = None
    for a0, a1 in zip(ends[:-1], ends[1:]):
        seq = rows[a0 + 1:a1 + 1]
        if rows[a1][0] - rows[a0][1] > 5e8 or len(seq) < len(labels):     # a gap of the host (first iterations, graph capture): skip
            continue
        i, prev_end, out, ok = 0, rows[a0][1], [], True
        for lab in labels:
            pref, cnt = expect(lab)
            if i >= len(seq) or not seq[i][2].startswith(pref):
                ok = False
                bad_at = (lab, seq[i][2] if i < len(seq) else "<end of iteration>")
                break
            j = i + 1
            if cnt == 0:
                while j < len(seq) and seq[j][2].startswith(pref):
                    j += 1
            out.append((lab, (seq[j - 1][1] - prev_end) * 1e-3, j - i))
            prev_end = seq[j - 1][1]
            i = j
        if not ok or i != len(seq) - 1:                                        # (the last dispatch is the sampler kernel)
            n_bad += 1
            continue
        out.append(("ddim (sampler update + 7-row combine)", (seq[-1][1] - prev_end) * 1e-3, 1))
        n_ok += 1
        for k, (lab, us, nk) in enumerate(out):
            per_op[(k, lab, nk)].append(us)
    print(f"# {a.csv}: {len(rows)} dispatches, {len(ends)} sampler kernels, {n_ok} iterations matched to {len(labels)} op labels ({n_bad} not matched)")
    if not n_ok:
        print(f"# first mismatch: op label '{bad_at[0]}' against dispatch '{bad_at[1][:100]}'" if bad_at else "# no complete iteration in the trace")
        if len(ends) > 3:          # the kernel names of one iteration beside the labels, for the eye
            seq = rows[ends[-3] + 1:ends[-2] + 1]
            for k in range(max(len(seq), len(labels))):
                print(f"{(seq[k][2][:60] if k < len(seq) else ''):62s} | {labels[k] if k < len(labels) else ''}")
        return 1
    tot = 0.0
    by_cls, by_lvl, by_lvl_n = collections.Counter(), collections.Counter(), collections.Counter()
    lines = []
    lvl = 256
    for (k, lab, nk), v in sorted(per_op.items()):
        us = sum(v) / len(v)
        tot += us
        l = level_of(lab)
        lvl = l or lvl
        c = cls_of(lab) if not lab.startswith("ddim") else "small"
        by_cls[c] += us
        by_lvl[lvl] += us
        by_lvl_n[lvl] += nk
        lines.append(f"{us:9.2f} us  {nk} kernel{'s' if nk > 1 else ' '}  {lab}")
    print(f"# one DDIM iteration, batch {a.batch}: {tot / 1e3:.3f} ms = sum of (end of the op's last kernel - end of the previous op), {sum(by_lvl_n.values())} kernels")
    print("# by class (ms):  " + "  ".join(f"{c} {v / 1e3:.3f}" for c, v in sorted(by_cls.items(), key=lambda kv: -kv[1])))
    print("# by level (map width: ms, kernels):  " + "  ".join(f"{l}: {by_lvl[l] / 1e3:.3f} ({by_lvl_n[l]})" for l in sorted(by_lvl, reverse=True)))
    print(f"# levels <= 16 wide: {sum(v for l, v in by_lvl.items() if l <= 16) / 1e3:.3f} ms, {sum(n for l, n in by_lvl_n.items() if l <= 16)} kernels")
    for l in lines:
        print(l)
    return 0


if __name__ == "__main__":
    sys.exit(main())
