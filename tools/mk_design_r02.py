#!/usr/bin/env python3
"""Development aid: regenerate the 'Round 2' measurement block of DESIGN.md section 5 from profiles/*_r02*."""
import json, re, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
def last(f): return json.loads(open(f).read().strip().split('\n')[-1])
j = last('profiles/bench_r02.json'); r = j['roofline']
ops = [l.split(None, 5) for l in open('profiles/per_op_r02.txt')]
def agg(pred):
    sel = [(float(o[1]), float(o[3])) for o in ops if pred(o[5])]
    ms = sum(x[0] for x in sel); fl = sum(x[0] * x[1] for x in sel)
    return len(sel), ms, (fl / ms if ms else 0)
def res(lab):
    m = re.search(r'out\s+(\d+)x(\d+)', lab); return int(m.group(1)) if m else 0
rows = [
 ('3×3 s1 on `conv_pf_kernel` (pre-split planes)', lambda l: 'conv 3x3 s1' in l and ' PF' in l),
 ('3×3 s1 on `conv_split2_kernel`, ≥ 64² (the 128² decoder concat)', lambda l: 'conv 3x3 s1' in l and ' PF' not in l and res(l) >= 64),
 ('3×3 s1 on `conv_split2_kernel`, < 64² (16², 8²: split-K)', lambda l: 'conv 3x3 s1' in l and ' PF' not in l and res(l) < 64),
 ('1×1 on `conv_pw_kernel` / `conv_pf_kernel`', lambda l: 'conv 1x1' in l and (' PW' in l or ' PF' in l)),
 ('1×1 on `conv_split2_kernel` (small launches, per-image weights at 16² / 8²)', lambda l: 'conv 1x1' in l and ' PW' not in l and ' PF' not in l),
 ('`kvctx16_kernel` (fused attention front)', lambda l: l.strip().startswith('kvctx')),
 ('ConvTranspose (4 phases of 2×2)', lambda l: 'conv 2x2' in l),
 ('3×3 s2 (Downsample)', lambda l: 'conv 3x3 s2' in l),
 ('first 7×1 (unfolded) / final 1×7 (row-folded)', lambda l: 'conv 7x1' in l or 'conv 1x7' in l),
 ('LayerNorm (+ split-K sum)', lambda l: l.strip().startswith('ln ')),
 ('ctx partial / reduce / statistics (unfused levels)', lambda l: l.strip().startswith(('ctxp', 'ctxr', 'kstats'))),
 ('fold GEMVs (`ctxf`)', lambda l: l.strip().startswith('ctxf')),
 ('split-K sums, copies, plane packs, sampler, rest', lambda l: l.strip().startswith(('copy', 'combine', 'unfold', 'ddim', 'temb', 'pfpack'))),
]
tot = sum(float(o[1]) for o in ops)
lines = []; n_all = 0
for name, pred in rows:
    n, ms, tf = agg(pred); n_all += n
    lines.append(f'| {name} | {n} | {ms:.2f} | {100*ms/tot:.1f} % | ' + (f'{tf:.0f} |' if tf > 1 else '– |'))
assert n_all == len(ops), (n_all, len(ops))
tab = '\n'.join(lines)
b1 = last('profiles/bench_r02_batch1.json'); e = last('profiles/bench_r02_eps_1000step.json'); f = last('profiles/bench_r02_512_b16.json')
blk = f'''### Round 2 (`profiles/*_r02*`)

`python bench.py` (defaults: 1 GPU, batch 32, 256², 500 iterations, one timed decode): **{j["value"]:.2f} images/s**
({j["ms_per_step"]/1e3:.2f} s per batch, {r["ms_per_ddim_iter"]:.2f} ms per DDIM iteration; round 1: 3.06).  The timed decode is checked, not
just `isfinite`: rows 0 and 31 are decoded again on their own (batch-1 launch plans) and agree to
{j["verify"]["max_rel_err_vs_batch1_decode"]:.1e} (tolerance 1e-4).  Other configurations of BASELINE.json (`profiles/bench_r02_*.json`): ε-param
1000 steps {e["value"]:.2f} images/s, 512² batch 16 {f["value"]:.2f} images/s, batch 1 {b1["value"]:.3f} images/s =
{b1["roofline"]["ms_per_ddim_iter"]:.2f} ms per iteration (round 1: 4.4).

`roofline` is the **dominant launch shape of one kernel** — `{r["launch_key"]}`, i.e. block2 of the
64-channel ResnetBlocks on `conv_pf_kernel<2,2,1,4,3,3>` with fused LayerNorm and residual operand, {r["launches_per_iteration"]} launches per
iteration: algorithmic {r["flops_per_launch"]/1e9:.1f} GFLOP / {r["avg_launch_ms"]:.3f} ms = {r["achieved"]:.0f} TFLOP/s against 2500/3 = 833 (three fp16
products per fp32 product): `frac` = {r["frac"]:.2f}.  The rocprofv3 summary of the same command
(`profiles/rocprof_r02_kernel_stats.csv`) lists that kernel name over all its shapes (64 ch @256² and @128²); the
per-launch average of the dominant shape is the `[op]` lines of `profiles/per_op_r02.txt`.  Counter traffic of that
launch (`profiles/pmc_r02_conv3x3_traffic.txt`, separate `--pmc` passes, FETCH_SIZE ×2): {r["traffic"]/1e6:.0f} MB against
{r["algorithmic_bytes_per_launch"]/1e6:.0f} MB algorithmic (planes in + fp32 out + fp32 residual) = {r["traffic"]/r["algorithmic_bytes_per_launch"]:.2f}×; `pmc_r02_conv3x3_mfma.txt`:
`SQ_INSTS_MFMA` 14 155 776 = 3 × 154.6e9 / 32768 exactly, MFMA busy 42 % of SIMD cycles, no LDS bank conflicts.
This layer is as close to the HBM roof as to the matrix roof: 1.61 GB / 8 TB/s = 0.20 ms, 154.6 GFLOP / 833 = 0.19 ms,
measured {r["avg_launch_ms"]:.2f}.  Whole path: {r["whole_path_tflops_canonical"]:.0f} TFLOP/s of canonical fp32 work ({r["whole_path_tflops_executed"]:.0f} executed — context hoisting
removes a quarter of the reference's operations).

Per DDIM iteration (`profiles/per_op_r02.txt`, hipEvent averages of the instrumented iterations inside the timed
region; batch 32, {len(ops)} launches, {tot:.1f} ms):

| class | launches | ms / iteration | share | fp32-equivalent TFLOP/s |
|---|---|---|---|---|
{tab}

'''
s = open('DESIGN.md').read()
marker = 'CPU baseline (oracle port, 256 host threads, EPYC 9575F)'
a = s.index('### Round 2 (`profiles/*_r02*`)') if '### Round 2 (`profiles/*_r02*`)' in s else s.index(marker)
b = s.index(marker)
open('DESIGN.md', 'w').write(s[:a] + blk + s[b:])
print("DESIGN.md section 5 round-2 block regenerated")
