#!/bin/bash
# Round-4: prologue / epilogue share of every conv_pf_kernel layer (ablation variant; dbg 256 = no epilogue, 367 = prologue only)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04o; mkdir -p $OUT; cd $R
export CDC_DEV=1 CDC_NO_RANGE_GUARD=1 CDC_HIP_LIB=$R/cdc_compression_amd/libcdc_hip_pfabl.so
for dbg in 0 256 367; do
    CDC_PF_DBG=$dbg CDC_BENCH_OPS=400 python bench.py --sample-steps 30 --prof-every 5 --no-cpu-baseline --no-alt-arith --no-extras --no-other-configs --no-verify > $OUT/bench_$dbg.json 2> $OUT/bench_$dbg.err
    grep -E '^\[op\].* PF( LN)?( |$)' $OUT/bench_$dbg.err | grep -v "PF3\|PW" | sed 's/  */ /g' | cut -d' ' -f2,6-20 > $OUT/pf_$dbg.txt
done
paste -d'|' <(cut -d' ' -f1 $OUT/pf_256.txt) /dev/null | head -0
python3 - <<PY
import re
def load(d):
    rows={}
    for l in open('$OUT/pf_%d.txt'%d):
        t,rest=l.split(' ',1); rows.setdefault(rest.strip(),[]).append(float(t))
    return rows
a,b,c=load(0),load(256),load(367)
tot=[0,0,0]
for k in a:
    x=sum(a[k]); y=sum(b.get(k,[0])); z=sum(c.get(k,[0])); tot[0]+=x; tot[1]+=y; tot[2]+=z
    print('%-75s n=%d full %.3f noepi %.3f proonly %.3f'%(k[:75],len(a[k]),x,y,z))
print('total',tot)
PY
