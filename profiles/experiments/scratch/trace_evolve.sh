#!/bin/bash
# usage (GPU box): bash profiles/experiments/scratch/trace_evolve.sh <T> : kernel trace of one evolve run, last time step's kernels in launch order
T=${1:-4096}
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $o/trace_$T -- python3 $R/bench.py --no-cpu-baseline --no-extras --workload evolve --D 16 --batch $T --steps 4 --warmup 2 > $o/trace_$T.log 2>&1
f=$(find $o/trace_$T -name "*kernel_trace.csv" | head -1)
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
out=[]
for r in rows:
    n=r['Kernel_Name']; n=n.split('(')[0].replace('void qmps::','').replace('qmps::','')
    out.append((int(r['Start_Timestamp'])-t0, int(r['End_Timestamp'])-int(r['Start_Timestamp']), n, r.get('Grid_Size_X') or r.get('Grid_Size')))
# last 400 launches
prev=None
for s,d,n,g in out[-260:]:
    gap = (s-prev) if prev is not None else 0
    print("%12.1f us  dur %8.1f  gap %7.1f  %-44s grid %s" % (s/1e3, d/1e3, gap/1e3, n[:44], g))
    prev=s+d
PY
rm -rf $o/trace_$T
