#!/bin/bash
# usage (GPU box): bash profiles/experiments/scratch/trace_groups.sh <T> <K_py> <K_lib>: per-stream kernel statistics of the timed call
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $o/trace_g
rocprofv3 --kernel-trace --output-format csv -d $o/trace_g -- python3 $R/profiles/experiments/scratch/evolve_groups.py $1 $2 $3 > $o/trace_g.log 2>&1
tail -2 $o/trace_g.log
f=$(find $o/trace_g -name "*kernel_trace.csv" | head -1)
python3 - <<PY
import csv, collections
rows=list(csv.DictReader(open("$f")))
print(list(rows[0].keys()))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
tend=int(rows[-1]['End_Timestamp'])
# the timed call = the last ~45 % of the trace: take rows after the largest gap in the second half
starts=[int(r['Start_Timestamp']) for r in rows]
t0=starts[0]
cut=t0+(tend-t0)*0.35
sel=[r for r in rows if int(r['Start_Timestamp'])>cut]
by=collections.defaultdict(list)
for r in sel:
    by[(r.get('Queue_Id'), r.get('Stream_Id'))].append(r)
for k,v in sorted(by.items()):
    busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in v)
    span=int(v[-1]['End_Timestamp'])-int(v[0]['Start_Timestamp'])
    names=collections.Counter(r['Kernel_Name'].split('(')[0][-40:] for r in v)
    print(k, 'launches', len(v), 'busy %.2f ms' % (busy/1e6), 'span %.2f ms' % (span/1e6), names.most_common(3))
PY
rm -rf $o/trace_g
