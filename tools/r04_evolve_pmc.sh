#!/bin/bash
# usage (GPU box): bash tools/r04_evolve_pmc.sh [T]  -> gpurun_out/prof_r04h_evolve_pmc/: matrix-pipe counters of the evolve workload's kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
o=$R/gpurun_out/prof_r04h_evolve_pmc
rm -rf $o; mkdir -p $o
rocprofv3 -L > $o/counters.txt 2>&1
grep -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" $o/counters.txt | sort -u > $o/mfma_counters.txt
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $o/pmc -- python3 $R/bench.py --workload evolve --D 16 --batch ${1:-256} --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $o/bench.log 2>&1
python3 - <<PY
import csv, glob, collections, json
f = glob.glob("$o/pmc/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    agg[r['Kernel_Name'].split('(')[0].replace('void ', '')][r['Counter_Name']].append(float(r['Counter_Value']))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} | {'dispatches': len(next(iter(d.values())))} for k, d in agg.items() if 'qmps' in k and 'probe_fp64' not in k}
json.dump(out, open("$o/summary.json", "w"), indent=1)
for k, d in out.items(): print(k, d)
PY
