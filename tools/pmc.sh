#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters>" [bench args...]
tag=$1; shift; ctr=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/pmc_$tag
mkdir -p $out
rocprofv3 --pmc $ctr --output-format csv -d $out -- python3 $R/bench.py --no-cpu-baseline --no-extras "$@" > $out/bench.log 2>&1
python3 - <<PY
import csv,glob,collections
for f in glob.glob('$out/**/*counter_collection.csv', recursive=True):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:44]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,d in agg.items():
        if 'qmps' in k and 'sum_' not in k: print(k, {c: round(sum(v)/len(v)) for c,v in d.items()})
PY
