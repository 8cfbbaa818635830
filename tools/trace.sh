#!/bin/bash
# usage: tools/trace.sh <tag> [bench args...]  -- kernel-trace + stats only
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/trace_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $R/bench.py --no-cpu-baseline "$@" > $out/bench.log 2>&1
for f in $(find $out -name "*kernel_stats.csv"); do cat $f; done
