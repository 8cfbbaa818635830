#!/bin/bash
# usage: tools/prof.sh <tag> [bench args...]   -- run on the GPU box via gpurun
# rocprofv3 kernel-trace + stats, then separate PMC passes (never combined with trace domains other than kernel-trace)
set -u
EXTRA=${BENCH_EXTRA:-}   # bench arguments that define the configuration (e.g. "--D 8 --batch 768 --rotate 1"), applied to every pass
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
mkdir -p $out
# the trace pass runs the bench with its DEFAULT step / warm-up counts (the settled-clock regime the bench line is quoted
# in); the PMC passes only need a few dispatches ("$@", e.g. --steps 5 --warmup 1): counters do not depend on the clock
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/bench.py --no-cpu-baseline --no-extras $EXTRA > $out/bench_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc_sq -- python3 $R/bench.py --no-cpu-baseline --no-extras $EXTRA "$@" > $out/bench_pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --no-extras $EXTRA "$@" > $out/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $R/bench.py --no-cpu-baseline --no-extras $EXTRA "$@" > $out/bench_pmc_write.log 2>&1
find $out -name "*.csv" | head -30
for f in $(find $out/trace -name "*kernel_stats.csv"); do echo "== $f"; cat $f; done
