#!/bin/bash
# usage (GPU box): bash profiles/experiments/r05/r05_evolve_ab.sh <tag> [bench args]   - dev / host algebra A/B of the config-4 workload + kernel trace window
tag=$1; shift
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out
for hs in "" "QMPS_EVOLVE_HOST_ALGEBRA=1"; do
  n=${hs:+host}; n=${n:-dev}
  env $hs timeout 300 python $R/bench.py --workload evolve --D 16 --batch 256 --steps 10 --warmup 3 --no-cpu-baseline "$@" > $o/${tag}_evolve_d16_t256_$n.json 2> $o/${tag}_evolve_$n.err
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$o/${tag}_evolve_d16_t256_*.json")):
    try:
        d=json.load(open(f)); c=d["config"]; print(f.split("/")[-1], "ms/step %.4f"%d["ms_per_step"], "iters", c.get("bfgs_iterations_per_step"), "share %.3f"%c.get("kernel_share_of_wall"), "identity", (d.get("identity_start") or {}).get("ms_per_step"), "median", (d.get("repeats") or {}).get("value_median"))
    except Exception as e: print(f, "ERR", e)
PY
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_$tag -- python3 $R/bench.py --workload evolve --D 16 --batch 256 --steps 10 --warmup 3 --no-cpu-baseline --no-extras "$@" > $o/${tag}_prof.log 2>&1
t=$(find $o/prof_$tag -name "*kernel_trace.csv" | head -1); python3 $R/profiles/experiments/r05/trace_window.py $t 40
f=$(find $o/prof_$tag -name "*kernel_stats.csv" | head -1); cp $f $o/${tag}_evolve_d16_t256_kernel_stats.csv; rm -rf $o/prof_$tag
