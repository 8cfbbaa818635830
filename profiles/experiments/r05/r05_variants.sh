#!/bin/bash
# usage (GPU box): bash profiles/experiments/r05/r05_variants.sh <tag>  - config-4 workload under the neighbour-build / algebra switches
tag=$1; R=$GRAFT_REPO_ROOT; o=$R/gpurun_out
i=0
for v in "" "QMPS_NEIGHBOURS_BESIDE=1" "QMPS_FUSED_PROBE=1" "QMPS_EVOLVE_HOST_ALGEBRA=1" "QMPS_EVOLVE_HOST_ALGEBRA=1 QMPS_NEIGHBOURS_BESIDE=1"; do
  i=$((i+1))
  env $v timeout 300 python $R/bench.py --workload evolve --D 16 --batch 256 --steps 10 --warmup 3 --no-cpu-baseline > $o/${tag}_v$i.json 2> $o/${tag}_v$i.err
  python3 - <<PY
import json
try:
    d=json.load(open("$o/${tag}_v$i.json")); c=d["config"]
    print("[%s]"%"$v", "ms/step %.4f"%d["ms_per_step"], "iters", c.get("bfgs_iterations_per_step"), "share %.3f"%c.get("kernel_share_of_wall"), "identity %.3f"%(d.get("identity_start") or {}).get("ms_per_step"), "median ms %.4f"%(256e3/(d.get("repeats") or {}).get("value_median")))
except Exception as e: print("[%s]"%"$v", "ERR", e)
PY
done
