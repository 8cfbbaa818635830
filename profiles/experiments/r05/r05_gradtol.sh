#!/bin/bash
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out
for gt in 1e-8 1e-7 1e-6 1e-5; do
  QMPS_GRAD_TOL=$gt timeout 300 python $R/bench.py --workload evolve --D 16 --batch 256 --steps 10 --warmup 3 --no-cpu-baseline > $o/gt_$gt.json 2> $o/gt_$gt.err
  python3 - <<PY
import json
try:
    d=json.load(open("$o/gt_$gt.json")); c=d["config"]
    print("grad_tol $gt", "ms/step %.4f"%d["ms_per_step"], "iters", c.get("bfgs_iterations_per_step"), "rounds mean %.1f max %s"%(c.get("solver_rounds_mean_gradient_batches"), c.get("solver_rounds_max_gradient_batches")), "final f %.10f worst %.10f"%(c.get("mean_final_objective"), c.get("worst_final_objective")), "identity %.3f it %s"%((d.get("identity_start") or {}).get("ms_per_step"), (d.get("identity_start") or {}).get("bfgs_iterations_per_step")), "median ms %.4f"%(256e3/(d.get("repeats") or {}).get("value_median")))
except Exception as e: print("grad_tol $gt ERR", e)
PY
done
