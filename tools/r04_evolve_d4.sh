#!/bin/bash
tag=${1:-r04e}
R=$GRAFT_REPO_ROOT
o=$R/gpurun_out
run() { n=$1; shift; timeout 600 python $R/bench.py "$@" > $o/${tag}_$n.json 2> $o/${tag}_$n.err || echo "FAILED $n"; }
run evolve_d4_t256 --workload evolve --D 4 --batch 256 --steps 8 --warmup 2
run evolve_d4_t256_host --workload evolve --D 4 --batch 256 --steps 8 --warmup 2 --host-driver --no-cpu-baseline
run evolve_d4_t4096 --workload evolve --D 4 --batch 4096 --steps 8 --warmup 2 --no-cpu-baseline
run evolve_d4_t4096_host --workload evolve --D 4 --batch 4096 --steps 8 --warmup 2 --host-driver --no-cpu-baseline
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob("$o/${tag}_*.json")):
    try:
        d=json.load(open(f)); r=d.get("roofline") or {}; c=d["config"]
        print(os.path.basename(f), "value=%.4g"%d["value"], "ms/step=%.4g"%d["ms_per_step"], "frac=%.3g"%r.get("frac"), "iters/step", c["bfgs_iterations_per_step"], "share %.2f"%c["kernel_share_of_wall"], "f", c["mean_final_objective"], c["driver"][:40], (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e: print(os.path.basename(f), "ERR", e, open(f.replace(".json",".err")).read()[-300:])
PY
