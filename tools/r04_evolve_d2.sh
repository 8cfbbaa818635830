#!/bin/bash
# usage (GPU box): bash tools/r04_evolve_d2.sh <tag> : the D = 2 time evolution with the device-resident optimiser against the host loop
tag=${1:-r04c}
R=$GRAFT_REPO_ROOT
o=$R/gpurun_out
run() { n=$1; shift; timeout 600 python $R/bench.py "$@" > $o/${tag}_$n.json 2> $o/${tag}_$n.err || echo "FAILED $n"; }
run evolve_d2_t256 --workload evolve --D 2 --batch 256 --steps 8 --warmup 2
run evolve_d2_t256_host --workload evolve --D 2 --batch 256 --steps 8 --warmup 2 --host-driver --no-cpu-baseline
run evolve_d2_full_t256 --workload evolve --D 2 --batch 256 --steps 8 --warmup 2 --ansatz shallow-full
run evolve_d2_full_t256_host --workload evolve --D 2 --batch 256 --steps 8 --warmup 2 --ansatz shallow-full --host-driver --no-cpu-baseline
run evolve_d2_full_t4096 --workload evolve --D 2 --batch 4096 --steps 8 --warmup 2 --ansatz shallow-full --no-cpu-baseline
run evolve_d2_full_t65536 --workload evolve --D 2 --batch 65536 --steps 4 --warmup 1 --ansatz shallow-full --no-cpu-baseline
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob("$o/${tag}_*.json")):
    try:
        d=json.load(open(f)); r=d.get("roofline") or {}; c=d["config"]
        print(os.path.basename(f), "value=%.4g"%d["value"], "ms/step=%.4g"%d["ms_per_step"], "frac=%.3g"%r.get("frac"), "iters/step", c["bfgs_iterations_per_step"], "share", c["kernel_share_of_wall"], "f", c["mean_final_objective"], c["driver"][:40], (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e: print(os.path.basename(f), "ERR", e, open(f.replace(".json",".err")).read()[-300:])
PY
