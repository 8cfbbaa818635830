#!/bin/bash
# usage (GPU box): bash tools/r04_krylov_check.sh <tag> : the Krylov fall-back's tests + the bench lines it must not slow down
tag=${1:-r04a}
R=$GRAFT_REPO_ROOT
o=$R/gpurun_out
python -m pytest $R/tests/test_overlap_gpu.py $R/tests/test_evolve_gpu.py -m gpu -x -q 2>&1 | tail -8
run() { n=$1; shift; timeout 600 python $R/bench.py "$@" > $o/${tag}_$n.json 2> $o/${tag}_$n.err || echo "FAILED $n"; }
run overlap_d16_b768 --workload overlap --D 16 --batch 768 --steps 20 --warmup 3 --no-cpu-baseline
run overlap_d8_b768 --workload overlap --D 8 --batch 768 --steps 20 --warmup 3 --no-cpu-baseline
run evolve_d16_t256 --workload evolve --D 16 --batch 256 --steps 10 --warmup 3 --no-cpu-baseline
run evolve_d16_t2048 --workload evolve --D 16 --batch 2048 --steps 8 --warmup 3 --no-cpu-baseline
run evolve_d8_t256 --workload evolve --D 8 --batch 256 --steps 8 --warmup 2 --no-cpu-baseline
run evolve_d8_t2048 --workload evolve --D 8 --batch 2048 --steps 6 --warmup 2 --no-cpu-baseline
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob("$o/${tag}_*.json")):
    try:
        d=json.load(open(f)); r=d.get("roofline") or {}
        print(os.path.basename(f), "value=%.4g"%d["value"], d["unit"], "ms/step=%.4g"%d["ms_per_step"], "frac=%s"%r.get("frac"), (d.get("identity_start") or {}).get("value"), d.get("config",{}).get("solver_stats"))
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
