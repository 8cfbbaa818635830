#!/bin/bash
# usage (GPU box): bash tools/r04_numbers.sh <tag>   -> gpurun_out/<tag>_*.json + rocprofv3 kernel statistics of the shipped library:
# the bench lines and CSVs DESIGN.md section 6 quotes for round 4
tag=${1:-r04h}
R=$GRAFT_REPO_ROOT
o=$R/gpurun_out
run() { n=$1; shift; timeout 900 python $R/bench.py "$@" > $o/${tag}_$n.json 2> $o/${tag}_$n.err || echo "FAILED $n"; }
run headline --steps 20 --warmup 5
run d8_b768 --D 8 --batch 768 --steps 400 --warmup 100 --no-cpu-baseline
run d8_b96 --D 8 --batch 96 --steps 400 --warmup 100 --no-cpu-baseline
run d16_b768 --D 16 --batch 768 --steps 100 --warmup 20 --no-cpu-baseline
run d16_b96 --D 16 --batch 96 --steps 100 --warmup 20 --no-cpu-baseline
run d2_b4096 --D 2 --batch 4096 --steps 400 --warmup 100 --no-cpu-baseline
run roto_d2 --workload rotosolve --D 2 --batch 4096
run roto_d4 --workload rotosolve --D 4 --batch 65535 --steps 40 --warmup 4 --no-cpu-baseline
run roto_d8 --workload rotosolve --D 8 --batch 768
run overlap_d16_b768 --workload overlap --D 16 --batch 768 --steps 20 --warmup 3
run overlap_d16_b96 --workload overlap --D 16 --batch 96 --steps 20 --warmup 3 --no-cpu-baseline
run overlap_d8_b768 --workload overlap --D 8 --batch 768 --steps 20 --warmup 3 --no-cpu-baseline
run overlap_d4_b65536 --workload overlap --D 4 --batch 65536 --steps 10 --warmup 2 --no-cpu-baseline
run evolve_d16_t256 --workload evolve --D 16 --batch 256 --steps 10 --warmup 3
run evolve_d16_t1024 --workload evolve --D 16 --batch 1024 --steps 10 --warmup 3 --no-cpu-baseline
run evolve_d16_t2048 --workload evolve --D 16 --batch 2048 --steps 8 --warmup 3 --no-cpu-baseline
run evolve_d16_t4096 --workload evolve --D 16 --batch 4096 --steps 6 --warmup 2 --no-cpu-baseline
run evolve_d8_t256 --workload evolve --D 8 --batch 256 --steps 8 --warmup 2 --no-cpu-baseline
run evolve_d8_t2048 --workload evolve --D 8 --batch 2048 --steps 6 --warmup 2 --no-cpu-baseline
run evolve_d4_t256 --workload evolve --D 4 --batch 256 --steps 8 --warmup 2
run evolve_d4_t256_host --workload evolve --D 4 --batch 256 --steps 8 --warmup 2 --host-driver --no-cpu-baseline
run evolve_d4_t4096 --workload evolve --D 4 --batch 4096 --steps 8 --warmup 2 --no-cpu-baseline
run evolve_d2_t256 --workload evolve --D 2 --batch 256 --steps 8 --warmup 2
run evolve_d2_full_t256 --workload evolve --D 2 --batch 256 --steps 8 --warmup 2 --ansatz shallow-full
run evolve_d2_full_t256_host --workload evolve --D 2 --batch 256 --steps 8 --warmup 2 --ansatz shallow-full --host-driver --no-cpu-baseline
run evolve_d2_full_t4096 --workload evolve --D 2 --batch 4096 --steps 8 --warmup 2 --ansatz shallow-full --no-cpu-baseline
# ---- rocprofv3 kernel statistics of the shipped library (one pass each, kernel-trace + stats only)
cd /tmp && export TMPDIR=/tmp
prof() { n=$1; shift; rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_${tag}_$n -- python3 $R/bench.py --no-cpu-baseline --no-extras "$@" > $o/${tag}_prof_$n.log 2>&1
         for f in $(find $o/prof_${tag}_$n -name "*kernel_stats.csv"); do cp $f $o/${tag}_${n}_kernel_stats.csv; done; rm -rf $o/prof_${tag}_$n; }
prof d16_b768 --D 16 --batch 768 --steps 100 --warmup 20
prof overlap_d16_b768 --workload overlap --D 16 --batch 768 --steps 20 --warmup 3
prof overlap_d8_b768 --workload overlap --D 8 --batch 768 --steps 20 --warmup 3
prof evolve_d16_t256 --workload evolve --D 16 --batch 256 --steps 10 --warmup 3
prof evolve_d16_t2048 --workload evolve --D 16 --batch 2048 --steps 8 --warmup 3
prof roto_d8 --workload rotosolve --D 8 --batch 768
prof roto_d2 --workload rotosolve --D 2 --batch 4096
prof evolve_d2_full_t256 --workload evolve --D 2 --batch 256 --steps 8 --warmup 2 --ansatz shallow-full
prof evolve_d4_t256 --workload evolve --D 4 --batch 256 --steps 8 --warmup 2
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob("$o/${tag}_*.json")):
    try:
        d=json.load(open(f)); r=d.get("roofline") or {}
        print(os.path.basename(f), "value=%.4g"%d["value"], d["unit"], "ms/step=%.4g"%d["ms_per_step"], "frac=%s"%r.get("frac"), "hbm=%s"%r.get("hbm_frac"), (d.get("identity_start") or {}).get("value"))
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
