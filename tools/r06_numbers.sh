#!/bin/bash
# usage (GPU box): bash tools/r06_numbers.sh <tag>   -> gpurun_out/<tag>_*.json + rocprofv3 kernel statistics of the shipped library:
# the bench lines and CSVs DESIGN.md sections 6 and 7 quote for round 6 (the r05 set + the plain power-iteration solver as a headline-type run)
tag=${1:-r06}
R=$GRAFT_REPO_ROOT
o=$R/gpurun_out
run() { n=$1; shift; timeout 900 python $R/bench.py "$@" > $o/${tag}_$n.json 2> $o/${tag}_$n.err || echo "FAILED $n"; }
run headline
# BASELINE configs[2] read literally: the same step with the power-iteration environment solves (plain = env_power_d4_kernel of round 6)
run d4_plain --solver plain --steps 200 --warmup 20 --no-cpu-baseline --no-extras
QMPS_POWER_LANE=1 timeout 900 python $R/bench.py --solver plain --steps 60 --warmup 10 --no-cpu-baseline --no-extras > $o/${tag}_d4_plain_lane.json 2> $o/${tag}_d4_plain_lane.err
run d4_squaring --solver squaring --steps 400 --warmup 50 --no-cpu-baseline --no-extras
# single-GPU times of the 8-GPU shards of configs 2, 3, 4 (what a strong-scaled split can gain)
run d4_b8192 --D 4 --batch 8192 --no-cpu-baseline --no-extras
run d8_b768 --D 8 --batch 768 --steps 400 --warmup 100 --no-cpu-baseline --no-extras
run d8_b96 --D 8 --batch 96 --steps 400 --warmup 100 --no-cpu-baseline --no-extras
run roto_d8 --workload rotosolve --D 8 --batch 768
run roto_d8_b96 --workload rotosolve --D 8 --batch 96 --no-cpu-baseline
run roto_d8_double --workload rotosolve --D 8 --batch 768 --double-frequency --no-cpu-baseline
run roto_d2 --workload rotosolve --D 2 --batch 4096
run evolve_d16_t256 --workload evolve --D 16 --batch 256 --steps 10 --warmup 3
run evolve_d16_t32 --workload evolve --D 16 --batch 32 --steps 10 --warmup 3 --no-cpu-baseline
run evolve_d16_t2048 --workload evolve --D 16 --batch 2048 --steps 8 --warmup 3 --no-cpu-baseline
run evolve_d8_t256 --workload evolve --D 8 --batch 256 --steps 8 --warmup 2 --no-cpu-baseline
run evolve_d2_full_t256 --workload evolve --D 2 --ansatz shallow-full --batch 256 --steps 10 --warmup 3
run evolve_d2_full_t4096 --workload evolve --D 2 --ansatz shallow-full --batch 4096 --steps 10 --warmup 3 --no-cpu-baseline
run evolve_d2_t256 --workload evolve --D 2 --batch 256 --steps 10 --warmup 3 --no-cpu-baseline
run evolve_d4_t256 --workload evolve --D 4 --batch 256 --steps 10 --warmup 3 --no-cpu-baseline
run evolve_d4_t4096 --workload evolve --D 4 --batch 4096 --steps 10 --warmup 3 --no-cpu-baseline
QMPS_EVOLVE_D2_SQUARING=1 timeout 600 python $R/bench.py --workload evolve --D 2 --ansatz shallow-full --batch 256 --steps 10 --warmup 3 --no-cpu-baseline > $o/${tag}_evolve_d2_full_t256_squaring.json 2> $o/${tag}_evolve_d2_full_t256_squaring.err
run overlap_d16 --workload overlap --D 16 --batch 768 --no-cpu-baseline
run overlap_d4 --workload overlap --D 4 --batch 65536 --no-cpu-baseline
# ---- rocprofv3 kernel statistics of the shipped library (one pass each, kernel-trace + stats only)
cd /tmp && export TMPDIR=/tmp
prof() { n=$1; shift; rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_${tag}_$n -- python3 $R/bench.py --no-cpu-baseline --no-extras "$@" > $o/${tag}_prof_$n.log 2>&1
         for f in $(find $o/prof_${tag}_$n -name "*kernel_stats.csv"); do cp $f $o/${tag}_${n}_kernel_stats.csv; done; rm -rf $o/prof_${tag}_$n; }
prof headline
prof d4_plain --solver plain --steps 200 --warmup 20
prof evolve_d16_t256 --workload evolve --D 16 --batch 256 --steps 10 --warmup 3
prof roto_d8 --workload rotosolve --D 8 --batch 768
prof evolve_d2_full_t256 --workload evolve --D 2 --ansatz shallow-full --batch 256 --steps 10 --warmup 3
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob("$o/${tag}_*.json")):
    try:
        d=json.load(open(f)); r=d.get("roofline") or {}; c=d.get("config") or {}
        print(os.path.basename(f), "value=%.4g"%d["value"], d["unit"], "ms/step=%.4g"%d["ms_per_step"], "frac=%s"%r.get("frac"), "upd_us=%s"%c.get("us_per_parameter_update"), "identity=%s"%(d.get("identity_start") or {}).get("ms_per_step"), "busy=%s"%((c.get("device_busy") or {}).get("share")))
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
