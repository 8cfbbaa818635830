tag=r03h
R=$GRAFT_REPO_ROOT
o=$R/gpurun_out
run() { n=$1; shift; timeout 600 python $R/bench.py "$@" > $o/${tag}_$n.json 2> $o/${tag}_$n.err || echo "FAILED $n"; }
run evolve_d16_t256 --workload evolve --D 16 --batch 256 --steps 10 --warmup 3
run evolve_d16_t1024 --workload evolve --D 16 --batch 1024 --steps 10 --warmup 3 --no-cpu-baseline
run evolve_d16_t2048 --workload evolve --D 16 --batch 2048 --steps 8 --warmup 3 --no-cpu-baseline
run evolve_d16_t256_numpy --workload evolve --D 16 --batch 256 --steps 10 --warmup 3 --python-driver --no-cpu-baseline
run evolve_d16_t4096 --workload evolve --D 16 --batch 4096 --steps 6 --warmup 2 --no-cpu-baseline
run evolve_d8_t256 --workload evolve --D 8 --batch 256 --steps 8 --warmup 2 --no-cpu-baseline
run evolve_d8_t1024 --workload evolve --D 8 --batch 1024 --steps 6 --warmup 2 --no-cpu-baseline
run evolve_d4_t256 --workload evolve --D 4 --batch 256 --steps 8 --warmup 2 --no-cpu-baseline
run evolve_d2_t256 --workload evolve --D 2 --batch 256 --steps 8 --warmup 2 --no-cpu-baseline
run overlap_d8_b768 --workload overlap --D 8 --batch 768 --steps 20 --warmup 3 --no-cpu-baseline
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_${tag}_evolve2 -- python3 $R/bench.py --workload evolve --D 16 --batch 256 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $o/${tag}_evolve_trace.log 2>&1
for f in $(find $o/prof_${tag}_evolve2 -name "*kernel_stats.csv"); do cp $f $o/${tag}_evolve_kernel_stats.csv; done
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob("$o/${tag}_evolve*.json")+glob.glob("$o/${tag}_overlap_d8*.json")):
    try:
        d=json.load(open(f)); r=d.get("roofline") or {}
        print(os.path.basename(f), "value=%.4g"%d["value"], d["unit"], "ms/step=%.4g"%d["ms_per_step"], "frac=%s"%r.get("frac"), (d.get("identity_start") or {}).get("value"))
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
