#!/bin/bash
# the lines the D = 4 tie path and the D = 2 conditioning test could have moved (compare with profiles/r06f_*.json of the library before them)
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out; cd $R
run() { n=$1; shift; timeout 600 python bench.py "$@" --no-cpu-baseline > $o/ties_$n.json 2>$o/ties_err.log; }
for rep in 1 2; do
run overlap_d4_$rep --workload overlap --D 4 --batch 65536
run evolve_d4_t256_$rep --workload evolve --D 4 --batch 256 --steps 10 --warmup 3
run evolve_d4_t4096_$rep --workload evolve --D 4 --batch 4096 --steps 10 --warmup 3
run evolve_d2_full_t256_$rep --workload evolve --D 2 --ansatz shallow-full --batch 256 --steps 10 --warmup 3
run evolve_d2_full_t4096_$rep --workload evolve --D 2 --ansatz shallow-full --batch 4096 --steps 10 --warmup 3
done
