#!/bin/bash
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out; cd $R
for rep in 1 2 3; do for lib in head cur var; do
  export QMPS_HIP_LIB=$R/profiles/experiments/r06/libqmps_hip_$lib.so
  timeout 600 python bench.py --workload overlap --D 4 --batch 65536 --no-cpu-baseline > $o/ties2_${lib}_overlap_d4_$rep.json 2>$o/ties_err.log
  timeout 600 python bench.py --workload evolve --D 4 --batch 4096 --steps 10 --warmup 3 --no-cpu-baseline > $o/ties2_${lib}_evolve_d4_t4096_$rep.json 2>$o/ties_err.log
done; done
