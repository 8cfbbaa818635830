#!/bin/bash
# same-box A/B: the D = 4 device driver without (libqmps_hip_head.so) and with the out-of-line neighbour solves of tied points
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out; cd $R
for rep in 1 2 3; do for lib in head cur; do
  export QMPS_HIP_LIB=$R/profiles/experiments/r06/libqmps_hip_$lib.so
  timeout 600 python bench.py --workload evolve --D 4 --batch 256 --steps 10 --warmup 3 --no-cpu-baseline > $o/tn_${lib}_t256_$rep.json 2>$o/tn_err.log
  timeout 600 python bench.py --workload evolve --D 4 --batch 4096 --steps 10 --warmup 3 --no-cpu-baseline > $o/tn_${lib}_t4096_$rep.json 2>>$o/tn_err.log
done; done
