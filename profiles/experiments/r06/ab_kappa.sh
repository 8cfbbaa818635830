#!/bin/bash
# same-box A/B: the D = 2 driver before (libqmps_hip_prekappa.so, commit 4e61e10) and after the conditioning-based hand-back
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out; cd $R
for rep in 1 2 3; do for lib in prekappa cur; do
  if [ $lib = cur ]; then unset QMPS_HIP_LIB; else export QMPS_HIP_LIB=$R/profiles/experiments/r06/libqmps_hip_$lib.so; fi
  timeout 600 python bench.py --workload evolve --D 2 --ansatz shallow-full --batch 256 --steps 10 --warmup 3 --no-cpu-baseline > $o/kap_${lib}_t256_$rep.json 2>$o/kap_err.log
  timeout 600 python bench.py --workload evolve --D 2 --ansatz shallow-full --batch 4096 --steps 10 --warmup 3 --no-cpu-baseline > $o/kap_${lib}_t4096_$rep.json 2>>$o/kap_err.log
done; done
