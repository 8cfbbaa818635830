#!/bin/bash
# A/B on one box: the committed library (libqmps_hip_head.so next to this script, built from HEAD~) against the working tree's, evolve D = 2.
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out
cd $R
timeout 1500 python -m pytest tests/test_evolve_gpu.py -x -q -m gpu > $o/cp2_tests.log 2>&1; echo "tests exit $?" >> $o/cp2_tests.log
for rep in 1 2; do
for lib in new head; do
  if [ $lib = head ]; then export QMPS_HIP_LIB=$R/profiles/experiments/r06/libqmps_hip_head.so; else unset QMPS_HIP_LIB; fi
  timeout 600 python bench.py --workload evolve --D 2 --ansatz shallow-full --batch 256 --steps 10 --warmup 3 --no-cpu-baseline > $o/cp2_${lib}_full_t256_$rep.json 2>$o/cp2_err.log
  timeout 600 python bench.py --workload evolve --D 2 --ansatz shallow-full --batch 4096 --steps 10 --warmup 3 --no-cpu-baseline > $o/cp2_${lib}_full_t4096_$rep.json 2>>$o/cp2_err.log
  timeout 600 python bench.py --workload evolve --D 2 --batch 256 --steps 10 --warmup 3 --no-cpu-baseline > $o/cp2_${lib}_cnot_t256_$rep.json 2>>$o/cp2_err.log
done; done
