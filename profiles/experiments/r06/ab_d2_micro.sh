#!/bin/bash
# same-box A/B of micro-changes to the D = 2 characteristic-polynomial solve (libqmps_hip_head.so = before, libqmps_hip_cur.so = after)
R=$GRAFT_REPO_ROOT; o=$R/gpurun_out; cd $R
for rep in 1 2 3; do for lib in head cur; do
  export QMPS_HIP_LIB=$R/profiles/experiments/r06/libqmps_hip_$lib.so
  timeout 600 python bench.py --workload evolve --D 2 --ansatz shallow-full --batch 256 --steps 10 --warmup 3 --no-cpu-baseline > $o/mic_${lib}_t256_$rep.json 2>$o/mic_err.log
  timeout 600 python bench.py --workload evolve --D 2 --ansatz shallow-full --batch 4096 --steps 10 --warmup 3 --no-cpu-baseline > $o/mic_${lib}_t4096_$rep.json 2>>$o/mic_err.log
done; done
unset QMPS_HIP_LIB
timeout 900 python -m pytest tests/test_evolve_gpu.py -q -m gpu -x  > $o/mic_tests.log 2>&1
