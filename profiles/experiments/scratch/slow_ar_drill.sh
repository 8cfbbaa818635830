#!/bin/bash
# exchange pipeline under a SLOW all-reduce at world size 1 (needs a -DQMPS_DEBUG_KNOBS build in qmps_amd/lib/libqmps_hip_dbg.so):
# correctness of every step's cost while the ring is full, and the step time the pipeline sustains
cd $GRAFT_REPO_ROOT
export QMPS_HIP_LIB=$PWD/qmps_amd/lib/libqmps_hip_dbg.so
for it in 0 400 1300 2600; do
  echo "== QMPS_DBG_SLOW_AR=$it"
  QMPS_DBG_SLOW_AR=$it timeout 300 python -m pytest tests/test_direct_gpu.py -q -x -k "accumulation" 2>&1 | grep -E "passed|failed|^E " | head -5
  QMPS_DBG_SLOW_AR=$it timeout 300 python -m pytest tests/test_dist_gpu.py -q -x -k "grouped or world1" 2>&1 | grep -E "passed|failed|^E " | head -5
  QMPS_DBG_SLOW_AR=$it QMPS_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --no-cpu-baseline --no-extras --steps 300 --warmup 300 2>&1 | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('bench', d['value'], d['ms_per_step'], d['summed_cost'])"
done
