#!/bin/bash
# world-size-1 run of bench.py's N > 1 branch (RCCL communicator of one rank): what the per-step exchange costs on the streams
# (the QMPS_DBG_* knobs need a library built with -DQMPS_DEBUG_KNOBS: make -C qmps_amd/csrc clean all EXTRA=-DQMPS_DEBUG_KNOBS)
cd $GRAFT_REPO_ROOT
run() { QMPS_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --no-cpu-baseline --no-extras --steps 300 --warmup 300 --exchange-every 1 2>&1 | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('$1', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"; }
run default
QMPS_DBG_NOWAIT=1 run nowait
QMPS_DBG_NOAR=1 run noar
QMPS_DBG_NOAR=1 QMPS_DBG_NOFINISH=1 run noar_nofinish
QMPS_DBG_NOEVENT=1 run noevent
QMPS_DBG_NOEVENT=1 QMPS_DBG_NOAR=1 QMPS_DBG_NOFINISH=1 run noevent_noar_nofinish
