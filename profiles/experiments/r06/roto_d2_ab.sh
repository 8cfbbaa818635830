#!/bin/bash
# usage (GPU box): bash profiles/experiments/r06/roto_d2_ab.sh - the whole-run D = 2 rotosolve kernel with the restart's cos / sin table (round 6) against
# the library built before it (profiles/experiments/r06/libqmps_hip_head.so), same box, interleaved
cd $GRAFT_REPO_ROOT
cp qmps_amd/lib/libqmps_hip.so /tmp/lib_new.so
cp profiles/experiments/r06/libqmps_hip_head.so /tmp/lib_old.so
one() { python bench.py --workload rotosolve $1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); c=d['config']
print('%-4s %-70s value %.4g  us/update %.3f  last %.6f' % ('$2', '$1', d['value'], c.get('us_per_parameter_update') or 0, c.get('mean_energy_last_sweep') or 0))"; }
for rep in 1 2 3; do
  for which in old new; do
    cp /tmp/lib_$which.so qmps_amd/lib/libqmps_hip.so
    one "--D 2 --batch 4096" $which
    one "--D 2 --batch 4096 --double-frequency --ansatz shallow-full --steps 24 --warmup 2" $which
    one "--D 2 --batch 4096 --double-frequency" $which
  done
done
cp /tmp/lib_new.so qmps_amd/lib/libqmps_hip.so
