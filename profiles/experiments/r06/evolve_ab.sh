#!/bin/bash
# usage (GPU box): bash profiles/experiments/r06/evolve_ab.sh  - same-box A/B of the device-resident BFGS drivers (D = 2, 4; config 4's D = 16 lock-step as a control):
# library A = profiles/experiments/r06/libqmps_hip_head.so (built from the commit before the change), library B = the tree's; three runs each, interleaved
cd $GRAFT_REPO_ROOT
cp qmps_amd/lib/libqmps_hip.so /tmp/lib_new.so
cp profiles/experiments/r06/libqmps_hip_head.so /tmp/lib_old.so
one() { python bench.py --workload evolve $1 --batch ${2:-256} --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); c=d['config']
print('%s  %-28s value %.4g  ms/step %.4f  median %.4g  it %.1f  f %.12f' % ('$3', '$1 T=${2:-256}', d['value'], d['ms_per_step'], (d.get('repeats') or {}).get('value_median') or 0, c.get('bfgs_iterations_per_step') or 0, c.get('mean_final_objective') or 0))"; }
for rep in 1 2 3; do
  for which in old new; do
    cp /tmp/lib_$which.so qmps_amd/lib/libqmps_hip.so
    one "--D 2 --ansatz shallow-full" 256 $which
    one "--D 2" 256 $which
    one "--D 4" 256 $which
    [ $rep = 1 ] && one "--D 2 --ansatz shallow-full" 4096 $which
    [ $rep = 1 ] && one "--D 4" 4096 $which
  done
done
cp /tmp/lib_new.so qmps_amd/lib/libqmps_hip.so
