#!/bin/bash
# usage (GPU box): bash profiles/experiments/r06/evolve_d2_solver_ab.sh - the D = 2 device-resident BFGS driver with the characteristic-polynomial solve (default, round 6)
# against the squaring solve (QMPS_EVOLVE_D2_SQUARING=1), same box, interleaved, three runs each
cd $GRAFT_REPO_ROOT
one() { python bench.py --workload evolve $1 --batch ${2:-256} --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); c=d['config']
print('%-9s %-28s value %.4g  ms/step %.4f  median %.4g  it %.1f  f %.12f' % ('$3', '$1 T=${2:-256}', d['value'], d['ms_per_step'], (d.get('repeats') or {}).get('value_median') or 0, c.get('bfgs_iterations_per_step') or 0, c.get('mean_final_objective') or 0))"; }
for rep in 1 2 3; do
  for which in charpoly squaring; do
    if [ $which = squaring ]; then export QMPS_EVOLVE_D2_SQUARING=1; else unset QMPS_EVOLVE_D2_SQUARING; fi
    one "--D 2 --ansatz shallow-full" 256 $which
    one "--D 2" 256 $which
    [ $rep = 1 ] && one "--D 2 --ansatz shallow-full" 4096 $which
    [ $rep = 1 ] && one "--D 2" 4096 $which
  done
done
