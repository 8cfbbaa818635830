#!/bin/bash
# usage (GPU box): bash tools/r06_stress_campaign.sh [seed_base=601] [tag=r06_stress]  -> gpurun_out/<tag>/*.json : the randomised stress scripts of round 5
# (profiles/experiments/r05/stress_*.py: device against the oracle on random inputs, a third at symmetric angles) on the ABI 6.4 library with fresh seeds,
# plus the energy stress through the PLAIN power iteration (D = 4: env_power_d4_kernel of round 6) and the squaring solver
B=${1:-601}; T=${2:-r06_stress}
mkdir -p gpurun_out/$T; cd $GRAFT_REPO_ROOT
S=profiles/experiments/r05
run() { name=$1; shift; tagn=${name}_$2${3:+_$3}; timeout 420 python3 $S/$name.py "$@" > gpurun_out/$T/$tagn.json 2> gpurun_out/$T/$tagn.err; echo "$name $* rc=$?"; }
run stress_energy 300 $((B+0))
run stress_energy 300 $((B+15)) plain
run stress_energy 200 $((B+16)) squaring
run stress_overlap 300 $((B+1))
run stress_evolve 150 $((B+2))
run stress_evolve_device 200 $((B+3))
run stress_gradient 80 $((B+4))
run stress_rotosolve 80 $((B+5))
run stress_cell2_optenv 10 $((B+6))
run stress_brickwall 20 $((B+7))
run stress_api_state 60 $((B+8)) 4
run stress_api_state 40 $((B+9)) 16
run stress_api_overlap 40 $((B+10)) 8
run stress_api_overlap 30 $((B+11)) 2
run stress_su 10 $((B+12))
run stress_api_state 60 $((B+13)) 8
run stress_api_state 60 $((B+14)) 2
run stress_api_state 80 $((B+17)) 4
