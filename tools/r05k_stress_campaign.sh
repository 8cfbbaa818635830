mkdir -p gpurun_out/r05k_stress; cd $GRAFT_REPO_ROOT
S=profiles/experiments/r05
run() { name=$1; shift; timeout 420 python3 $S/$name.py "$@" > gpurun_out/r05k_stress/${name}_$2.json 2> gpurun_out/r05k_stress/${name}_$2.err; echo "$name $* rc=$?"; }
run stress_energy 300 101
run stress_overlap 300 102
run stress_evolve 150 103
run stress_evolve_device 200 104
run stress_gradient 80 105
run stress_rotosolve 80 106
run stress_cell2_optenv 10 107
run stress_brickwall 20 108
run stress_api_state 60 109 4
run stress_api_state 40 110 16
run stress_api_overlap 40 111 8
run stress_api_overlap 30 112 2
run stress_su 10 113
