#!/bin/bash
# builds qmps_amd/lib/libqmps_hip_prof.so (-DQMPS_D8_PROFILE: phase clocks in the D = 8 whole-run rotosolve kernel); run d8_profile.py on the GPU box
set -e
cd "$(dirname "$0")/../../qmps_amd/csrc"
mkdir -p build_prof
for f in qmps_kernels qmps_energy_block qmps_energy_d16 qmps_direct qmps_overlap qmps_overlap_grad qmps_roto_d8 qmps_ansatz qmps_su qmps_brickwall qmps_util qmps_capi; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -I../../include -DQMPS_D8_PROFILE -c $f.hip -o build_prof/$f.o ) &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libqmps_hip_prof.so build_prof/*.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
