#!/bin/bash
cd $GRAFT_REPO_ROOT/tools/debug
for opt in "-O3" "-O3 -ffp-contract=off" "-O3 -ffp-contract=on" "-O1" "-O0" "-O3 -fno-finite-math-only -fhonor-nans" "-O3 -DQMPS_SHOW_DEFAULT"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $opt -std=c++17 -I ../../qmps_amd/csrc -o /tmp/rv rule_variants.hip 2>/dev/null && echo "[$opt]" && /tmp/rv fits.bin 486
done
