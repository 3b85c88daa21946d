#!/bin/bash
# Experimental build of the kernels with extra -D flags -> point_diffusion_refinement_amd/libpdr_lab.so
# usage: bash tools/lab/build_lab.sh -DPDR_LAB_NO_STORE
set -e
cd "$(dirname "$0")/../../point_diffusion_refinement_amd/csrc"
mkdir -p /tmp/pdr_lab
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -c $f -o /tmp/pdr_lab/${f%.hip}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpdr_lab.so /tmp/pdr_lab/*.o
