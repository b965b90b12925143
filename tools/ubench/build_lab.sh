#!/bin/bash
# builds tools/ubench/mlp_lab against a diagnostic build of the two projection units (stamps on); run from the repo root
set -e
HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function -DGBNNS_NET_STAMPS $EXTRA"
mkdir -p /tmp/lab_build
/opt/rocm/bin/hipcc $HIPFLAGS -c gbnns_dim_red_amd/csrc/mlp_net.hip -o /tmp/lab_build/mlp_net.o &
/opt/rocm/bin/hipcc $HIPFLAGS -c gbnns_dim_red_amd/csrc/mlp.hip -o /tmp/lab_build/mlp.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -x hip -Igbnns_dim_red_amd/csrc tools/ubench/mlp_lab.cpp -x none /tmp/lab_build/mlp.o /tmp/lab_build/mlp_net.o -o tools/ubench/mlp_lab
