#!/bin/bash
# Prints the vmcnt waits of the hot walk kernel instance (L2, 8 steps, 32-bit offsets, ef <= 64) with
# context.  A `s_waitcnt vmcnt(0)` right before the row gather's address computation means the
# register allocator reused a load destination for the address: the adjacency prefetch is then
# serialised with the gather (costs ~10 % of the walk time) -- perturb the source until it is gone.
set -e
cd "$(dirname "$0")/../gbnns_dim_red_amd/csrc"
OUT=${1:-/tmp/gbnns_isa.s}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math \
    -fhip-fp32-correctly-rounded-divide-sqrt --cuda-device-only -S -o $OUT.full kernels.hip 2>/dev/null
awk '/^_ZN5gbnns12_GLOBAL__N_115walk_reg_kernelILi0ELi8ELb1ELb0ELi1ELb1EEEvNS_10WalkParamsE:/{f=1} f{print} /^.Lfunc_end/{if(f){exit}}' $OUT.full \
    | grep -v "^\s*;" | grep -v "^\s*\.\(p2align\|loc\|cfi\)" > $OUT
echo "instructions: $(grep -c "^\s[a-z]" $OUT)   vgprs: $(grep "walk_reg_kernelILi0ELi8ELb1ELb0ELi1ELb1EEEvNS_10WalkParamsE.num_vgpr" $OUT.full | awk '{print $NF}')"
if grep -B1 -A1 "vmcnt(0)" $OUT | grep -A1 "vmcnt(0)" | grep -q "v_mul_lo_u32.*s3"; then echo "ARTIFACT: vmcnt(0) before the gather address"; else echo "gather address: no forced wait"; fi
grep -n -A1 "vmcnt" $OUT | grep -v "^--" | paste - - | head -20
