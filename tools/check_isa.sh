#!/bin/bash
# The arithmetic / scheduling contract of the shipped code object, checked on its disassembly:
# no fused multiply-add in walk_* / rerank_* / knn_scan / gd_prune, and no `s_waitcnt vmcnt(0)` between the
# adjacency prefetch and the row gather of the walk_hot kernels.  Same checks as the CPU test suite runs
# (tests/test_isa_contract.py); pass a kernel-name pattern to also dump that kernel's instructions.
#   tools/check_isa.sh [pattern, e.g. walk_hot_kernel] [out.s]
set -e
cd "$(dirname "$0")/.."
python -m pytest tests/test_isa_contract.py -q
if [ -n "$1" ]; then
  OUT=${2:-/tmp/gbnns_isa.s}
  W=$(mktemp -d)
  cp gbnns_dim_red_amd/lib/libgbnns_hip.so $W/
  /opt/rocm/lib/llvm/bin/llvm-objdump --offloading $W/libgbnns_hip.so > /dev/null 2>&1
  : > $OUT
  for O in $W/*gfx950; do   # one code object per compilation unit; a kernel present in several is dumped once
    /opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn $O | awk -v pat="$1" '/^[0-9a-f]+ <.*>:$/{f = ($0 ~ pat)} f{print}' | sed 's,[[:space:]]*//.*,,' > $OUT.part
    if [ -s $OUT.part ] && [ ! -s $OUT ]; then mv $OUT.part $OUT; fi
  done
  rm -f $OUT.part
  echo "$(grep -c '^\s' $OUT) instructions of kernels matching '$1' written to $OUT"
  rm -rf $W
fi
