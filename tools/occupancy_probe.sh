#!/bin/bash
# Run ON THE GPU BOX from the repo root: how the first-pass walk kernel's time depends on the resident wavefronts per CU.
# GBNNS_LDS_PAD makes every first-pass wavefront ask for that many unused LDS bytes, everything else unchanged.
#   tools/occupancy_probe.sh <config> <ef> <pad bytes> [<pad bytes> ...]
CFG=${1:-sift}; EF=${2:-180}; shift 2
mkdir -p gpurun_out/occ && export GBNNS_CACHE=/tmp/gbnns_cache
for pad in "$@"; do
  GBNNS_LDS_PAD=$pad python3 bench.py --config $CFG --ef $EF --steps 10 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/occ/${CFG}_${EF}_$pad.json 2> gpurun_out/occ/${CFG}_${EF}_$pad.err
  echo "$CFG ef=$EF pad=$pad $(grep -o '"kernel_ms": [0-9.]*' gpurun_out/occ/${CFG}_${EF}_$pad.json | head -1)"
done
