#!/bin/bash
# GIST-shaped batches in flight by projection kernel (GPU box, repo root): the default (slab kernel for the narrow layer, big-tile
# kernel for the hidden ones), the small-footprint kernel for the hidden layers (GBNNS_MLP_SMALL below the batch size), at depth 3 / 4.
# Usage: tools/gist_flight_ab.sh [steps]
STEPS=${1:-90}
run() {  # label, env assignments..., -- bench args
    local label=$1; shift
    local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
    env "${envs[@]}" python3 bench.py --config gist --steps $STEPS --warmup 10 --no-other-configs --no-cpu-baseline --no-extras "$@" 2>/dev/null |
        python3 -c "import sys, json; d = json.loads(sys.stdin.readline()); print('%-44s %.4f ms per batch  %.3f M queries/s  project %.4f ms  walk %.4f ms (serialised)' % ('$label', d['ms_per_step'], d['value'] / 1e6, d['kernels_ms']['project'], d['kernels_ms']['walk']))"
}
run "default, depth 3" X=1 --
run "hidden layers small-footprint, depth 3" GBNNS_MLP_SMALL=512 --
run "hidden layers small-footprint, depth 4" GBNNS_MLP_SMALL=512 -- --depth 4
run "hidden layers small-footprint, depth 4, 8 queues" GBNNS_MLP_SMALL=512 GPU_MAX_HW_QUEUES=8 -- --depth 4
run "default, depth 3 (again)" X=1 --
