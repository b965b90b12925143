#!/bin/bash
# On the GPU box: rebuild the library with each GBNNS_EXP variant and time the walk kernel.
export GBNNS_CACHE=/tmp/gbnns_cache
for v in 0 1 2 3; do
  make -s -C gbnns_dim_red_amd/csrc clean >/dev/null; make -s -C gbnns_dim_red_amd/csrc EXP=$v >/dev/null 2>&1
  python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('EXP $v walk_ms', j['kernels_ms']['walk'], 'qps', j['value'])
"
done
