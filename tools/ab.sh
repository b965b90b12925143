#!/bin/bash
# A/B harness (GPU box): benches every variants/<name>.so in turn. Usage: tools/ab.sh [names...]
export GBNNS_CACHE=/tmp/gbnns_cache
cp gbnns_dim_red_amd/lib/libgbnns_hip.so /tmp/orig.so
trap 'cp /tmp/orig.so gbnns_dim_red_amd/lib/libgbnns_hip.so' EXIT   # the shipped library comes back even when a variant crashes
NAMES=${@:-$(ls variants/*.so | xargs -n1 basename | sed 's/\.so$//')}
for v in $NAMES; do
  cp variants/$v.so gbnns_dim_red_amd/lib/libgbnns_hip.so
  python3 bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 | python3 -c "
import sys,json; j=json.loads(sys.stdin.read()); k=j['kernels_ms']; print('%-28s walk %.4f  qps %.3fM  proj %.4f rerank %.4f' % ('$v', k['walk'], j['value']/1e6, k['project'], (k['rerank'] or 0.0)))"
done
