#!/bin/bash
# GPU box, repo root: the DEEP10M-shaped 1 M-query launch on variants/<name>.so, optionally "name@waves" = that library
# with the first pass capped at `waves` wavefronts per CU (GBNNS_MAX_WAVES).
#   tools/ab_deep.sh cur3 cur3@28 cur3@24 base
export GBNNS_CACHE=/tmp/gbnns_cache
cp gbnns_dim_red_amd/lib/libgbnns_hip.so /tmp/orig.so
trap 'cp /tmp/orig.so gbnns_dim_red_amd/lib/libgbnns_hip.so' EXIT
for spec in "$@"; do
  v=${spec%@*}; w=0; [[ $spec == *@* ]] && w=${spec#*@}
  cp variants/$v.so gbnns_dim_red_amd/lib/libgbnns_hip.so
  GBNNS_MAX_WAVES=$w timeout -k 10 400 python3 bench.py --config deep --no-cpu-baseline --no-extras --steps ${STEPS:-4} --warmup 1 2>/tmp/abd_err.txt | tail -1 | python3 -c "
import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('deep %-14s in flight %7.3f M  serial %7.3f M  %s %.3f ms frac %.4f' % ('$spec', j['value']/1e6, j['serial']['queries_per_s']/1e6, r['kernel'].split(' ')[0], r['kernel_ms'], r['frac']))" || tail -5 /tmp/abd_err.txt
done
