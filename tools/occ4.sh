#!/bin/bash
# Run ON THE GPU BOX from the repo root: first-pass kernel time by resident wavefronts per CU (GBNNS_MAX_WAVES caps the
# LDS shares) for variants/<name>.so.   tools/occ4.sh "<config> <ef>" "<waves> <waves> ..." <variant> [<variant> ...]
export GBNNS_CACHE=/tmp/gbnns_cache
P=$1; WAVES=$2; shift 2
CFG=${P% *}; EF=${P#* }
cp gbnns_dim_red_amd/lib/libgbnns_hip.so /tmp/orig.so
trap 'cp /tmp/orig.so gbnns_dim_red_amd/lib/libgbnns_hip.so' EXIT
for v in "$@"; do
  cp variants/$v.so gbnns_dim_red_amd/lib/libgbnns_hip.so
  for w in $WAVES; do
    GBNNS_MAX_WAVES=$w timeout -k 10 300 python3 bench.py --config $CFG --ef $EF --no-cpu-baseline --no-extras --steps ${STEPS:-60} --warmup 10 2>/tmp/occ_err.txt | tail -1 | python3 -c "
import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('%-8s ef %-4s %-8s waves<=%-3s in flight %7.3f M  serial %7.3f M  %s %.4f ms frac %.4f' % ('$CFG', '$EF', '$v', '$w', j['value']/1e6, j['serial']['queries_per_s']/1e6, r['kernel'].split(' ')[0], r['kernel_ms'], r['frac']))" || tail -3 /tmp/occ_err.txt
  done
done
