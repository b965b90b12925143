#!/bin/bash
# A/B harness (GPU box, repo root): benches variants/<name>.so in turn on the given "config ef" pairs.
#   tools/ab4.sh "sift 64,sift 36" base qlds ...
# Prints: in-flight rate, serial rate, first-pass kernel, kernel ms, roofline fraction, projection ms.
export GBNNS_CACHE=/tmp/gbnns_cache
PAIRS=$1; shift
VARIANTS="$@"
cp gbnns_dim_red_amd/lib/libgbnns_hip.so /tmp/orig.so
trap 'cp /tmp/orig.so gbnns_dim_red_amd/lib/libgbnns_hip.so' EXIT   # the shipped library comes back even when a variant crashes
IFS=',' read -ra PP <<< "$PAIRS"
for P in "${PP[@]}"; do
  CFG=${P% *}; EF=${P#* }
  for v in $VARIANTS; do
    cp variants/$v.so gbnns_dim_red_amd/lib/libgbnns_hip.so
    timeout -k 10 300 python3 bench.py --config $CFG --ef $EF --no-cpu-baseline --no-extras --steps ${STEPS:-100} --warmup 10 2>/tmp/ab_err.txt | tail -1 | python3 -c "
import sys,json; j=json.loads(sys.stdin.read()); k=j['kernels_ms']; r=j['roofline']
print('%-10s ef %-4s %-14s in flight %7.3f M  serial %7.3f M  %s %.4f ms frac %.4f  proj %.4f  general %s' % ('$CFG', '$EF', '$v', j['value']/1e6, j['serial']['queries_per_s']/1e6, r['kernel'].split(' ')[0], r['kernel_ms'], r['frac'], k['project'], k.get('general_queries')))" || tail -5 /tmp/ab_err.txt
  done
done
