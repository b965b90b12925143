#!/bin/bash
# Run ON THE GPU BOX from the repo root (gpurun): A/B of the visited-set forms / the bitmap-pass crossover; output kept in profiles/r03_quotient_ab.txt
for CFG_EF in "sift 64" "sift 128" "sift 140" "sift 180" "glove 64" "glove 300" "glove-dot 300"; do set -- $CFG_EF; for Q in 0 1; do GBNNS_QUOTIENT=$Q GBNNS_DEBUG_SIZING=1 timeout -k 10 300 python bench.py --config $1 --ef $2 --no-cpu-baseline --no-extras 2> /tmp/err.txt | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1 ef $2 quotient $Q: in flight %.2f M, serial %.2f M, kernel %s %.4f ms frac %.3f general %s' % (d['value']/1e6, d['serial']['queries_per_s']/1e6, d['roofline']['kernel'].split(' ')[0], d['roofline']['kernel_ms'], d['roofline']['frac'], d['kernels_ms'].get('general_queries')))"; grep "gbnns sizing" /tmp/err.txt | tail -1; done; done
