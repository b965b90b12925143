#!/bin/bash
# Run ON THE GPU BOX from the repo root (gpurun): A/B of the visited-set forms / the bitmap-pass crossover; output kept in profiles/r03_quotient_ab.txt
for CFG_EF in "glove 400" "glove 500" "glove 600" "sift 450" "sift 500"; do set -- $CFG_EF; for B in 385 2000; do GBNNS_BITMAP_MIN_EF=$B timeout -k 10 300 python bench.py --config $1 --ef $2 --no-cpu-baseline --no-extras 2> /tmp/err.txt | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1 ef $2 bitmap_min_ef $B: in flight %.2f M, serial %.2f M, kernel %s %.4f ms frac %.3f general %s' % (d['value']/1e6, d['serial']['queries_per_s']/1e6, d['roofline']['kernel'].split(' (')[0], d['roofline']['kernel_ms'], d['roofline']['frac'], d['kernels_ms'].get('general_queries')))"; done; done
