#!/bin/bash
# GIST-shaped 1 000-query batches at the reference's large beams (efs 600 / 800 / 1 000), three in flight: the table form (one batch fills
# the LDS: nothing runs beside it) against the HBM-bitmap first pass (GBNNS_FLAG_BITMAP_PASS; lists only in LDS).  GPU box, repo root.
for EF in ${@:-600 800 1000}; do
  for BP in "" "--bitmap-pass"; do
    python3 bench.py --config gist --ef $EF --steps 30 --warmup 5 --no-other-configs --no-cpu-baseline --no-extras $BP 2>/dev/null |
      python3 -c "import sys, json; d = json.loads(sys.stdin.readline()); print('ef %4d %-14s %.4f ms per batch in flight  %.3f M queries/s   serial %.4f ms   %s' % ($EF, '$BP' or 'table', d['ms_per_step'], d['value'] / 1e6, d['serial']['ms_per_step'], d['roofline']['kernel'][:60]))"
  done
done
