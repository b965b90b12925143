#!/bin/bash
# First-pass walk time by explicit visited-set capacity (entries) -- what fill of the table (longest walk / capacity) the two-list kernels
# like: SIFT-shaped, graphs GD(M = 16) (one-pass adjacency rows) and GD(M = 30) (two-pass), ef 140 / 160 / 180.  GPU box, repo root.
# Usage: tools/fill_scan.sh "<M list>" "<ef list>" "<fill list in %>"
MS=${1:-"16 30"}; EFS=${2:-"140 160 180"}; FILLS=${3:-"95 88 80 72 65 58"}
for M in $MS; do for EF in $EFS; do
  # the library's own choice first: its capacity and the longest walk it saw
  GBNNS_DEBUG_SIZING=1 python3 bench.py --config sift --graph-M $M --ef $EF --steps 12 --warmup 4 --no-other-configs --no-cpu-baseline --no-extras 2> /tmp/fs.err |
    python3 -c "import sys, json; d = json.loads(sys.stdin.readline()); print('M $M ef $EF auto: in flight %.4f ms  serial %.4f  walk %.4f' % (d['ms_per_step'], d['serial']['ms_per_step'], d['kernels_ms']['walk']))"
  LINE=$(grep "gbnns sizing" /tmp/fs.err | tail -1); echo "      $LINE"
  MAXDC=$(echo "$LINE" | sed 's/.*maxdc \([0-9]*\).*/\1/')
  for F in $FILLS; do
    CAP=$(( MAXDC * 100 / F ))
    GBNNS_DEBUG_SIZING=1 python3 bench.py --config sift --graph-M $M --ef $EF --steps 12 --warmup 4 --no-other-configs --no-cpu-baseline --no-extras --hash-capacity $CAP 2>/dev/null |
      python3 -c "import sys, json; d = json.loads(sys.stdin.readline()); print('      fill %2d %% cap %5d: in flight %.4f ms  serial %.4f  walk %.4f  handed over %d' % ($F, $CAP, d['ms_per_step'], d['serial']['ms_per_step'], d['kernels_ms']['walk'], d['kernels_ms']['general_queries']))"
  done
done; done
