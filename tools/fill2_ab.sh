#!/bin/bash
# A/B of the sizing rule for graphs with two-pass adjacency rows (knob vs_fill2: 0 = the one-pass rule, 72 = default): ms per batch, alone and in flight,
# over the reference's beams on GD(M = 30) graphs.  GPU box, repo root.  Usage: tools/fill2_ab.sh "<fill list>"
for F in ${1:-0 72}; do
  echo "GBNNS_VS_FILL2=$F"
  for c in "sift 40,60,80,120,140,160,180" "gist 200,400,600" "deep1m 40,80,120,160,200" "glove1m 300,400"; do
    set -- $c
    GBNNS_VS_FILL2=$F timeout -k 10 250 python3 tools/ref_sweep.py --config $1 --graph-M 30 --only net --efs $2 --sample 32 --reps 4 2>/dev/null | grep "^net" |
      awk -v c=$1 '{printf "  %-8s ef %4d  %-34s serial %8.3f  in flight %8.3f\n", c, $2, substr($3,1,34), $(NF-7), $(NF-6)}'
  done
done
