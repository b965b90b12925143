#!/bin/bash
# GPU box, repo root: gbnns_exact_knn (filter path) of 10^6 x 32 on variants/<name>.so for the given k's.
#   tools/ab_knn.sh "48 100 1000" knn_old knn_new
export GBNNS_CACHE=/tmp/gbnns_cache
KS=$1; shift
cp gbnns_dim_red_amd/lib/libgbnns_hip.so /tmp/orig.so
trap 'cp /tmp/orig.so gbnns_dim_red_amd/lib/libgbnns_hip.so' EXIT
for k in $KS; do
  for v in "$@"; do
    cp variants/$v.so gbnns_dim_red_amd/lib/libgbnns_hip.so
    echo -n "k $k $v: "
    KNN_FILTER_ONLY=1 timeout -k 10 300 python3 tools/knn_bench.py 1000000 $k 2>&1 | tail -1 | cut -c72-110
  done
done
