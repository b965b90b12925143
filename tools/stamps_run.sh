#!/bin/bash
# GPU box, repo root: tools/stamps.py on variants/stamps.so (a `make STAMPS=1` build), e.g.
#   tools/stamps_run.sh "sift 140 180" "gist 200" "glove 300"
export GBNNS_CACHE=/tmp/gbnns_cache
cp gbnns_dim_red_amd/lib/libgbnns_hip.so /tmp/orig.so
trap 'cp /tmp/orig.so gbnns_dim_red_amd/lib/libgbnns_hip.so' EXIT
cp variants/${STAMPS_LIB:-stamps}.so gbnns_dim_red_amd/lib/libgbnns_hip.so
for spec in "$@"; do
  set -- $spec; cfg=$1; shift
  echo "== $cfg"
  CONFIG=$cfg timeout -k 10 300 python3 tools/stamps.py "$@" 2>/tmp/stamps_err.txt || tail -5 /tmp/stamps_err.txt
done
