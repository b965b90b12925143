#!/bin/bash
# Usage (on the GPU box): tools/counter_passes.sh <tag> "<C1 C2 ..>" "<C3 C4 ..>" ... -- [bench args]
# One rocprofv3 --pmc pass per quoted group (kernel-trace only), then a per-kernel digest.
TAG=$1; shift
CGROUPS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do CGROUPS+=("$1"); shift; done
[ "$1" == "--" ] && shift
OUT=gpurun_out/cp_$TAG
mkdir -p $OUT
export GBNNS_CACHE=/tmp/gbnns_cache
ARGS="bench.py --steps 10 --warmup 2 --no-cpu-baseline $@"
python3 $ARGS > $OUT/bench_plain.json 2> $OUT/bench_plain.err
i=0
for g in "${CGROUPS[@]}"; do
  i=$((i+1))
  echo "pass $i: $g"
  timeout -k 5 150 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $OUT/pmc_g$i -- python3 $ARGS > /dev/null 2> $OUT/pmc_g$i.err || echo "pass $i ($g) failed: $(grep -v "^\s*@" $OUT/pmc_g$i.err | grep -i "error\|exceeds" | head -2)"
done
python3 tools/digest_profile.py $OUT 2>/dev/null | grep -E "^walk_reg|^walk_fast|^rerank|^mlp" 
python3 -c "
import json;j=json.load(open('$OUT/bench_plain.json'));print('QPS',j['value'],j['kernels_ms'])"
find $OUT -name "*.csv" -size +1M -delete
