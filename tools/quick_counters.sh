#!/bin/bash
# Quick SQ counter pass for the walk kernel (run on the GPU box). Usage: tools/quick_counters.sh <tag> [bench args]
TAG=${1:-q}; shift
OUT=gpurun_out/qc_$TAG
mkdir -p $OUT
export GBNNS_CACHE=/tmp/gbnns_cache
ARGS="bench.py --steps 10 --warmup 2 --no-cpu-baseline $@"
python3 $ARGS > $OUT/bench_plain.json 2> $OUT/bench_plain.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq1 -- python3 $ARGS > /dev/null 2> $OUT/pmc_sq1.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- python3 $ARGS > /dev/null 2> $OUT/pmc_sq2.err
python3 tools/digest_profile.py $OUT 2>/dev/null | grep -E "^walk_|^rerank|^mlp|counters"
python3 -c "
import json;j=json.load(open('$OUT/bench_plain.json'));print('QPS',j['value'],j['kernels_ms'])"
find $OUT -name "*.csv" -size +1M -delete
