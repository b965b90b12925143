#!/bin/bash
# Run ON THE GPU BOX from the repo root: vector-memory-path counters (TA / TCP / UTCL1 / TCC->fabric) of one bench.py
# configuration, one rocprofv3 --pmc pass per group (kernel-trace only beside them), serial steps.
#   tools/mem_counters.sh <tag> <config> <ef> [extra bench args]
set -u
TAG=${1:-m}; CFG=${2:-sift}; EF=${3:-64}
shift 3 2>/dev/null || shift $#
OUT=gpurun_out/mem_${TAG}_${CFG}_ef${EF}
mkdir -p $OUT
export GBNNS_CACHE=/tmp/gbnns_cache
ARGS="bench.py --config $CFG --ef $EF --steps 10 --warmup 2 --no-cpu-baseline --no-extras --serial $*"
cd /tmp >/dev/null; export TMPDIR=/tmp; cd - >/dev/null
pass() {  # name counters...   (at most two counters per hardware block and pass: more is refused for TA, and a refused pass hangs)
  local name=$1; shift
  timeout -k 5 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 $ARGS > /dev/null 2> $OUT/pmc_$name.err
  echo "pass $name rc=$?" >> $OUT/passes.txt
}
pass a TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum GRBM_GUI_ACTIVE
pass b TD_TD_BUSY_sum TD_TC_STALL_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCC_REQ_sum TCC_TAG_STALL_sum
pass c TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum
pass d SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LEVEL_WAVES
pass e SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INSTS SQ_ACTIVE_INST_MISC
pass f SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_FLAT
pass g SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQC_DCACHE_BUSY_CYCLES SQ_INST_LEVEL_VMEM
python3 tools/digest_profile.py $OUT --last 10 --config $CFG --tag $TAG > $OUT/summary.txt 2> $OUT/digest.err
grep -E "^== counters|^walk_" $OUT/summary.txt
cat $OUT/passes.txt
find $OUT -name "*.csv" -size +1M -delete
