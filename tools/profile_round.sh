#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: collects the rocprofv3 evidence for one bench.py configuration.
#   tools/profile_round.sh <tag> [config] [ef] [extra bench args]      e.g.  r02a sift 64   |   r02a gist 400
# Outputs under gpurun_out/prof_<tag>_<config>_ef<ef>/; tools/digest_profile.py turns them into a summary (copy it to
# profiles/<tag>_<config>_summary.txt) and into an entry of profiles/counters_latest.json.
# Counter passes are separate runs (gfx950: FETCH_SIZE needs 3 of the 4 TCC slots, WRITE_SIZE 2), never combined with
# trace domains other than kernel-trace, and the program sits directly after `--`.
set -u
TAG=${1:-r00}
CFG=${2:-sift}
EF=${3:-0}
shift 3 2>/dev/null || shift $#
STEPS=10
[ "$CFG" == "deep" ] && STEPS=2
OUT=gpurun_out/prof_${TAG}_${CFG}_ef${EF}
mkdir -p $OUT
export GBNNS_CACHE=/tmp/gbnns_cache
ARGS="bench.py --config $CFG --steps $STEPS --warmup 2 --no-cpu-baseline --no-extras --serial $*"
[ "$EF" != "0" ] && ARGS="$ARGS --ef $EF"
cd /tmp >/dev/null; export TMPDIR=/tmp; cd - >/dev/null
python3 $ARGS > $OUT/bench_plain.json 2> $OUT/bench_plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ARGS > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq1 -- python3 $ARGS > /dev/null 2> $OUT/pmc_sq1.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- python3 $ARGS > /dev/null 2> $OUT/pmc_sq2.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_tcc -- python3 $ARGS > /dev/null 2> $OUT/pmc_tcc.err
python3 tools/digest_profile.py $OUT --last $STEPS --config $CFG --tag $TAG > $OUT/summary.txt 2> $OUT/digest.err
cat $OUT/summary.txt
# keep the merge small: drop the raw per-dispatch CSVs after digesting
find $OUT -name "*.csv" -size +1M -delete
