#!/bin/bash
# LDS-side counters of the walk kernel (run on the GPU box). Usage: tools/lds_counters.sh <tag>
TAG=${1:-l}; shift
OUT=gpurun_out/lds_$TAG
mkdir -p $OUT
export GBNNS_CACHE=/tmp/gbnns_cache
ARGS="bench.py --steps 10 --warmup 2 --no-cpu-baseline $@"
python3 $ARGS > $OUT/bench_plain.json 2> $OUT/bench_plain.err
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_l1 -- python3 $ARGS > /dev/null 2> $OUT/pmc_l1.err
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_l2 -- python3 $ARGS > /dev/null 2> $OUT/pmc_l2.err
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $OUT/pmc_l3 -- python3 $ARGS > /dev/null 2> $OUT/pmc_l3.err
python3 - <<PY
import csv,glob,collections
for d in ("pmc_l1","pmc_l2","pmc_l3"):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv"%d, recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "walk_reg_kernel" not in k: continue
            key="retry" if "Lb1ELb1" in k or "ELb1ELi1" in k and "Lb0ELi1" not in k else "walk"
            acc[k[:70]][r["Counter_Name"]]+=float(r["Counter_Value"])
            if r["Counter_Name"]=="GRBM_GUI_ACTIVE" or d!="pmc_l1": pass
    for k,v in acc.items():
        print(d,k)
        for c,x in sorted(v.items()): print("   %-28s %.4g"%(c,x))
import json;j=json.load(open("$OUT/bench_plain.json"));print('QPS',j['value'],j['kernels_ms'])
PY
find $OUT -name "*.csv" -size +1M -delete
