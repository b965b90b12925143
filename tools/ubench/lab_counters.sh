#!/bin/bash
# SQ counters of the projection kernels in tools/ubench/mlp_lab (run on the GPU box, from the repo root).  Usage: tools/ubench/lab_counters.sh <tag> [lab args]
TAG=${1:-lab}; shift
OUT=gpurun_out/labpmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export MLP_LAB_NO_RATES=1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $R/$OUT/p1 -- $R/tools/ubench/mlp_lab "$@" > /dev/null 2> $R/$OUT/p1.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$OUT/p2 -- $R/tools/ubench/mlp_lab "$@" > /dev/null 2> $R/$OUT/p2.err
rocprofv3 --pmc SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM --kernel-trace --output-format csv -d $R/$OUT/p3 -- $R/tools/ubench/mlp_lab "$@" > /dev/null 2> $R/$OUT/p3.err
cd $R
python3 - <<PY
import csv,glob,collections,re
for d in ("p1","p2","p3"):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv"%d, recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            m=re.search(r"(mlp_\w+?_kernel)(<[^>]*>)?",k)
            if not m: continue
            k=m.group(0)
            acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    for k,v in sorted(acc.items()):
        n=len(cnt[k])
        print(d,k,"dispatches",n)
        for c,x in sorted(v.items()): print("   %-28s %.4g per dispatch"%(c,x/n))
PY
find $OUT -name "*.csv" -size +1M -delete
