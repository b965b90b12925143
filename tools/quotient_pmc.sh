#!/bin/bash
# counters of the walk kernel with the visited set in its packed (GBNNS_QUOTIENT=0) and quotient forms: tools/quotient_pmc.sh <config> <ef>
cd /tmp >/dev/null; export TMPDIR=/tmp; cd - >/dev/null
for Q in 0 1; do
  export GBNNS_QUOTIENT=$Q
  O=gpurun_out/qpmc_$1_$2_q$Q
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O -- python3 bench.py --config $1 --ef $2 --steps 5 --warmup 2 --no-cpu-baseline --no-extras --serial > /dev/null 2> $O.err
  python3 - $O <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0].split("::")[-1][:30]
    if "walk_hot" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    print(sys.argv[1][-12:], k, " ".join("%s=%.3g" % (c, sum(x[-5:])/len(x[-5:])) for c,x in sorted(v.items())))
PY
done
