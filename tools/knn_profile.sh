#!/bin/bash
# Run ON THE GPU BOX from the repo root: kernel times and matrix-core counters of gbnns_exact_knn (tools/knn_bench.py <n> <k>).
N=${1:-1000000}; K=${2:-48}
OUT=gpurun_out/prof_knn_${N}_k${K}
mkdir -p $OUT
export GBNNS_CACHE=/tmp/gbnns_cache
cd /tmp >/dev/null; export TMPDIR=/tmp; cd - >/dev/null
python3 tools/knn_bench.py $N $K > $OUT/bench_plain.txt 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/knn_bench.py $N $K > $OUT/bench_trace.txt 2> $OUT/trace.err
timeout -k 5 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_BF16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -- python3 tools/knn_bench.py $N $K > /dev/null 2> $OUT/pmc_mfma.err
python3 - <<PY
import csv, glob, collections
rows = []
for f in glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    import re
    m = re.search(r"(knn_\w+|[A-Za-z_]\w*)(?=<|\()", r["Kernel_Name"].replace("(anonymous namespace)::", ""))
    n = re.search(r"knn_\w+", r["Kernel_Name"]).group(0) if "knn_" in r["Kernel_Name"] else (m.group(1) if m else r["Kernel_Name"][:40])
    agg[n][0] += 1; agg[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("== kernel trace (whole knn_bench.py run: warm-up + filter path + exact scan + torch comparison) ==")
for n, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print("%-34s calls %5d  total %10.1f us  avg %9.1f us" % (n, c, us, us / c))
pm = []
for f in glob.glob("$OUT/pmc_mfma/**/*counter_collection.csv", recursive=True):
    pm += list(csv.DictReader(open(f)))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in pm:
    if "knn_filter" not in r["Kernel_Name"]: continue
    n = "knn_filter_kernel"
    acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
for n, d in acc.items():
    disp = len({(r["Dispatch_Id"]) for r in pm if "knn_filter" in r["Kernel_Name"]})
    print("== counters, %s, summed over %d dispatches ==" % (n, disp))
    for k, v in sorted(d.items()): print("  %-28s %.4g" % (k, v))
    cyc = d.get("GRBM_GUI_ACTIVE", 0) / 8
    if cyc: print("  matrix-pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles) = %.3f" % (d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * cyc)))
PY
