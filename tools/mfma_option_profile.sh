#!/bin/bash
# Counters of the throughput option's projection kernel (round 6: mlp_mfma_net_kernel, one launch; GBNNS_MFMA_LAYERS=1: round 5's
# mlp_layer_mfma_kernel x 3) and of the exact one-launch projection beside it (run on the GPU box from the repo root; tools/mfma_probe.py
# is the workload).  Writes gpurun_out/r06_mfma_option_summary.txt (copy it to profiles/).
OUT=$GRAFT_REPO_ROOT/gpurun_out/mfma_opt
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
ARGS="$R/tools/mfma_probe.py"
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/pmc -- python3 $ARGS > /dev/null 2> $OUT/pmc.err
cd $R
python3 - <<PY > gpurun_out/r06_mfma_option_summary.txt
import csv, glob, re, collections, json
print("# projection kernels of \`python tools/mfma_probe.py\` (SIFT1M-shaped, 10 000-query batches: 25 calls with the exact projection, 25 with the option)")
print("# rocprofv3 --kernel-trace (durations) and a separate --pmc pass (counters, mean per dispatch)")
dur = collections.defaultdict(list)
for f in glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(mlp_\w+?_kernel|normalize_kernel)", r["Kernel_Name"])
        if m: dur[m.group(1)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(dur.items()):
    v.sort()
    print("%-26s calls %5d  avg %8.2f us  median %8.2f us  min %8.2f us" % (k, len(v), sum(v) / len(v), v[len(v) // 2], v[0]))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for f in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(mlp_\w+?_kernel|normalize_kernel)", r["Kernel_Name"])
        if not m: continue
        acc[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[m.group(1)].add(r["Dispatch_Id"])
for k, v in sorted(acc.items()):
    n = len(cnt[k])
    print(k, " ".join("%s=%.4g" % (c, x / n) for c, x in sorted(v.items())))
    if v.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        cyc = v["GRBM_GUI_ACTIVE"] / n / 8
        print("%s matrix-pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles %.0f) = %.3f" % (k, cyc, v["SQ_VALU_MFMA_BUSY_CYCLES"] / n / (1024 * cyc)))
    if k == "mlp_net_kernel" and v.get("SQ_INSTS_VALU"):
        cyc = v["GRBM_GUI_ACTIVE"] / n / 8
        print("%s VALU issue = SQ_INSTS_VALU x 4 / (1024 SIMDs x kernel cycles %.0f) = %.3f" % (k, cyc, v["SQ_INSTS_VALU"] / n * 4 / (1024 * cyc)))
try:
    j = json.loads([l for l in open("$OUT/bench.json") if l.startswith("{")][-1])
    print("probe line of the traced run: %s" % json.dumps(j))
except Exception as e:
    print("bench line unreadable:", e)
PY
find $OUT -name "*.csv" -size +2M -delete
