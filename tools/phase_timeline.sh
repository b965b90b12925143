#!/bin/bash
# Kernel timeline of tools/phase_probe.py (GPU box, from the repo root): rocprofv3 --kernel-trace around the probe with one group
# size; N dispatches from the middle of the run (= the phased stretch: the probe runs ordinary calls, the phased rounds, ordinary
# calls again).  Usage: tools/phase_timeline.sh <tag> [N] [phase_probe args]
TAG=${1:-ph}; N=${2:-80}; shift; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/tl_$TAG
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/tools/phase_probe.py "$@" > $OUT/probe.out 2> $OUT/probe.err
cd $R
cat $OUT/probe.out
python3 - <<PY
import csv, glob, re
rows = []
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if re.search(r"walk_|mlp_", r["Kernel_Name"])]
# the timed loops are the last dense stretch: drop everything before the last gap of more than 50 ms
cut = 0
for i in range(1, len(rows)):
    if int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]) > 50_000_000: cut = i
rows = rows[cut:]
mid = len(rows) // 2
sel = rows[max(0, mid - $N // 2): mid + $N // 2]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    m = re.search(r"(walk_\w+|mlp_\w+)", r["Kernel_Name"])
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("start %8.1f  end %8.1f  dur %7.1f  q %s  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), m.group(1) if m else r["Kernel_Name"][:40]))
PY
find $OUT -name "*.csv" -size +2M -delete
