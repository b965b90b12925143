#!/bin/bash
# Kernel timeline of batches in flight (run on the GPU box from the repo root): rocprofv3 --kernel-trace around a short bench run,
# the last N dispatches by start time with their queue.  Usage: tools/flight_timeline.sh <tag> [N] [bench args]
TAG=${1:-tl}; N=${2:-60}; shift; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/tl_$TAG
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --steps 30 --warmup 5 --no-other-configs --no-cpu-baseline --no-extras "$@" > $OUT/bench.json 2> $OUT/bench.err
cd $R
python3 - <<PY
import csv, glob, re
rows = []
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the timed steps: the last dispatches before the final idle period; take the last N of the densest stretch
rows = [r for r in rows if re.search(r"walk_|mlp_", r["Kernel_Name"])]
# a stretch in which the dispatches come from three queues and more (batches in flight): the last such, its middle
qs = [r.get("Queue_Id", "?") for r in rows]
good = [i for i in range(len(rows) - 12) if len(set(qs[i:i + 12])) >= 3]
if good:
    # last contiguous run of such windows
    end = good[-1]
    start = end
    gs = set(good)
    while start - 1 in gs: start -= 1
    mid = (start + end) // 2
    sel = rows[max(start, mid - $N // 2): max(start, mid - $N // 2) + $N]
else:
    sel = rows[-$N:]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    m = re.search(r"(walk_\w+|mlp_\w+)", r["Kernel_Name"])
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("start %8.1f  end %8.1f  dur %7.1f  q %s  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), m.group(1) if m else r["Kernel_Name"][:40]))
PY
find $OUT -name "*.csv" -size +2M -delete
