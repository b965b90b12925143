#!/bin/bash
# First-pass kernel time by explicit visited-set capacity (GPU box, repo root): tools/cap_profile.sh <ef> <cap> [<cap> ...]
EF=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for CAP in "$@"; do
  OUT=$R/gpurun_out/cap_${EF}_$CAP
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/cap_probe.py --config ${CONFIG:-sift} --ef $EF --caps $CAP --reps ${REPS:-10} > $OUT/probe.txt 2>&1
  grep "^cap" $OUT/probe.txt
  python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "walk_" in r["Name"]:
            print("   %-60s calls %5s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  find $OUT -name "*.csv" -size +1M -delete
done
