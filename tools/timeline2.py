"""Kernel dispatches AND memory copies of a rocprofv3 --kernel-trace --memory-copy-trace run, last N events, one timeline."""
import csv, glob, sys
ev = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1][:40], "q" + r.get("Queue_Id", "?")))
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", r.get("Name", "?")), r.get("Stream_Id", "")))
ev.sort()
ev = ev[-int(sys.argv[2]):]
t0 = ev[0][0]
for st, en, name, q in ev:
    print("%-46s start %9.1f  end %9.1f  dur %7.1f  %s" % (name, (st - t0) / 1e3, (en - t0) / 1e3, (en - st) / 1e3, q))
