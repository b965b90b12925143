"""Prints the last N dispatches of a rocprofv3 --kernel-trace CSV as a timeline: start (us, relative), duration, queue."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-int(sys.argv[2]):]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = r["Kernel_Name"].split("(")[0].split("::")[-1][:44]
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-46s start %9.1f  end %9.1f  dur %7.1f  q %s grid %s" % (name, (st - t0) / 1e3, (en - t0) / 1e3, (en - st) / 1e3,
          r.get("Queue_Id", "?"), r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
