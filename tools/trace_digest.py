import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
prev_end=None
out=[]
for r in rows:
    name=r["Kernel_Name"]
    short=name.split("(")[0].split("::")[-1][:60]
    st=int(r["Start_Timestamp"]); en=int(r["End_Timestamp"])
    gap=(st-prev_end)/1000 if prev_end else 0
    out.append((short,(en-st)/1000,gap,r.get("Grid_Size_X", r.get("Grid_Size","?"))))
    prev_end=en
for o in out[-int(sys.argv[2]):]:
    print("%-62s dur %9.1f us  gap %8.1f us grid %s"%o)
