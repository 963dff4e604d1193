"""Print the tail of a rocprofv3 kernel trace (csv) in launch order: offset from the first listed kernel, duration."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = rows[-last:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  {name}")
    prev_end = max(prev_end, e)
