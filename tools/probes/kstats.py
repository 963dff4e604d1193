"""Print the rows of a rocprofv3 --stats kernel table whose kernel name contains a substring.
   python3 tools/probes/kstats.py <rocprof output dir> [substring]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for row in csv.DictReader(open(f)):
    if pat in row["Name"]:
        print(row["Name"][:70], row["Calls"], "avg_us %.2f" % (float(row["AverageNs"]) / 1e3), "total_ms %.3f" % (float(row["TotalDurationNs"]) / 1e6))
