"""experiments only: per-call durations and gaps of the last launches in a rocprofv3 kernel trace (argv[1] = directory)"""
import csv, os, sys
d = sys.argv[1]
f = [os.path.join(r, x) for r, _, fs in os.walk(d) for x in fs if x.endswith("kernel_trace.csv")][0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if any(k in r["Kernel_Name"] for k in sys.argv[2].split(","))]
prev = None
for r in sel[-int(sys.argv[3]):]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(r["Kernel_Name"][:34].ljust(34), "dur %9.1f us" % ((e - s) / 1e3), "gap %6.1f us" % (((s - prev) / 1e3) if prev else 0))
    prev = e
