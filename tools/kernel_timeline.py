"""Prints the dispatch timeline of a rocprofv3 --kernel-trace CSV: start offset, duration, queue, kernel (tail of the file or a window).  python tools/kernel_timeline.py <kernel_trace.csv> [first] [count]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else max(0, len(rows) - 120)
count = int(sys.argv[3]) if len(sys.argv) > 3 else 120
t0 = int(rows[first]["Start_Timestamp"])
for r in rows[first:first + count]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%10.3f ms  +%9.3f ms  q=%s  scratch=%s  %s" % ((s - t0) / 1e6, (e - s) / 1e6, r.get("Queue_Id"), r.get("Scratch_Size"), r["Kernel_Name"].split("(")[0][:60]))
