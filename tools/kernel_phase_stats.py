"""Per-phase kernel statistics of a rocprofv3 --kernel-trace CSV whose phases are separated by a torch elementwise kernel (tools/probes/lone_call_probe.py):
average duration of every kernel per phase and the average gap between consecutive dispatches.  python tools/kernel_phase_stats.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
phase, phases = -1, collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0]
    if "at::native" in name:
        phase += 1
        continue
    if phase >= 0 and ("k_vid" in name or "k_pair" in name or "fillBuffer" in name):
        phases[phase].append(r)
for p, rs in sorted(phases.items()):
    dur = collections.defaultdict(list)
    for r in rs:
        dur[r["Kernel_Name"].split("(")[0].replace("void ", "")[:40] + " grid=" + r.get("Grid_Size", r.get("Grid_Size_X", "?"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(rs, rs[1:])]
    span = (int(rs[-1]["End_Timestamp"]) - int(rs[0]["Start_Timestamp"])) / 1e3
    print("phase %d: %d dispatches, span %.1f us, sum of kernel time %.1f us, mean gap %.1f us" % (p, len(rs), span, sum(sum(v) for v in dur.values()), sum(gaps) / max(1, len(gaps))))
    for k, v in dur.items():
        print("    %-60s n=%3d  avg %9.1f us  min %9.1f  max %9.1f" % (k, len(v), sum(v) / len(v), min(v), max(v)))
