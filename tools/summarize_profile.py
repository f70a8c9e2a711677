#!/usr/bin/env python3
"""Condenses the rocprofv3 output of tools/profile_round.sh into the small files kept under profiles/."""
import collections
import csv
import json
import os
import sys

d, tag = sys.argv[1], sys.argv[2]
out = {"tag": tag, "window": int(sys.argv[3]) if len(sys.argv) > 3 else None}
rows = list(csv.DictReader(open(os.path.join(d, "trace", tag + "_kernel_stats.csv"))))
out["kernel_stats"] = [{"name": r["Name"][:90], "calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6, "pct": float(r["Percentage"])}
                       for r in rows[:12]]
pmc = {}
meta = {}
for sub in ("pmc_a", "pmc_b"):
    rr = list(csv.DictReader(open(os.path.join(d, sub, tag + "_counter_collection.csv"))))
    agg = collections.defaultdict(list)
    for r in rr:
        if "k_verify_id<elp::BN254>" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {"vgpr": int(r["VGPR_Count"]), "agpr": int(r["Accum_VGPR_Count"]), "scratch_bytes_per_lane": int(r["Scratch_Size"]),
                    "grid": int(r["Grid_Size"]), "workgroup": int(r["Workgroup_Size"])}
    for k, v in agg.items():
        pmc[k] = sum(v) / len(v)
out["k_verify_id"] = meta
out["pmc_per_launch"] = pmc
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    # MI355X_MICROARCH.md (HBM): counters are KiB; FETCH_SIZE reports half the bytes of wide reads on gfx950 -> doubled (upper bound
    # for the mixed-width scratch accesses of this kernel); WRITE_SIZE is exact.
    out["k_verify_id_bytes_per_launch"] = int((2 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024)
    out["k_verify_id_bytes_per_launch_uncorrected"] = int((pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024)
if "TCC_HIT_sum" in pmc:
    out["l2_hit_rate"] = pmc["TCC_HIT_sum"] / (pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"])
json.dump(out, open(os.path.join(d, "summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
