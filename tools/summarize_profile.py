#!/usr/bin/env python3
"""Condenses the rocprofv3 output of tools/profile_round.sh into the small files kept under profiles/."""
import collections
import csv
import json
import os
import sys

# the dominant kernel: k_verify_id_staged (coalesced record loads, the default since round 3) or k_verify_id (records read in place)
DOMINANT = ("k_verify_id<elp::BN254>", "k_verify_id_staged<elp::BN254>")

d, tag = sys.argv[1], sys.argv[2]
out = {"tag": tag, "window": int(sys.argv[3]) if len(sys.argv) > 3 else None}
rows = list(csv.DictReader(open(os.path.join(d, "trace", tag + "_kernel_stats.csv"))))
out["kernel_stats"] = [{"name": r["Name"][:90], "calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6, "pct": float(r["Percentage"])}
                       for r in rows[:12]]
pmc = {}
meta = {}
dur = {}
for sub in ("pmc_a", "pmc_b", "pmc_c"):
    f = os.path.join(d, sub, tag + "_counter_collection.csv")
    if not os.path.exists(f):
        continue
    rr = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(list)
    for r in rr:
        if any(k in r["Kernel_Name"] for k in DOMINANT):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.setdefault(sub, {})[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            meta = {"vgpr": int(r["VGPR_Count"]), "agpr": int(r["Accum_VGPR_Count"]), "scratch_bytes_per_lane": int(r["Scratch_Size"]),
                    "grid": int(r["Grid_Size"]), "workgroup": int(r["Workgroup_Size"])}
    for k, v in agg.items():
        v = sorted(v)
        pmc[k] = v[len(v) // 2] if len(v) % 2 else (v[len(v) // 2 - 1] + v[len(v) // 2]) / 2      # median over the dispatches: the first dispatch of a pass can carry
                                                                                                    # cycles counted before it started (GRBM_GUI_ACTIVE)
# rocprofv3's VGPR_Count / Accum_VGPR_Count columns are the dispatch packet's granule encoding (236 / 0 for this kernel), not the allocation: the
# registers judged are the code object's (.vgpr_count = arch + accumulation registers of the unified file)
meta = {("rocprof_" + k if k in ("vgpr", "agpr") else k): v for k, v in meta.items()}
try:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import kernel_resources
    obj = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "build", "obj", "elpasso_bn254_stage.o")
    for name, f in kernel_resources.kernel_fields(obj, "k_verify_id_staged").items():
        tot, acc = int(f["vgpr_count"]), int(f.get("agpr_count", 0))
        meta["code_object"] = {"kernel": name, "vgpr_count": tot, "arch_vgpr": tot - acc, "agpr_count": acc, "private_segment_fixed_size": int(f["private_segment_fixed_size"]),
                               "vgpr_spill_count": int(f["vgpr_spill_count"]), "sgpr_spill_count": int(f["sgpr_spill_count"]),
                               "lds_bytes": int(f["group_segment_fixed_size"]), "source": "build/obj/elpasso_bn254_stage.o (tools/kernel_resources.py)"}
except Exception as e:  # the object is absent on the GPU box (build/ does not travel): tools/publish_profile.py fills this in
    meta["code_object"] = {"error": str(e)}
out["k_verify_id"] = meta
out["pmc_per_launch"] = pmc
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    # MI355X_MICROARCH.md (HBM): counters are KiB; FETCH_SIZE reports half the bytes of wide reads on gfx950 -> doubled (upper bound
    # for the mixed-width scratch accesses of this kernel); WRITE_SIZE is exact.
    out["k_verify_id_bytes_per_launch"] = int((2 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024)
    out["k_verify_id_bytes_per_launch_uncorrected"] = int((pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024)
if "TCC_HIT_sum" in pmc:
    out["l2_hit_rate"] = pmc["TCC_HIT_sum"] / (pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"])
if "GRBM_GUI_ACTIVE" in pmc and dur.get("pmc_c"):
    dv = sorted(dur["pmc_c"].values())
    ms_c = dv[len(dv) // 2] if len(dv) % 2 else (dv[len(dv) // 2 - 1] + dv[len(dv) // 2]) / 2
    # GRBM_GUI_ACTIVE is reported once per XCD and the rows of a dispatch are summed above: 8 XCDs on MI355X
    out["xcds"] = 8
    out["clock_ghz"] = pmc["GRBM_GUI_ACTIVE"] / out["xcds"] / ms_c / 1e6
    out["grbm_gui_active_sum_over_xcds"] = pmc["GRBM_GUI_ACTIVE"]
    out["kernel_ms_in_clock_pass"] = ms_c
if "SQ_INSTS_VALU" in pmc and "SQ_WAVES" in pmc:
    out["valu_instructions_per_wave"] = pmc["SQ_INSTS_VALU"] / pmc["SQ_WAVES"]
    out["wait_fraction_of_wave_cycles"] = pmc["SQ_WAIT_ANY"] / pmc["SQ_WAVE_CYCLES"]
# same-lease agreement: the average of ALL k_verify_id dispatches under rocprofv3 (--headline-only: each one is a launch of the headline size) against the
# ms_per_step of the plain bench line taken minutes earlier on the same box
ok = True
try:
    line = json.loads(open(os.path.join(d, "bench_headline.json")).read().strip().splitlines()[-1])
    kv = [r for r in rows if any(k in r["Name"] for k in DOMINANT)]
    if kv:
        avg = float(kv[0]["AverageNs"]) / 1e6
        ok = avg <= 1.02 * line["ms_per_step"]
        out["same_lease_agreement"] = {"rocprof_avg_ms": avg, "rocprof_calls": int(kv[0]["Calls"]), "bench_ms_per_step": line["ms_per_step"],
                                       "bench_kernel_ms": line.get("roofline", {}).get("kernel_ms"), "ratio": avg / line["ms_per_step"], "ok": ok,
                                       "rule": "rocprof avg <= 1.02 x ms_per_step"}
except (OSError, ValueError, KeyError, IndexError) as e:
    out["same_lease_agreement"] = {"error": str(e)}
json.dump(out, open(os.path.join(d, "summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
if not ok:
    sys.exit("rocprof average of k_verify_id exceeds 1.02 x ms_per_step of the same lease's bench line")
