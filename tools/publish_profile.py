#!/usr/bin/env python3
"""Copies the evidence collected by tools/profile_round.sh (gpurun_out/prof_<tag>/) into profiles/ under the round's names, keeping
only the rows of the dominant kernel from the per-dispatch tables, and refreshes profiles/hbm_traffic.json (read by bench.py)."""
import csv
import json
import os
import shutil
import sys

# the dominant kernel: k_verify_id_staged (coalesced record loads, the default since round 3) or k_verify_id (records read in place)
DOMINANT = ("k_verify_id<elp::BN254>", "k_verify_id_staged<elp::BN254>")

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src, dst = os.path.join(root, "gpurun_out", "prof_" + tag), os.path.join(root, "profiles")
shutil.copy(f"{src}/bench_n1.json", f"{dst}/{tag}_bench_n1.json")
shutil.copy(f"{src}/bench_under_rocprof.json", f"{dst}/{tag}_bench_under_rocprof.json")
if os.path.exists(f"{src}/bench_headline.json"):
    shutil.copy(f"{src}/bench_headline.json", f"{dst}/{tag}_bench_headline.json")
shutil.copy(f"{src}/trace/{tag}_kernel_stats.csv", f"{dst}/{tag}_kernel_stats.csv")
shutil.copy(f"{src}/summary.json", f"{dst}/{tag}_summary.json")
# registers of the dominant kernel from the CODE OBJECT (build/ does not travel to the GPU box where summarize_profile.py ran)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_resources  # noqa: E402
_s = json.load(open(f"{dst}/{tag}_summary.json"))
_m = _s.get("k_verify_id", {})
if "code_object" not in _m or "error" in _m.get("code_object", {}):
    for name, f in kernel_resources.kernel_fields(os.path.join(root, "build", "obj", "elpasso_bn254_stage.o"), "k_verify_id_staged").items():
        tot, acc = int(f["vgpr_count"]), int(f.get("agpr_count", 0))
        _m["code_object"] = {"kernel": name, "vgpr_count": tot, "arch_vgpr": tot - acc, "agpr_count": acc, "private_segment_fixed_size": int(f["private_segment_fixed_size"]),
                             "vgpr_spill_count": int(f["vgpr_spill_count"]), "sgpr_spill_count": int(f["sgpr_spill_count"]), "lds_bytes": int(f["group_segment_fixed_size"]),
                             "source": "build/obj/elpasso_bn254_stage.o (tools/kernel_resources.py)"}
    _s["k_verify_id"] = _m
    json.dump(_s, open(f"{dst}/{tag}_summary.json", "w"), indent=1)


def filt(inp, out):
    rows = list(csv.reader(open(inp)))
    ki = rows[0].index("Kernel_Name")
    keep = [rows[0]] + [r for r in rows[1:] if any(k in r[ki] for k in DOMINANT)]
    csv.writer(open(out, "w", newline="")).writerows(keep)


filt(f"{src}/trace/{tag}_kernel_trace.csv", f"{dst}/{tag}_kernel_trace_k_verify_id.csv")
filt(f"{src}/pmc_a/{tag}_counter_collection.csv", f"{dst}/{tag}_pmc_a_k_verify_id.csv")
filt(f"{src}/pmc_b/{tag}_counter_collection.csv", f"{dst}/{tag}_pmc_b_k_verify_id.csv")
if os.path.exists(f"{src}/pmc_c/{tag}_counter_collection.csv"):
    filt(f"{src}/pmc_c/{tag}_counter_collection.csv", f"{dst}/{tag}_pmc_c_k_verify_id.csv")
s = json.load(open(f"{src}/summary.json"))
h = {"kernel": "k_verify_id_staged<BN254>", "items_per_launch": s["k_verify_id"]["grid"],
     "tag": tag, "window": s.get("window"),
     "build": "round-%s build (29-bit limbs, single-reduction Fp2 products, lazy sums, LDS hot slot, fused loops; plain layout, coalesced record loads, signed-digit W = %s tables)" % (tag.lstrip("r0"), s.get("window")),
     "FETCH_SIZE_KiB": s["pmc_per_launch"]["FETCH_SIZE"], "WRITE_SIZE_KiB": s["pmc_per_launch"]["WRITE_SIZE"],
     "k_verify_id_bytes_per_launch": s["k_verify_id_bytes_per_launch"],
     "k_verify_id_bytes_per_launch_uncorrected": s["k_verify_id_bytes_per_launch_uncorrected"], "l2_hit_rate": s["l2_hit_rate"],
     "note": "separate --pmc passes (tools/profile_round.sh, bench.py --headline-only so every launch has the headline size); FETCH_SIZE "
             "doubled per MI355X_MICROARCH.md; the traffic is per-lane private (scratch) memory of temporaries, call frames and "
             "callee-saved registers, not input data (52.7 MB algorithmic per launch)"}
json.dump(h, open(f"{dst}/hbm_traffic.json", "w"), indent=1)
print(json.dumps(s["pmc_per_launch"], indent=1))
