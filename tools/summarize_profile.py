#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (tools/profile_gpu.sh) into profiles/<tag>_rocprof.md and
profiles/traffic.json.  HBM traffic follows MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE reads half the bytes of a wide (16 B/lane) coalesced streaming read, so it is
doubled for k_jacobi_sweep (exactly that pattern); for the traversal kernel (scattered 16-B loads) the
factor is uncalibrated and both the raw and the doubled figure are listed."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
KERNELS = ["k_jacobi_sweep", "k_primary_closest", "k_form_factors_flat", "k_form_factors", "k_primary_work", "k_trace_rays"]   # (the longer name before its prefix)


def short(name):
    for k in KERNELS:
        if k in name:
            return k
    return None


lines = ["# rocprofv3 summary `%s` — `python3 bench.py --steps 5 --warmup 2 --no-cpu --ff-rows 4096` on one MI355X" % tag, "",
         "(k_form_factors: the figures are those of the 4096-row launch — the largest dispatch — not an average with the 8-row warm-up)", ""]
lines += ["## --kernel-trace --stats (rtmi kernels; torch helper kernels that build the synthetic inputs omitted)", "",
          "| kernel | calls | avg ms | min ms | max ms | % of GPU time |", "|---|---|---|---|---|---|"]
for row in csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_stats.csv"))):
    k = short(row["Name"])
    if k:
        lines.append("| %s | %s | %.4f | %.4f | %.4f | %s |" % (k, row["Calls"], float(row["AverageNs"]) / 1e6, float(row["MinNs"]) / 1e6,
                                                         float(row["MaxNs"]) / 1e6, row["Percentage"]))
# launch geometry from the trace
geo = {}
for row in csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_trace.csv"))):
    k = short(row["Kernel_Name"])
    if k and k not in geo:
        geo[k] = row
lines += ["", "| kernel | grid (threads) | workgroup | LDS B/wg | VGPR | SGPR | scratch |", "|---|---|---|---|---|---|---|"]
for k, r in geo.items():
    lines.append("| %s | %s | %s | %s | %s | %s | %s |" % (k, r.get("Grid_Size", r.get("Grid_Size_X")), r.get("Workgroup_Size", r.get("Workgroup_Size_X")),
                                                      r.get("LDS_Block_Size"), r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("Scratch_Size")))

pmc = defaultdict(lambda: defaultdict(list))     # kernel -> counter -> values per dispatch
dur = defaultdict(list)
for f in sorted(glob.glob(os.path.join(src, "pmc*", "pmc_counter_collection.csv"))):
    for row in csv.DictReader(open(f)):
        k = short(row["Kernel_Name"])
        if k:
            pmc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
lines += ["", "## --pmc passes (one run per counter group; averages per dispatch)", ""]
traffic = {}
for k in KERNELS:
    if k not in pmc:
        continue
    lines += ["### %s" % k, "", "| counter | avg per dispatch |", "|---|---|"]
    avg = {c: (max(v) if k == "k_form_factors" else sum(v) / len(v)) for c, v in pmc[k].items()}
    for c in sorted(avg):
        lines.append("| %s | %.6g |" % (c, avg[c]))
    lines.append("")
    if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
        rd, wr = avg["FETCH_SIZE"] * 1024, avg["WRITE_SIZE"] * 1024
        lines.append("HBM-side bytes per dispatch: FETCH_SIZE %.4g B raw (x2 = %.4g B, gfx950 wide-read correction), WRITE_SIZE %.4g B." % (rd, 2 * rd, wr))
        traffic[k] = {"fetch_raw": rd, "write": wr, "hbm_bytes": 2 * rd + wr}
    if "TCC_HIT_sum" in avg:
        traffic.setdefault(k, {})["l2_hit_rate"] = avg["TCC_HIT_sum"] / (avg["TCC_HIT_sum"] + avg["TCC_MISS_sum"])
        lines.append("L2 hit rate = TCC_HIT/(TCC_HIT+TCC_MISS) = %.3f." % (avg["TCC_HIT_sum"] / (avg["TCC_HIT_sum"] + avg["TCC_MISS_sum"])))
    if "SQ_WAVE_CYCLES" in avg and "SQ_ACTIVE_INST_VALU" in avg:
        traffic.setdefault(k, {})["valu_active_share_of_wave_cycles"] = avg["SQ_ACTIVE_INST_VALU"] / avg["SQ_WAVE_CYCLES"]
        traffic[k]["wait_any_share_of_wave_cycles"] = avg["SQ_WAIT_ANY"] / avg["SQ_WAVE_CYCLES"]
        if "SQ_THREAD_CYCLES_VALU" in avg:
            traffic[k]["lanes_active_per_valu_instruction"] = avg["SQ_THREAD_CYCLES_VALU"] / avg["SQ_ACTIVE_INST_VALU"]
        lines.append("Wave-cycle shares: WAIT_ANY %.2f, WAIT_INST_ANY %.2f, ACTIVE_INST_ANY %.2f; VALU-active share of wave cycles %.2f; VALU insts per wave %.0f." % (
            avg["SQ_WAIT_ANY"] / avg["SQ_WAVE_CYCLES"], avg["SQ_WAIT_INST_ANY"] / avg["SQ_WAVE_CYCLES"], avg["SQ_ACTIVE_INST_ANY"] / avg["SQ_WAVE_CYCLES"],
            avg["SQ_ACTIVE_INST_VALU"] / avg["SQ_WAVE_CYCLES"], avg["SQ_INSTS_VALU"] / avg["SQ_WAVES"]))
    if "SQ_THREAD_CYCLES_VALU" in avg and "SQ_ACTIVE_INST_VALU" in pmc[k]:
        pass
    lines.append("")
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
open(os.path.join(root, "profiles", tag + "_rocprof.md"), "w").write("\n".join(lines) + "\n")
# traffic.json: what bench.py reports as roofline.traffic (keyed by the launch it was measured on)
n = 65536
out = {}
if "k_jacobi_sweep" in traffic and "hbm_bytes" in traffic["k_jacobi_sweep"]:
    out["k_jacobi_sweep"] = {"bytes_per_launch": 4.0 * n * n, "hbm_bytes": traffic["k_jacobi_sweep"]["hbm_bytes"], "source": tag}
if "k_primary_closest" in traffic and "hbm_bytes" in traffic["k_primary_closest"]:
    t = traffic["k_primary_closest"]
    out["k_primary_closest"] = {"rays_per_launch": 3840 * 2160, "hbm_bytes": t["hbm_bytes"], "hbm_bytes_uncorrected": t["fetch_raw"] + t["write"], "source": tag}
    for key in ("l2_hit_rate", "valu_active_share_of_wave_cycles", "wait_any_share_of_wave_cycles", "lanes_active_per_valu_instruction"):
        if key in t:
            out["k_primary_closest"][key] = round(t[key], 4)
    a = {c: sum(v) / len(v) for c, v in pmc["k_primary_closest"].items()}
    if "SQ_INSTS_VALU" in a and "GRBM_GUI_ACTIVE" in a:
        # a wave64 vector instruction occupies its SIMD's issue for 4 cycles (tools/ub/valu_rate.hip: 3.1 fma_f32 ... 4.7);
        # 1024 SIMDs; GRBM_GUI_ACTIVE sums the 8 XCDs' cycles of the launch
        out["k_primary_closest"]["valu_issue_utilisation"] = round(a["SQ_INSTS_VALU"] * 4.0 / (1024.0 * a["GRBM_GUI_ACTIVE"] / 8.0), 3)
        out["k_primary_closest"]["valu_instructions_per_ray"] = round(a["SQ_INSTS_VALU"] / (3840 * 2160), 1)
if "k_form_factors" in traffic:
    t = traffic["k_form_factors"]
    a = {c: max(v) for c, v in pmc["k_form_factors"].items()}
    n_box = 64744 + 1
    rows = 4096
    entries = rows * n_box
    ff = {"rows_per_launch": rows, "source": tag + "_rocprof.md"}
    if "fetch_raw" in t:
        ff["hbm_bytes_per_launch"] = t["fetch_raw"] + t["write"]
        ff["hbm_bytes_per_entry_fetch"] = round(t["fetch_raw"] / entries, 2)
        ff["hbm_bytes_per_entry_write"] = round(t["write"] / entries, 2)
        ff["note"] = "FETCH_SIZE uncorrected (scattered 16-byte loads: the x2 of a streaming read does not apply); WRITE_SIZE is exact for these store patterns (profiles/README.md, calibration): the bytes above 4 per entry are partial 64-byte sectors of scattered patch ids"
    if "l2_hit_rate" in t:
        ff["l2_hit_rate"] = round(t["l2_hit_rate"], 4)
    ms = [float(r["AverageNs"]) for r in csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_stats.csv"))) if "k_form_factors" in r["Name"] and "flat" not in r["Name"]]
    mx = [float(r["MaxNs"]) for r in csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_stats.csv"))) if "k_form_factors" in r["Name"] and "flat" not in r["Name"]]
    if mx and "SQ_INSTS_VALU" in a:
        # 1024 SIMDs, a wave64 vector instruction occupies its SIMD for 4 cycles, 2.4 GHz nominal
        ff["valu_issue_utilisation"] = round(a["SQ_INSTS_VALU"] * 4.0 / (1024 * mx[0] * 1e-9 * 2.4e9), 3)
        ff["valu_instructions_per_lane_sample"] = round(a["SQ_INSTS_VALU"] / (a["SQ_WAVES"] * 64 * 10), 1)
        ff["kernel_ms"] = round(mx[0] / 1e6, 3)
    g = geo.get("k_form_factors")
    if g:
        ff["scratch_bytes"] = int(g.get("Scratch_Size") or 0)
        ff["vgprs_in_rocprof_trace"] = int(g.get("VGPR_Count") or 0) + int(g.get("Accum_VGPR_Count") or 0)   # the trace's unit is not registers per lane
    try:   # registers per lane from the code object's own note (tools/kernel_resources.sh compiles the device side, ~20 s)
        import re, subprocess
        notes = subprocess.run(["bash", os.path.join(root, "tools", "kernel_resources.sh"), os.path.join(root, "raytracer_amd", "csrc", "rtmi_formfactor.hip")],
                               check=True, capture_output=True, text=True).stdout
        m = re.search(r"_ZN4rtmi14k_form_factorsE\S*\s+\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_count:\s+(\d+)", notes)
        if m:
            ff["scratch_bytes"], ff["vgprs"] = int(m.group(1)), int(m.group(2))
    except Exception:
        pass
    out["k_form_factors"] = ff
json.dump(out, open(os.path.join(root, "profiles", "traffic.json"), "w"), indent=1)
print("\n".join(lines))
