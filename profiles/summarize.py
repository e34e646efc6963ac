"""Summarise rocprofv3 outputs (gpurun_out/<tag>_{trace,pmc_*}) into profiles/<tag>_* files.
usage: python profiles/summarize.py <tag> [workload] [pmc_scale]
pmc_scale = spp of the traced launch / spp of the PMC launches (collect.sh EXTRA="--spp N"): instruction counts scale
with spp, so the derived per-launch instruction totals are multiplied by it to match avg_ms."""
import collections
import csv
import glob
import json
import shutil
import sys

tag = sys.argv[1]
pmc_scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
out = {"workload": sys.argv[2] if len(sys.argv) > 2 else "cornell_1080p_1024spp_b8", "n_gpus": 1, "pmc_scale": pmc_scale,
       "command": "python3 bench.py --no-cpu-baseline (trace) / --steps 1 --warmup 0 (pmc passes)"}
ks = glob.glob(f"gpurun_out/{tag}_trace/**/*kernel_stats.csv", recursive=True)
if ks:
    shutil.copy(ks[0], f"profiles/{tag}_kernel_stats.csv")
    for r in csv.DictReader(open(ks[0])):
        if "pt_megakernel" in r["Name"]:
            out["kernel"] = r["Name"]
            out["calls"] = int(r["Calls"])
            out["avg_ms"] = float(r["AverageNs"]) / 1e6
            out["pct_gpu_time"] = float(r["Percentage"])
pmc = {}
meta = {}
for f in sorted(glob.glob(f"gpurun_out/{tag}_pmc_*/**/*counter_collection.csv", recursive=True)):
    one = collections.defaultdict(float)          # one --pmc pass = one launch of the kernel
    for r in csv.DictReader(open(f)):
        if "pt_megakernel" in r["Kernel_Name"]:
            one[r["Counter_Name"]] += float(r["Counter_Value"])
            meta = {k: r[k] for k in ("VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size") if k in r}
    for k, v in one.items():
        pmc.setdefault(k, v)                      # a counter collected in several passes: keep the first pass
out["pmc_per_launch"] = dict(pmc)
out["dispatch"] = meta
d = {}
if "SQ_INSTS_VALU" in pmc:
    d["valu_wave_instr"] = pmc["SQ_INSTS_VALU"] * pmc_scale
    d["lane_utilisation"] = pmc["SQ_THREAD_CYCLES_VALU"] / (pmc["SQ_ACTIVE_INST_VALU"] * 64.0)
if "SQ_INSTS_VALU_FLOPS_FP32" in pmc and "SQ_INSTS_VALU" in pmc:
    # dynamic instruction mix (wave-instructions per launch); FLOPS_FP32 counts add + mul + 2 x fma + trans
    d["mix"] = {k.replace("SQ_INSTS_VALU_", "").lower(): pmc[k] / pmc["SQ_INSTS_VALU"] for k in pmc if k.startswith("SQ_INSTS_VALU_")}
    d["fp32_flop_wave_instr"] = pmc["SQ_INSTS_VALU_FLOPS_FP32"] * pmc_scale
if "GRBM_GUI_ACTIVE" in pmc and "SQ_INSTS_VALU" in pmc:
    # GRBM_GUI_ACTIVE sums the busy cycles of the 8 XCDs (checked on microbench/valu_types.hip, whose instruction counts are exact):
    # cycles of ONE shader clock = / 8.  VALU wave-instructions per SIMD per shader cycle -- against 0.5 for a stream of full-rate
    # opcodes with VGPR operands and 0.25 for the half-rate class (compares, selects, min / max, SGPR-operand forms, ...)
    d["shader_cycles"] = pmc["GRBM_GUI_ACTIVE"] / 8.0
    d["valu_instr_per_simd_cycle"] = pmc["SQ_INSTS_VALU"] / 1024.0 / d["shader_cycles"]
if "SQ_WAIT_ANY" in pmc and "SQ_WAVE_CYCLES" in pmc:
    d["wait_any_frac"] = pmc["SQ_WAIT_ANY"] / pmc["SQ_WAVE_CYCLES"]
    d["wait_inst_any_frac"] = pmc["SQ_WAIT_INST_ANY"] / pmc["SQ_WAVE_CYCLES"]
    d["active_inst_any_frac"] = pmc["SQ_ACTIVE_INST_ANY"] / pmc["SQ_WAVE_CYCLES"]
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    # MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB-units of 1024 B; on gfx950 FETCH_SIZE reads half
    # the bytes of a coalesced streaming read -> doubled; WRITE_SIZE is exact.  Separate --pmc passes.
    d["hbm_read_bytes"] = 2.0 * pmc["FETCH_SIZE"] * 1024.0
    d["hbm_write_bytes"] = pmc["WRITE_SIZE"] * 1024.0
    d["hbm_traffic_bytes"] = d["hbm_read_bytes"] + d["hbm_write_bytes"]
if "fp32_flop_wave_instr" in d and out.get("avg_ms") and "lane_utilisation" in d:
    t = out["avg_ms"] * 1e-3
    d["fp32_tflops_issued"] = d["fp32_flop_wave_instr"] * 64 / t / 1e12                       # every lane of an issued instruction
    d["fp32_tflops_useful"] = d["fp32_tflops_issued"] * d["lane_utilisation"]               # active lanes only
out["derived"] = d
json.dump(out, open(f"profiles/{tag}_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
