"""Fold the two SQ counter passes of tools/profile_round.sh into profiles/r01_pmc_sq_counters.json."""
import json, subprocess, sys
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
out = {"command": "rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --kernel-trace -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --serial ; second pass --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE",
       "note": "per-launch averages (configs[2]: 128 images per front-end launch, 64 per FCN launch). SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are in quad-cycles summed over waves; SQ_BUSY_CU_CYCLES summed over 256 CUs; SQ_VALU_MFMA_BUSY_CYCLES summed over 1024 SIMDs (32 per v_mfma_f32_32x32x16_f16)",
       "per_launch_avg": {}}
d = {}
for f in ("sq1", "sq2"):
    j = json.loads(subprocess.check_output([sys.executable, "tools/pmc_summary.py", "%s/%s_counter_collection.csv" % (src, f), ""]))
    for k, v in j.items():
        if "rocclr" in k or "at::" in k or "elementwise" in k: continue
        d.setdefault(k, {}).update({c: int(x) for c, x in v.items() if c != "launches"})
for k, v in sorted(d.items()):
    wc = v.get("SQ_WAVE_CYCLES", 0); busy = v.get("SQ_BUSY_CU_CYCLES", 0)
    if wc:
        v["wait_any_frac_of_wave_cycles"] = round(v.get("SQ_WAIT_ANY", 0) / wc, 3)
        v["wait_inst_frac_of_wave_cycles"] = round(v.get("SQ_WAIT_INST_ANY", 0) / wc, 3)
        v["valu_inst_frac_of_wave_cycles"] = round(v.get("SQ_INSTS_VALU", 0) / wc, 3)
    if busy and v.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        v["mfma_pipe_busy_frac"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (busy / 256), 3)
    out["per_launch_avg"][k] = v
json.dump(out, open("profiles/r01_pmc_sq_counters.json", "w"), indent=1)
print(len(out["per_launch_avg"]), "kernels")
