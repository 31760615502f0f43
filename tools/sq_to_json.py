"""Fold the two SQ counter passes of tools/profile_round.sh (gpurun_out/prof/sq{1,2}_summary.json) into profiles/<tag>_pmc_sq_counters.json.

usage: python tools/sq_to_json.py [src_dir] [tag]"""
import json, sys
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
out = {"command": "rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --kernel-trace -- python3 bench.py --steps 1 --warmup 1 --batches-per-step 4 --no-cpu-baseline --no-extras --serial ; second pass --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE",
       "note": "per-launch averages (configs[2]: 256 images per front-end launch, 64 per FCN launch since r06 (128 before)). SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are in quad-cycles summed over waves; SQ_BUSY_CU_CYCLES summed over 256 CUs; SQ_VALU_MFMA_BUSY_CYCLES summed over 1024 SIMDs (32 per v_mfma_f32_32x32x16_f16); GRBM_GUI_ACTIVE summed over 8 XCDs. valu_issue_frac_at_4cyc = SQ_INSTS_VALU * 4 / (GRBM_GUI_ACTIVE / 8 * 1024): 1.0 = every SIMD issues a vector instruction every 4 cycles",
       "per_launch_avg": {}}
d = {}
for f in ("sq1", "sq2"):
    j = json.load(open("%s/%s_summary.json" % (src, f)))
    for k, v in j.items():
        if "rocclr" in k or "at::" in k or "elementwise" in k: continue
        d.setdefault(k, {}).update({c: int(x) for c, x in v.items() if c != "launches"})
for k, v in sorted(d.items()):
    wc = v.get("SQ_WAVE_CYCLES", 0); gui = v.get("GRBM_GUI_ACTIVE", 0)
    if wc:
        v["wait_any_frac_of_wave_cycles"] = round(v.get("SQ_WAIT_ANY", 0) / wc, 3)
        v["wait_inst_frac_of_wave_cycles"] = round(v.get("SQ_WAIT_INST_ANY", 0) / wc, 3)
    if gui:
        simd = gui / 8 * 1024
        v["valu_issue_frac_at_4cyc"] = round(v.get("SQ_INSTS_VALU", 0) * 4 / simd, 3)
        if v.get("SQ_VALU_MFMA_BUSY_CYCLES"): v["mfma_pipe_busy_frac"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / simd, 3)
        if wc: v["waves_per_simd"] = round(wc * 4 / simd, 2)
    out["per_launch_avg"][k] = v
json.dump(out, open("profiles/%s_pmc_sq_counters.json" % tag, "w"), indent=1)
print(len(out["per_launch_avg"]), "kernels")
