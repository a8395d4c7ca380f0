"""Reduce a rocprofv3 --pmc + --kernel-trace run of scripts/pmc_attn.py to the attention forward kernel's figures:
    python3 scripts/pmc_attn_reduce.py DIR
MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); VALU busy likewise from
SQ_ACTIVE_INST_VALU (counted in quad-cycles: x4)."""
import csv, glob, json, sys
d = sys.argv[1]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
vals = {}
for r in csv.DictReader(open(cc)):
    if "attn_fwd_mfma" not in r["Kernel_Name"]:
        continue
    e = vals.setdefault(int(r["Dispatch_Id"]), {})
    e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
dur = {int(r["Dispatch_Id"]): (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3 for r in csv.DictReader(open(kt))
       if "attn_fwd_mfma" in r["Kernel_Name"]}
i = sorted(vals)[-1]
e = vals[i]
simd_cycles = e["GRBM_GUI_ACTIVE"] / 8 * 1024
out = {"kernel": "attn_fwd_mfma_kernel<false>", "shape": "B=8 H=12 T=1500 d=64", "us_under_counters": round(dur[i], 1),
       "counters": {k: v for k, v in e.items()},
       "mfma_busy": round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles, 4)}
if "SQ_ACTIVE_INST_VALU" in e:
    out["valu_busy"] = round(4 * e["SQ_ACTIVE_INST_VALU"] / simd_cycles, 4)
if "SQ_INSTS_VALU" in e:
    out["valu_insts_per_wave_subtile"] = round(e["SQ_INSTS_VALU"] / (8 * 12 * 12 * 4 * 47), 1)
print(json.dumps(out, indent=1))
