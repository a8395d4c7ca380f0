"""MFMA-busy of one gated cross-attention block (forward + backward through avsr_gated_xattn_fwd / _bwd) from a
rocprofv3 counter pass and the kernel trace of the same run:

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d DIR -- \\
        python3 scripts/bench_xattn_block.py B,T,fused --once
    python3 scripts/pmc_xattn.py DIR B T

SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe cycles summed over the SIMDs; GRBM_GUI_ACTIVE the busy cycles summed
over the 8 XCDs.  Per kernel: mfma_busy = MFMA cycles / (GUI_ACTIVE / 8 * 1024 SIMDs).  The block's kernels are the
dispatches from the first layernorm_fwd of the run to the last layernorm_bwd; algorithmic FLOPs as in
bench_xattn_block.py; 'of_peak' = FLOPs / sum of kernel durations / 2.5 PFLOP/s."""
import csv, glob, json, sys

d, B, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
vals = {}
for r in csv.DictReader(open(cc)):
    k = int(r["Dispatch_Id"])
    e = vals.setdefault(k, {"name": r["Kernel_Name"]})
    e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
dur = {}
for r in csv.DictReader(open(kt)):
    dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
ids = sorted(vals)
first = next(i for i in ids if "layernorm_fwd" in vals[i]["name"])
last = max(i for i in ids if "layernorm_bwd" in vals[i]["name"])
fwd_end = next(i for i in ids if i > first and "gated_residual_bwd" in vals[i]["name"])
dm, H = 512, 8
rows = B * T
gemm = rows * (2 * dm * dm + 2 * dm * dm + 4 * dm * 4 * dm) + rows * 4 * dm * dm
attn = 4 * B * H * T * T * 64
out = {"shape": {"B": B, "Tq": T, "Tk": T, "rows": rows, "d": dm, "heads": H}, "kernels": []}
for part, lo, hi, fl in (("forward", first, fwd_end - 1, gemm + attn), ("backward", fwd_end, last, 2 * gemm + 2.5 * attn)):
    mf = ga = us = 0.0
    n = 0
    for i in ids:
        if lo <= i <= hi:
            e = vals[i]
            m, g = e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), e.get("GRBM_GUI_ACTIVE", 0.0)
            mf += m; ga += g; us += dur.get(i, 0.0); n += 1
            short = e["name"].split("(")[0]
            short = short[:60]
            out["kernels"].append({"part": part, "kernel": short, "us": round(dur.get(i, 0.0), 1),
                                   "mfma_busy": round(m / (g / 8 * 1024), 4) if g else None})
    out[part] = {"launches": n, "sum_kernel_us": round(us, 1), "algorithmic_gflop": round(fl / 1e9, 2),
                 "tflops": round(fl / us / 1e6, 1), "of_peak_2500": round(fl / us / 1e6 / 2500, 4),
                 "mfma_busy_counter": round(mf / (ga / 8 * 1024), 4)}
print(json.dumps(out, indent=1))
