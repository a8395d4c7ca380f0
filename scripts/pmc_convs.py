"""Per-launch HBM traffic and MFMA-busy of every GEMM / implicit-GEMM convolution of the visual frontend, from three
separate rocprofv3 counter passes over `python3 scripts/visual_only.py 2 --eager` (FETCH_SIZE; WRITE_SIZE;
SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; each with --kernel-trace) and the launch-order shape list that
`scripts/visual_only.py --dump-shapes` writes:

    python3 scripts/pmc_convs.py DIR_FETCH DIR_WRITE DIR_MFMA SHAPES.json > profiles/r03_conv_pmc.json

HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts the 128-byte requests of 16-byte-per-lane
streams at 64 bytes).  Algorithmic bytes of a launch = A read once + B read once + C written once (+ the residual read
once), bf16.  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)."""
import csv, glob, json, sys


def counters(d):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    by = {}
    for r in csv.DictReader(open(f)):
        e = by.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"]})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return by


def durations(d):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    return {int(r["Dispatch_Id"]): (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3 for r in csv.DictReader(open(f))}


def gemms_of_last_pass(by, n):
    ids = [i for i in sorted(by) if "gemm_bf16_kernel" in by[i]["name"]]
    return ids[-n:]


shapes = json.load(open(sys.argv[4]))
n = len(shapes)
fe, wr, mf = counters(sys.argv[1]), counters(sys.argv[2]), counters(sys.argv[3])
dur = durations(sys.argv[3])
out, tot_alg, tot_hbm, tot_us = [], 0.0, 0.0, 0.0
for s, i_f, i_w, i_m in zip(shapes, gemms_of_last_pass(fe, n), gemms_of_last_pass(wr, n), gemms_of_last_pass(mf, n)):
    M, N, K = s["M"], s["N"], s["K"]
    conv = s["conv"]
    a_elems = M * K
    if conv:                                   # the input tensor is read once, whatever the filter size
        H, W, C, KH, KW, stride, ph, pw, OH, OW = conv
        a_elems = (M // (OH * OW)) * H * W * C
        if KH == 1 and stride > 1:             # a strided 1x1 touches only the sampled pixels
            a_elems = M * C
    hbm = (2 * fe[i_f].get("FETCH_SIZE", 0.0) + wr[i_w].get("WRITE_SIZE", 0.0)) * 1024
    alg = 2.0 * (a_elems + N * K + M * N)
    us = dur[i_m]
    g = mf[i_m].get("GRBM_GUI_ACTIVE", 0.0)
    busy = mf[i_m].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (g / 8 * 1024) if g else None
    out.append({"M": M, "N": N, "K": K, "conv": conv, "kernel": mf[i_m]["name"].split("(")[0][-60:], "us": round(us, 1),
                "tflops": round(2.0 * M * N * K / us / 1e6, 1), "hbm_mb": round(hbm / 1e6, 1),
                "algorithmic_mb_without_residual": round(alg / 1e6, 1), "hbm_gbps": round(hbm / us / 1e3, 0),
                "mfma_busy": None if busy is None else round(busy, 3)})
    tot_alg += alg; tot_hbm += hbm; tot_us += us
print(json.dumps({"command": "rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE> --kernel-trace "
                             "--output-format csv -- python3 scripts/visual_only.py 2 --eager  (three passes)",
                  "launches": n, "sum_us": round(tot_us, 1), "hbm_mb_sum": round(tot_hbm / 1e6, 1),
                  "algorithmic_mb_sum_without_residual_reads": round(tot_alg / 1e6, 1), "per_launch": out}, indent=1))
