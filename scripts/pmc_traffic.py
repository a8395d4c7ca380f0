"""HBM bytes per launch of the GEMM kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- \\
        python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-graph --no-pipeline
    (the same with WRITE_SIZE into gpurun_out/pmc_write), then
    python3 scripts/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/r01_pmc_traffic.json

Both counters are in KiB; on gfx950 FETCH_SIZE counts the 128-byte requests of 16-byte-per-lane streams at 64 bytes,
so HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  The last step's launches are used (3 steps run: warm-up + 2)."""
import csv, glob, json, sys


def per_dispatch(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and "gemm_bf16_kernel" in r["Kernel_Name"]]
    by = {}
    for r in rows:
        by[int(r["Dispatch_Id"])] = by.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    vals = [by[k] for k in sorted(by)]
    n = len(vals) // 3                       # launches per step
    return vals[-n:], n


fetch, n = per_dispatch(sys.argv[1], "FETCH_SIZE")
write, n2 = per_dispatch(sys.argv[2], "WRITE_SIZE")
assert n == n2, (n, n2)
print(json.dumps({
    "command": "rocprofv3 --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) --kernel-trace -- python3 bench.py --steps 2 "
               "--warmup 1 --no-cpu-baseline --no-roofline --no-graph --no-pipeline",
    "kernel": "gemm_bf16_kernel (all instantiations), last step", "launches": n,
    "fetch_size_kb_sum": sum(fetch), "write_size_kb_sum": sum(write),
    "correction": "HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: on gfx950 FETCH_SIZE counts 128-B requests of "
                  "16-B-per-lane streams at 64 B",
    "hbm_bytes_per_launch": (2 * sum(fetch) + sum(write)) * 1024 / n}, indent=1))
