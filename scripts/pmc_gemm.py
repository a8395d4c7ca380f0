"""Launches the encoder's four dense products (and 4096^3) three times each, for a rocprofv3 --pmc FETCH_SIZE /
WRITE_SIZE pass:  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR -- python3 scripts/pmc_gemm.py
then  python3 scripts/pmc_gemm.py --reduce DIR  prints MB fetched per launch ((2 * FETCH_SIZE) KiB, see pmc_traffic.py)."""
import sys
SHAPES = [("qkv", 12000, 2304, 768, "bf16"), ("out", 12000, 768, 768, "f32+res"), ("fc1", 12000, 3072, 768, "bf16"),
          ("fc2", 12000, 768, 3072, "f32+res"), ("4096^3", 4096, 4096, 4096, "bf16")]
if "--reduce" in sys.argv:
    import csv, glob
    d = sys.argv[sys.argv.index("--reduce") + 1]
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    by = {}
    for r in csv.DictReader(open(f)):
        if "gemm_bf16_kernel" in r["Kernel_Name"]:
            k = int(r["Dispatch_Id"])
            by[k] = by.get(k, 0.0) + float(r["Counter_Value"])
    vals = [by[k] for k in sorted(by)]
    for i, (name, M, N, K, kind) in enumerate(SHAPES):
        v = vals[3 * i + 2]
        alg = 2 * (M * K + N * K) + (M * N * 4 if kind != "bf16" else 0)
        print(f"{name:7s} counter {v / 1024:8.1f} MiB raw, x2 = {2 * v * 1024 / 1e6:8.1f} MB; operands + residual once = {alg / 1e6:7.1f} MB")
    sys.exit(0)
import torch
sys.path.insert(0, '.')
from mocov2_whisper_flamingo_amd import ops
dev = torch.device('cuda:0')
for name, M, N, K, kind in SHAPES:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    out = torch.zeros(M, N, device=dev, dtype=torch.bfloat16 if kind == "bf16" else torch.float32)
    for _ in range(3):
        ops.gemm_nt(a, b, out=out, residual=None if kind == "bf16" else out)
torch.cuda.synchronize()
