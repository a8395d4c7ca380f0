"""Does a tiny kernel on one stream start while a large GEMM of another stream is still dispatching its workgroups?
For each of several streams (created in order, so they land on different hardware queues): launch the encoder's fc1
product (2256 workgroups, ~70 us) on stream A, then a one-thread kernel on stream B, and read when B's kernel finished
relative to the GEMM's start."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mocov2_whisper_flamingo_amd import ops

dev = torch.device("cuda:0")
a = torch.randn(12000, 768, device=dev).to(torch.bfloat16)
b = torch.randn(3072, 768, device=dev).to(torch.bfloat16)
out = torch.empty(12000, 3072, device=dev, dtype=torch.bfloat16)
slot = torch.zeros(1, dtype=torch.int64, device=dev)
sA = torch.cuda.Stream()
cands = [torch.cuda.Stream() for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8)]
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
for i, sB in enumerate(cands):
    lat, dur = [], []
    for it in range(12):
        torch.cuda.synchronize()
        e0, e1, f1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        with torch.cuda.stream(sA):
            e0.record()
            ops.gemm_nt(a, b, out=out)
            ops.gemm_nt(a, b, out=out)
            e1.record()
        with torch.cuda.stream(sB):
            ops.store_u64(slot, it)
            f1.record()
        torch.cuda.synchronize()
        if it >= 2:
            lat.append(e0.elapsed_time(f1) * 1e3)
            dur.append(e0.elapsed_time(e1) * 1e3)
    print(f"stream {i}: tiny kernel done {sum(lat)/len(lat):7.1f} us after the GEMMs' start (min {min(lat):6.1f}); the two GEMMs take {sum(dur)/len(dur):6.1f} us")
