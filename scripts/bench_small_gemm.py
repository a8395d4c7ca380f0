"""Small (tail) GEMM shapes, one launch group per shape; run under rocprofv3 --kernel-trace for true durations."""
import sys, torch
sys.path.insert(0, '.')
from mocov2_whisper_flamingo_amd import ops
dev = torch.device('cuda:0')
shapes = [(600, 512, 512), (600, 512, 2048), (600, 2048, 512), (512, 512, 600), (600, 512, 768), (512, 2048, 600),
          (2048, 512, 600), (600, 1024, 512), (600, 512, 1024), (128, 128, 64), (128, 128, 4096)]
for M, N, K in shapes:
    Kp = (K + 7) // 8 * 8
    a = torch.randn(M, Kp, device=dev).to(torch.bfloat16)[:, :K]
    b = torch.randn(N, Kp, device=dev).to(torch.bfloat16)[:, :K]
    for odt in (torch.bfloat16, torch.float32):
        out = torch.empty(M, N, device=dev, dtype=odt)
        for _ in range(5): ops.gemm_nt(a, b, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): ops.gemm_nt(a, b, out=out)
        e1.record(); torch.cuda.synchronize()
        print(f"M={M} N={N} K={K} out={str(odt)[6:]}: {e0.elapsed_time(e1)/50*1e3:7.1f} us/launch (back-to-back)", flush=True)
