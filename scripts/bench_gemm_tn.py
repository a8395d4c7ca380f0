"""gemm_tn (dW = dY^T X from row-major operands) against the contraction length and the output size."""
import sys, torch
sys.path.insert(0, '.')
from mocov2_whisper_flamingo_amd import ops
dev = torch.device('cuda:0')
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M, N, K in [(64, 512, 512), (128, 512, 512), (600, 512, 512), (1200, 512, 512), (2400, 512, 512), (600, 2048, 512), (3200, 512, 512), (3200, 2048, 512), (3200, 512, 2048), (12000, 768, 768),
                (600, 512, 2048), (64, 51865, 512), (600, 51865, 512)]:
    ldn = (N + 7) // 8 * 8
    dy = torch.randn(M, ldn, device=dev).to(torch.bfloat16)[:, :N]
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    out = torch.zeros(N, K, device=dev)
    cs = torch.zeros(N, device=dev)
    t0 = timeit(lambda: ops.gemm_tn(dy, x, out=out, accumulate=True, colsum=cs, accumulate_colsum=True))
    t1 = timeit(lambda: ops.gemm_tn(dy, x, out=out, accumulate=False))
    print(f"M={M:5d} N={N:6d} K={K:5d}: accumulate+colsum {t0:7.1f} us | plain {t1:7.1f} us  {2.0*M*N*K/t1/1e6:7.1f} TF/s", flush=True)
