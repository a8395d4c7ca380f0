"""Times of the Gram-statistics kernels (avsr_gram_stats + avsr_bn_gram_finalize) at the layer1 / layer2 shapes."""
import sys, torch
sys.path.insert(0, '.')
from mocov2_whisper_flamingo_amd import ops
dev = torch.device('cuda:0')
def timed(fn, n=20, warm=3):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M, K, N in ((375000, 64, 256), (101400, 128, 512)):
    h = torch.randn(M, K, device=dev).abs().to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    g, b = torch.ones(N, device=dev), torch.zeros(N, device=dev)
    aff = torch.empty(2, N, device=dev)
    gp, cp = ops.gram_stats(h)
    flush = torch.empty(64 * 1024 * 1024, device=dev)
    t1 = timed(lambda: ops.gram_stats(h))
    t2 = timed(lambda: ops.bn_gram_finalize(gp, cp, M, w, g, b, None, None, aff[0], aff[1]))
    print(f"M={M} K={K}: parts {gp.shape[0]}, gram_stats {t1:.1f} us ({M * K * 2 / t1 / 1e6:.2f} TB/s of input), reduce + finalize {t2:.1f} us")
