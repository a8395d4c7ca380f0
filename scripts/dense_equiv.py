import sys, torch
sys.path.insert(0, '.')
from mocov2_whisper_flamingo_amd import ops
dev = torch.device('cuda:0')
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for name, n, H, C, Co, s in (("l1.c2", 600, 25, 64, 64, 1), ("l2.c2", 600, 13, 128, 128, 1), ("l3.c2", 600, 7, 256, 256, 1), ("l4.c2", 600, 4, 512, 512, 1)):
    M, K = n * H * H, 9 * C
    x = torch.randn(n, H, H, C, device=dev).to(torch.bfloat16)
    w = (torch.randn(Co, K, device=dev) * 0.05).to(torch.bfloat16)
    tc = timeit(lambda: ops.conv_nhwc(x, w, 3, 3, 1, 1))
    # overlapping-row dense operand over a W-padded copy: rows of 3C contiguous values, 3 such segments are NOT contiguous,
    # so this only times K = 3C x 3 separate products as a proxy: one product with K = 9C on an im2col-free overlapped view of stride C
    xp = torch.randn(M + 8 * H, C, device=dev).to(torch.bfloat16)
    a = torch.as_strided(xp, (M, K), (C, 1))           # rows overlap: same bytes-from-cache behaviour as the conv
    td = timeit(lambda: ops.gemm_nt(a, w))
    print(f"{name}: conv {tc*1e6:6.1f} us {2*M*Co*K/tc/1e12:6.1f} TF | dense overlapped-row {td*1e6:6.1f} us {2*M*Co*K/td/1e12:6.1f} TF")
