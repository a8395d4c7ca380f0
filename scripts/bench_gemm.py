"""Micro-benchmark of the GEMM / implicit-GEMM conv kernel on the shapes the AV network uses."""
import sys, time
import torch
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

shapes = [  # encoder (B=8): M=12000
    ("qkv", 12000, 2304, 768, torch.bfloat16), ("out", 12000, 768, 768, torch.float32),
    ("fc1", 12000, 3072, 768, torch.bfloat16), ("fc2", 12000, 768, 3072, torch.float32),
    ("4096^3", 4096, 4096, 4096, torch.bfloat16), ("8192^3", 8192, 8192, 8192, torch.bfloat16),
    ("head", 600, 51865, 512, torch.float32),
]
for name, M, N, K, odt in shapes:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    ld = (N + 7) // 8 * 8
    out = torch.empty(M, ld, device=dev, dtype=odt)[:, :N]
    t = timeit(lambda: ops.gemm_nt(a, b, out=out, bias=bias))
    tr = timeit(lambda: ops.gemm_nt(a, b, out=out, bias=bias, residual=out))      # in-place residual update
    print(f"gemm {name:8s} M={M} N={N} K={K} out={str(odt)[6:]:8s} {t*1e6:9.1f} us  {2*M*N*K/t/1e12:7.1f} TF/s"
          f" | +residual {tr*1e6:9.1f} us")

convs = [  # (name, n_img, H, C, Cout, k, stride)
    ("l1.c1", 600, 25, 256, 64, 1, 1), ("l1.c2", 600, 25, 64, 64, 3, 1), ("l1.c3", 600, 25, 64, 256, 1, 1),
    ("l2.c1", 600, 13, 512, 128, 1, 1), ("l2.c2", 600, 13, 128, 128, 3, 1), ("l2.c3", 600, 13, 128, 512, 1, 1),
    ("l3.c1", 600, 7, 1024, 256, 1, 1), ("l3.c2", 600, 7, 256, 256, 3, 1), ("l3.c3", 600, 7, 256, 1024, 1, 1),
    ("l4.c1", 600, 4, 2048, 512, 1, 1), ("l4.c2", 600, 4, 512, 512, 3, 1), ("l4.c3", 600, 4, 512, 2048, 1, 1),
    ("l2.ds", 600, 25, 256, 512, 1, 2), ("l2.c2s", 600, 25, 128, 128, 3, 2),
]
for name, n, H, C, Co, k, s in convs:
    x = torch.randn(n, H, H, C, device=dev).to(torch.bfloat16)
    w = (torch.randn(Co, k * k * C, device=dev) * 0.05).to(torch.bfloat16)
    pad = 1 if k == 3 else 0
    OH = (H + 2 * pad - k) // s + 1
    fl = 2.0 * n * OH * OH * Co * k * k * C
    sc = torch.ones(Co, device=dev); sh = torch.zeros(Co, device=dev)
    t_eval = timeit(lambda: ops.conv_nhwc(x, w, k, k, s, pad, scale=sc, bias=sh, act=ops.ACT_RELU))
    st = ops.gemm_stats_slabs(n * OH * OH, Co, torch.bfloat16, dev)
    t_train = timeit(lambda: ops.conv_nhwc(x, w, k, k, s, pad, col_stats=(st[0], st[1])))
    print(f"conv {name:7s} M={n*OH*OH:7d} N={Co:5d} K={k*k*C:5d}  eval {t_eval*1e6:8.1f} us {fl/t_eval/1e12:6.1f} TF/s | train(raw+stats) {t_train*1e6:8.1f} us {fl/t_train/1e12:6.1f} TF/s")
