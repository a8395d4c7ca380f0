"""Interleaved A/B timing of GEMM kernel variants (scripts/variants/libavsr_*.so) in ONE process."""
import ctypes, glob, os, sys
import torch
sys.path.insert(0, '.')
from mocov2_whisper_flamingo_amd import _lib
from mocov2_whisper_flamingo_amd._lib import GemmDesc

dev = torch.device('cuda:0')
libs = {}
for f in sorted(glob.glob('scripts/variants/libavsr_*.so')):
    l = ctypes.CDLL(os.path.abspath(f))
    l.avsr_gemm_nt.restype = ctypes.c_int
    l.avsr_gemm_nt.argtypes = [ctypes.c_void_p, ctypes.POINTER(GemmDesc)]
    libs[os.path.basename(f)[9:-3]] = l

def desc(a, b, out, conv=None):
    d = GemmDesc()
    d.in_dtype, d.out_dtype = 1, (1 if out.dtype == torch.bfloat16 else 0)
    d.N, d.K = b.shape
    d.A, d.B, d.C = a.data_ptr(), b.data_ptr(), out.data_ptr()
    d.ldb, d.ldc, d.batch, d.alpha = b.stride(0), out.stride(0), 1, 1.0
    if conv:
        H, C, k, s, pad, OH = conv
        d.conv.enabled, d.conv.H, d.conv.W, d.conv.C, d.conv.KH, d.conv.KW = 1, H, H, C, k, k
        d.conv.stride, d.conv.pad_h, d.conv.pad_w, d.conv.OH, d.conv.OW = s, pad, pad, OH, OH
        d.M = out.shape[0]
    else:
        d.M, d.lda = a.shape[0], a.stride(0)
    return d

cases = []
for name, M, N, K, odt in [("qkv", 12000, 2304, 768, torch.bfloat16), ("out", 12000, 768, 768, torch.float32),
                           ("fc1", 12000, 3072, 768, torch.bfloat16), ("fc2", 12000, 768, 3072, torch.float32),
                           ("4096^3", 4096, 4096, 4096, torch.bfloat16), ("tiny", 128, 128, 64, torch.bfloat16),
                           ("t512", 600, 512, 512, torch.bfloat16), ("t2048", 600, 512, 2048, torch.float32),
                           ("dW", 512, 2048, 600, torch.float32), ("t80", 600, 2048, 512, torch.bfloat16),
                           ("t128", 1024, 2048, 512, torch.bfloat16), ("t192", 1500, 2048, 512, torch.bfloat16)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=odt)
    cases.append((name, desc(a, b, out), 2.0 * M * N * K, (a, b, out)))
    if M == 12000:                      # the same product with the epilogue the encoder uses
        bias = torch.randn(N, device=dev)
        d = desc(a, b, out)
        d.bias = bias.data_ptr()
        if name == "fc1": d.act = 1
        if odt == torch.float32: d.residual, d.ldr = out.data_ptr(), out.stride(0)
        cases.append((name + "+e", d, 2.0 * M * N * K, (a, b, out, bias)))
for name, n, H, C, Co, k, s in [("l1.c2", 600, 25, 64, 64, 3, 1), ("l1.c3", 600, 25, 64, 256, 1, 1), ("l2.c2", 600, 13, 128, 128, 3, 1),
                                ("l3.c1", 600, 7, 1024, 256, 1, 1), ("l3.c2", 600, 7, 256, 256, 3, 1), ("l4.c2", 600, 4, 512, 512, 3, 1)]:
    pad = 1 if k == 3 else 0
    OH = (H + 2 * pad - k) // s + 1
    x = torch.randn(n, H, H, C, device=dev).to(torch.bfloat16); w = (torch.randn(Co, k * k * C, device=dev) * 0.05).to(torch.bfloat16)
    out = torch.empty(n * OH * OH, Co, device=dev, dtype=torch.bfloat16)
    cases.append((name, desc(x, w, out, (H, C, k, s, pad, OH)), 2.0 * n * OH * OH * Co * k * k * C, (x, w, out)))
    slab = torch.zeros(2, (n * OH * OH + 127) // 128, Co, device=dev)
    d = desc(x, w, out, (H, C, k, s, pad, OH))
    d.col_sum, d.col_sumsq = slab[0].data_ptr(), slab[1].data_ptr()
    cases.append((name + "+s", d, 2.0 * n * OH * OH * Co * k * k * C, (x, w, out, slab)))

stream = torch.cuda.current_stream().cuda_stream
ROUNDS, REP = 7, 10
res = {v: {c[0]: [] for c in cases} for v in libs}
for r in range(ROUNDS):
    for cname, d, fl, _ in cases:
        for v, l in libs.items():
            for _ in range(2): l.avsr_gemm_nt(stream, ctypes.byref(d))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REP): assert l.avsr_gemm_nt(stream, ctypes.byref(d)) == 0
            e1.record(); torch.cuda.synchronize()
            res[v][cname].append(fl / (e0.elapsed_time(e1) / REP * 1e-3) / 1e12)
print(f"{'case':8s} " + " ".join(f"{v:>14s}" for v in libs))
for cname, *_ in cases:
    print(f"{cname:8s} " + " ".join(f"{sorted(res[v][cname])[ROUNDS//2]:8.1f}/{max(res[v][cname]):5.0f}" for v in libs))
