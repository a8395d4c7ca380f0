"""How do an HBM-bound and an MFMA-bound kernel behave on a stream restricted to part of the CUs
(hipExtStreamCreateWithCUMask)?  Prints the rate of bn_apply and of a Whisper-size GEMM for several masks,
alone and with both kernels running side by side on complementary masks."""
import ctypes, sys, torch
sys.path.insert(0, '.')
from mocov2_whisper_flamingo_amd import ops
dev = torch.device('cuda:0')
torch.cuda.init(); torch.zeros(1, device=dev)
hip = ctypes.CDLL("libamdhip64.so")

def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xffffffff for i in range(8)])
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)

def mask_first(n):            # the n lowest bits
    return (1 << n) - 1
def mask_range(lo, hi):
    return ((1 << hi) - 1) ^ ((1 << lo) - 1)

rows, C = 375000, 256
x = torch.randn(rows, C, device=dev).to(torch.bfloat16)
r = torch.randn(rows, C, device=dev).to(torch.bfloat16)
sc, sh = torch.randn(C, device=dev), torch.randn(C, device=dev)
a = torch.randn(12000, 768, device=dev).to(torch.bfloat16)
b = torch.randn(3072, 768, device=dev).to(torch.bfloat16)
out = torch.empty(12000, 3072, device=dev, dtype=torch.bfloat16)
def bn(): ops.bn_apply(x, sc, sh, residual=r, relu=True, out=x)
def gm(): ops.gemm_nt(a, b, out=out)
gb = rows * C * 2 * 3 / 1e9
gf = 2.0 * 12000 * 3072 * 768 / 1e12

def timeit(fn, stream, n=20):
    with torch.cuda.stream(stream):
        for _ in range(3): fn()
        stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(n): fn()
        e1.record(stream); stream.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

full = torch.cuda.Stream()
print(f"unmasked: bn_apply {gb/timeit(bn, full)/1e3:5.2f} TB/s   gemm {gf/timeit(gm, full):6.1f} TF/s", flush=True)
for n in (32, 64, 96, 128, 192, 256):
    s = masked_stream(mask_first(n))
    print(f"first {n:3d} bits: bn_apply {gb/timeit(bn, s)/1e3:5.2f} TB/s   gemm {gf/timeit(gm, s):6.1f} TF/s", flush=True)
def both(sa, sb, n=20):
    for st, fn in ((sa, gm), (sb, bn)):
        with torch.cuda.stream(st):
            for _ in range(3): fn()
    torch.cuda.synchronize()
    ea0, ea1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    eb0, eb1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ea0.record(sa); eb0.record(sb)
    for _ in range(n):
        with torch.cuda.stream(sa): gm()
        with torch.cuda.stream(sb): bn(); 
    ea1.record(sa); eb1.record(sb); torch.cuda.synchronize()
    return ea0.elapsed_time(ea1) / n * 1e-3, eb0.elapsed_time(eb1) / n * 1e-3

s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
ta, tb = both(s1, s2)
print(f"side by side, unmasked: gemm {gf/ta:6.1f} TF/s ({ta*1e6:.1f} us)  bn_apply {gb/tb/1e3:5.2f} TB/s ({tb*1e6:.1f} us)", flush=True)
for nv in (48, 64, 96):
    sv = masked_stream(mask_first(nv)); sa = masked_stream(mask_range(nv, 256))
    ta, tb = both(sa, sv)
    print(f"side by side, bn on first {nv}, gemm on the rest: gemm {gf/ta:6.1f} TF/s ({ta*1e6:.1f} us)  bn_apply {gb/tb/1e3:5.2f} TB/s ({tb*1e6:.1f} us)", flush=True)
