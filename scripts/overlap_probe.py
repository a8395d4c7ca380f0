"""An MFMA-bound GEMM (Whisper fc1 shape) and the HBM-bound bn_apply, each alone and side by side on two streams."""
import os, sys, torch
sys.path.insert(0, '.')
from mocov2_whisper_flamingo_amd import ops
dev = torch.device('cuda:0')
rows, C = 375000, 256
x = torch.randn(rows, C, device=dev).to(torch.bfloat16)
r = torch.randn(rows, C, device=dev).to(torch.bfloat16)
sc, sh = torch.randn(C, device=dev), torch.randn(C, device=dev)
GM = int(os.environ.get('PROBE_M', '12000'))
GK = int(os.environ.get('PROBE_K', '768'))
a = torch.randn(GM, GK, device=dev).to(torch.bfloat16)
b = torch.randn(3072, GK, device=dev).to(torch.bfloat16)
out = torch.empty(GM, 3072, device=dev, dtype=torch.bfloat16)
def bn(): ops.bn_apply(x, sc, sh, residual=r, relu=True, out=x)
def gm(): ops.gemm_nt(a, b, out=out)
gb, gf = rows * C * 2 * 3 / 1e9, 2.0 * GM * 3072 * GK / 1e12
def timeit(fn, stream, n=20):
    with torch.cuda.stream(stream):
        for _ in range(3): fn()
        stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(n): fn()
        e1.record(stream); stream.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
tg, tb = timeit(gm, s1), timeit(bn, s2)
n = 40
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
for _ in range(n):
    with torch.cuda.stream(s1): gm()
    with torch.cuda.stream(s2): bn()
torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
e1.record(); torch.cuda.synchronize()
pair = e0.elapsed_time(e1) / n * 1e-3
cfg = f"M={GM} K={GK} " + " ".join(f"{k[15:]}={v}" for k, v in os.environ.items() if k.startswith("AVSR_BN_APPLY_"))
print(f"[{cfg or 'default'}] alone: gemm {tg*1e6:6.1f} us ({gf/tg:5.0f} TF)  bn_apply {tb*1e6:6.1f} us ({gb/tb/1e3:4.2f} TB/s) | "
      f"side by side {pair*1e6:6.1f} us per pair (serial would be {(tg+tb)*1e6:6.1f})", flush=True)
