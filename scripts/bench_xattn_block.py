"""Times one gated cross-attention block (forward, and forward + backward) through avsr_gated_xattn_fwd / _bwd and
through the kernel-by-kernel composition, at the benchmark shape (B=8, T=75), at the reference's max_frames
(B=8, T=400: 3200 rows) and at shapes that fill the chip.  Prints achieved TFLOP/s against the 2.5 PFLOP/s bf16 peak.
Algorithmic FLOPs per block forward: rows*(2*d*d [q] + 2*d*d [out] + 4*d*d_ff [ffn]) + key_rows*4*d*d [k,v]
+ 4*B*H*Tq*Tk*64 [attention]; backward = 2x the GEMM part + 2.5x the attention part.
usage: python scripts/bench_xattn_block.py [B,T[,fused|plain]] [--once]"""
import sys
import torch
sys.path.insert(0, '.')
from mocov2_whisper_flamingo_amd import GatedCrossAttentionBlock
from mocov2_whisper_flamingo_amd import functional as Fn

dev = torch.device('cuda:0')
d, H = 512, 8
once = "--once" in sys.argv
args = [a for a in sys.argv[1:] if not a.startswith("--")]
shapes = [(8, 75), (8, 400), (32, 400), (64, 400)]
modes = ["fused", "plain"]
if args:
    parts = args[0].split(",")
    shapes = [(int(parts[0]), int(parts[1]))]
    if len(parts) > 2:
        modes = [parts[2]]


def flops(B, T):
    rows = B * T
    gemm = rows * (2 * d * d + 2 * d * d + 4 * d * 4 * d) + rows * 4 * d * d
    attn = 4 * B * H * T * T * 64
    return gemm + attn, 2 * gemm + 2.5 * attn


def timeit(f, n):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for B, T in shapes:
    blk = GatedCrossAttentionBlock(d, H, dropout=0.1).to(dev).train()
    with torch.no_grad():
        blk.attn_gate.fill_(0.5)
        blk.ff_gate.fill_(-0.4)
    blk._cache = Fn.WeightCache(torch.bfloat16)
    blk._seed_fn = lambda tag: 99
    for p in blk.parameters():                       # gradients accumulate straight into fp32 .grad buffers
        p.grad = torch.zeros_like(p)
        p._avsr_direct_grad = True
    x = torch.randn(B, T, d, device=dev).to(torch.bfloat16).requires_grad_(True)
    xa = torch.randn(B, T, d, device=dev).to(torch.bfloat16).requires_grad_(True)
    mask = torch.zeros(B, T, dtype=torch.bool, device=dev)
    mask[:, T - T // 8:] = True
    dout = torch.randn(B, T, d, device=dev).to(torch.bfloat16)
    ff, fb = flops(B, T)
    for mode in modes:
        blk.fused = mode == "fused"

        def fwd():
            with torch.no_grad():
                return blk(x, xa, mask)

        def fwdbwd():
            y = blk(x, xa, mask)
            y.backward(dout)
            x.grad = None
            xa.grad = None

        if once:                                     # one forward + backward, for a profiler
            fwdbwd()
            torch.cuda.synchronize()
            continue
        n = 50 if B * T <= 4000 else 10
        tf = timeit(fwd, n)
        tb = timeit(fwdbwd, n) - tf
        print(f"block B={B} T={T} rows={B*T} {mode:5s}: fwd {tf*1e6:8.1f} us {ff/tf/1e12:7.1f} TF/s = {ff/tf/2.5e15*100:5.1f} % of peak | "
              f"bwd {tb*1e6:8.1f} us {fb/tb/1e12:7.1f} TF/s = {fb/tb/2.5e15*100:5.1f} %", flush=True)
