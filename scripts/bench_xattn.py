"""Times the masked / dropout MFMA attention core of the gated cross-attention (forward, backward) against the
fp32-math kernels, at the benchmark shape (B=8, T=75) and at shapes large enough to fill the chip."""
import sys, torch
sys.path.insert(0, '.')
from mocov2_whisper_flamingo_amd import ops
dev = torch.device('cuda:0')


def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


shapes = [(8, 8, 75, 75), (8, 8, 400, 400), (64, 8, 400, 400), (64, 8, 1500, 400)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in sys.argv[1].split(","))]
for B, H, Tq, Tk in shapes:
    D = H * 64
    q = torch.randn(B, Tq, D, device=dev).to(torch.bfloat16)
    kv = torch.randn(B, Tk, 2 * D, device=dev).to(torch.bfloat16)
    k, v = kv[:, :, :D], kv[:, :, D:]
    mask = torch.zeros(B, Tk, dtype=torch.uint8, device=dev)
    mask[:, Tk - Tk // 8:] = 1
    do = torch.randn(B, Tq, D, device=dev).to(torch.bfloat16)
    for p in (0.0, 0.1):
        for mfma in (True, False):
            if not mfma and B * Tq * Tk > 8 * 400 * 400:
                continue
            o, lse = ops.attn_fwd(q, k, v, H, key_mask=mask, scale=0.125, use_mfma=mfma, save_lse=True, p_drop=p, seed=7)
            tf = timeit(lambda: ops.attn_fwd(q, k, v, H, key_mask=mask, scale=0.125, use_mfma=mfma, save_lse=True, p_drop=p, seed=7))
            tb = timeit(lambda: ops.attn_bwd(q, k, v, o, do, lse, H, key_mask=mask, scale=0.125, use_mfma=mfma, p_drop=p, seed=7))
            fl = 4.0 * B * H * Tq * Tk * 64
            print(f"xattn B={B} H={H} Tq={Tq} Tk={Tk} p={p} {'mfma' if mfma else 'valu'}: fwd {tf*1e6:8.1f} us {fl/tf/1e12:6.1f} TF/s | "
                  f"bwd {tb*1e6:8.1f} us {3.5*fl/tb/1e12:6.1f} TF/s (7 GEMMs)", flush=True)
