"""Attention forward time against the number of workgroups: shows how much of a launch is the half-empty last round."""
import sys, torch
sys.path.insert(0, '.')
from mocov2_whisper_flamingo_amd import ops
dev = torch.device('cuda:0')
H = 4
for pairs in (32, 48, 64, 80, 96, 112, 128, 160, 192, 256):
    for T in (1500,):
        B = pairs // H
        qkv = torch.randn(B, T, 3 * H * 64, device=dev).to(torch.bfloat16)
        D = H * 64
        ws = ops.AttnWorkspace()
        ts = []
        for w in (None, ws):
            f = lambda: ops.attn_fwd(qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:], H, scale=0.125, use_mfma=True, workspace=w)
            for _ in range(3): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e-3)
        wgs = pairs * ((T + 127) // 128)
        t, t2 = ts
        print(f"pairs={pairs:4d} T={T} wgs={wgs:5d} ({wgs/768:.2f} rounds of 768): {t*1e6:8.1f} us  {4.0*B*H*T*T*64/t/1e12:7.1f} TF/s"
              f" | with workspace {t2*1e6:8.1f} us  {4.0*B*H*T*T*64/t2/1e12:7.1f} TF/s ({'split' if ws._buf is not None else 'same'})", flush=True)
