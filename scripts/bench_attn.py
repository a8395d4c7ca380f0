import sys, torch
sys.path.insert(0, '.')
from mocov2_whisper_flamingo_amd import ops
dev = torch.device('cuda:0')
for B, H, T in [(8, 12, 1500), (8, 6, 1500), (8, 20, 1500)]:
    qkv = torch.randn(B, T, 3 * H * 64, device=dev).to(torch.bfloat16)
    D = H * 64
    for mode in (True, "fp8"):
        f = lambda: ops.attn_fwd(qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:], H, scale=0.125, use_mfma=mode)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20 * 1e-3
        print(f"attn {'fp8 ' if mode == 'fp8' else 'bf16'} B={B} H={H} T={T}: {t*1e6:8.1f} us  {4.0*B*H*T*T*64/t/1e12:7.1f} TF/s")
