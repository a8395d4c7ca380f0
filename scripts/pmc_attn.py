import sys, torch
sys.path.insert(0, '.')
from mocov2_whisper_flamingo_amd import ops
dev = torch.device('cuda:0')
B, H, T = 8, 12, 1500
D = H * 64
qkv = torch.randn(B, T, 3 * D, device=dev).to(torch.bfloat16)
for _ in range(3):
    ops.attn_fwd(qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:], H, scale=0.125, use_mfma=True)
torch.cuda.synchronize()
