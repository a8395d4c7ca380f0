import sys, torch
sys.path.insert(0, '.')
from mocov2_whisper_flamingo_amd import ops
dev = torch.device('cuda:0')
x = torch.randn(600, 4, 4, 2048, device=dev).to(torch.bfloat16)
lens = torch.full((8,), 75, device=dev)
f = lambda: ops.avgpool_mask(x, lens, 8, 75, torch.bfloat16)
for _ in range(3): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): f()
e1.record(); torch.cuda.synchronize()
print(f"avgpool {e0.elapsed_time(e1) / 50 * 1e3:.1f} us")
