"""Times the 3D stem kernels (direct fp32 vs MFMA bf16) on the benchmark shape."""
import sys, torch
sys.path.insert(0, '.')
from mocov2_whisper_flamingo_amd import ops
dev = torch.device('cuda:0')
B, T, H, W = 8, 75, 96, 96
video = torch.randn(B, T, 3, H, W, device=dev)
w = torch.randn(64, 3, 5, 3, 3, device=dev) * 0.1
wp = ops.stem_pack_weight(w); fr = ops.stem_pack_frags(wp)
cs = ops.stem_stats_slabs(video)
def timeit(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
fl = 2.0 * B * T * 2500 * 64 * 135
for name, kw in (("valu", {}), ("mfma", dict(frags=fr))):
    us = timeit(lambda: ops.stem3d_fwd(video, wp, torch.bfloat16, col_stats=(cs[0], cs[1]), **kw))
    print(f"stem {name}: {us:8.1f} us  {fl / us * 1e-6:6.1f} TFLOP/s (algorithmic)")
