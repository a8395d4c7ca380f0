"""The visual frontend alone (batch 8 x 75 frames, train-mode BN, bf16): replayed from a HIP graph N times.
Usage: python scripts/visual_only.py [n]   - under rocprofv3 --kernel-trace --stats this gives the per-kernel table."""
import sys, time, torch
sys.path.insert(0, '.')
import bench
from mocov2_whisper_flamingo_amd import MoCoVisualFrontend
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
torch.manual_seed(1)
vm = MoCoVisualFrontend().to(dev).train()
batch = bench.synthetic_batch(8, 75, 96, 96, 1, dev)
def run(): return vm(batch["video"], batch["video_lengths"])
for _ in range(2): run()
torch.cuda.synchronize()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        out = run()
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize()
    print(f"visual frontend, graph replay: {(time.perf_counter() - t0) / n * 1e3:.3f} ms (gram_bn3={vm.gram_bn3})")
