"""The visual frontend alone (batch 8 x 75 frames, train-mode BN, bf16).
  python scripts/visual_only.py [n]            replayed from a HIP graph n times (wall time per pass)
  python scripts/visual_only.py [n] --eager    launched kernel by kernel n times (for rocprofv3 --pmc passes)
  ... --dump-shapes FILE                       also writes the GEMM / conv shapes of one pass, in launch order"""
import json, sys, time, torch
sys.path.insert(0, '.')
import bench
from mocov2_whisper_flamingo_amd import MoCoVisualFrontend, ops
dev = torch.device('cuda:0')
args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if args and args[0].isdigit() else 20
eager = "--eager" in sys.argv
torch.manual_seed(1)
vm = MoCoVisualFrontend().to(dev).train()
batch = bench.synthetic_batch(8, 75, 96, 96, 1, dev)
def run(): return vm(batch["video"], batch["video_lengths"])
for _ in range(2): run()
torch.cuda.synchronize()
if "--dump-shapes" in sys.argv:
    ops.GEMM_PROFILE = []
    run(); torch.cuda.synchronize()
    shapes = [{"M": s[0], "N": s[1], "K": s[2], "conv": s[4], "us_by_events": round(a.elapsed_time(b) * 1e3, 1)}
              for a, b, fl, tag, s in ops.GEMM_PROFILE]
    ops.GEMM_PROFILE = None
    json.dump(shapes, open(sys.argv[sys.argv.index("--dump-shapes") + 1], "w"), indent=1)
if eager:
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): run()
    torch.cuda.synchronize()
    print(f"visual frontend, eager: {(time.perf_counter() - t0) / n * 1e3:.3f} ms (gram_bn3={vm.gram_bn3})")
    sys.exit(0)
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
