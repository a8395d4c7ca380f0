"""The visual frontend alone: kernel-sum vs eager wall time vs HIP-graph replay wall time (dependent-launch gaps)."""
import sys, time, torch
sys.path.insert(0, '.')
import bench
from mocov2_whisper_flamingo_amd import AVNet
dev = torch.device('cuda:0')
model = AVNet("audiovisual", None, 96, bench.MODELARGS, bench.VOCAB, whisper_config="small", dtype=torch.bfloat16).to(dev)
model.train()
batch = bench.synthetic_batch(8, 75, 96, 96, 1, dev)
vm = model.visual_model
def run(): return vm(batch["video"], batch["video_lengths"])
for _ in range(3): run()
torch.cuda.synchronize()
def wall(f, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print(f"visual eager: {wall(run):.3f} ms")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        out = run()
    print(f"visual graph: {wall(g.replay):.3f} ms")
    enc = model.whisper_model.encoder
    def runa(): return enc(batch["audio"], dtype=torch.bfloat16, channels_last=True)
    for _ in range(2): runa()
    print(f"audio eager: {wall(runa):.3f} ms")
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=s):
        o2 = runa()
    print(f"audio graph: {wall(g2.replay):.3f} ms")
    e = model.encode(model.batch_inputs(batch), ahead=True, graph=True)
    def both(): return model.encode(model.batch_inputs(batch), ahead=True, graph=True)
    print(f"both (encode graph): {wall(both):.3f} ms")
