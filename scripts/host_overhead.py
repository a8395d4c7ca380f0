"""How long does the host need to enqueue one training step (no synchronisation inside)?"""
import sys, time, torch
sys.path.insert(0, '.')
import bench
from mocov2_whisper_flamingo_amd import AVNet
from mocov2_whisper_flamingo_amd.dp import FlatTrainer
dev = torch.device('cuda:0')
model = AVNet("audiovisual", None, 96, bench.MODELARGS, bench.VOCAB, whisper_config="small", dtype=torch.bfloat16).to(dev)
model.train()
tr = FlatTrainer(model)
batch = bench.synthetic_batch(8, 75, 96, 96, 1, dev)
def step():
    tr.zero_grad()
    b = dict(batch); b["video_lengths"] = batch["video_lengths"].clone()
    loss, *_ = model.training_step(b)
    loss.backward(); tr.reduce_gradients(); tr.step()
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/5:.2f} ms/step, total {1e3*(t2-t0)/5:.2f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); step(); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(30)
