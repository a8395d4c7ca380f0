"""Where one benchmark step spends its time, from HIP events at phase boundaries (no profiler attached)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mocov2_whisper_flamingo_amd import AVNet
from mocov2_whisper_flamingo_amd.dp import FlatTrainer

dev = torch.device("cuda:0")
torch.manual_seed(1234)
model = AVNet("audiovisual", None, 96, bench.MODELARGS, bench.VOCAB, whisper_config="small", dtype=torch.bfloat16, seed=1234)
with torch.no_grad():
    for blk in model.fusion_module.layers:
        blk.attn_gate.fill_(0.5); blk.ff_gate.fill_(0.5)
model.to(dev).train()
tr = FlatTrainer(model, lr=1e-3)
batch = bench.synthetic_batch(8, 75, 96, 96, 1234, dev)
marks = {}
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.setdefault(name, []).append(e)
def wrap(mod, name):
    f = mod.forward
    def g(*a, **k):
        out = f(*a, **k); mark(name); return out
    mod.forward = g
wrap(model.whisper_model.encoder, "encoder_done(main)")
wrap(model.visual_model, "visual_done(side)")
def step():
    mark("start")
    tr.zero_grad()
    b = dict(batch); b["video_lengths"] = batch["video_lengths"].clone()
    loss, *_ = model.training_step(b)
    mark("forward_done")
    loss.backward()
    mark("backward_done(main)")
    tr.reduce_gradients()
    mark("grads_joined")
    tr.step()
    mark("end")
for _ in range(5): step()
torch.cuda.synchronize()
names = ["start", "visual_done(side)", "encoder_done(main)", "forward_done", "backward_done(main)", "grads_joined", "end"]
for it in (-2, -1):
    s = marks["start"][it]
    print("  ".join(f"{n}={s.elapsed_time(marks[n][it]):.2f}" for n in names[1:]))
