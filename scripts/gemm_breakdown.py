"""Per-shape table of the GEMM / implicit-GEMM-conv launches of one benchmark step (HIP-event timed)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

from mocov2_whisper_flamingo_amd import AVNet, ops
from mocov2_whisper_flamingo_amd.dp import FlatTrainer

dev = torch.device("cuda:0")
torch.manual_seed(1234)
model = AVNet("audiovisual", None, 96, bench.MODELARGS, bench.VOCAB, whisper_config="small", dtype=torch.bfloat16, seed=1234)
with torch.no_grad():
    for blk in model.fusion_module.layers:
        blk.attn_gate.fill_(0.5); blk.ff_gate.fill_(0.5)
model.to(dev).train()
trainer = FlatTrainer(model, lr=1e-3, bucket_mb=64)
batch = bench.synthetic_batch(8, 75, 96, 96, 1234, dev)


def step():
    trainer.zero_grad()
    b = dict(batch); b["video_lengths"] = batch["video_lengths"].clone()
    loss, _, _, _ = model.training_step(b)
    loss.backward(); trainer.reduce_gradients(); trainer.step()


for _ in range(3):
    step()
# per-launch durations of the kernel alone: nothing on a second stream sharing the CUs
from mocov2_whisper_flamingo_amd import functional as Fn
model.overlap_encoders = False
Fn.GRAD_STREAM = None
step()
ops.GEMM_PROFILE = []
ops.GEMM_PROFILE_REPEAT = '--repeat' in sys.argv
step()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for a, b, fl, tag, shape, *rep in ops.GEMM_PROFILE:
    t = a.elapsed_time(b) * 1e-3
    e = agg.setdefault((tag,) + shape, [0, 0.0, 0.0, 0.0])
    e[0] += 1; e[1] += t; e[2] += fl
    if rep: e[3] += rep[0].elapsed_time(rep[1]) * 1e-3
tot = sum(e[1] for e in agg.values())
print(f"total GEMM time {tot*1e3:.3f} ms in {sum(e[0] for e in agg.values())} launches")
for k, e in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tag, M, N, K, batch_, conv = k
    rep = f" | repeat {e[3]*1e6:8.1f} us {e[2]/e[3]/1e12:7.1f} TF" if e[3] else ""
    print(f"{e[1]*1e6:8.1f} us {e[1]/tot*100:5.1f}%  x{e[0]:3d}  {e[2]/e[1]/1e12:7.1f} TF{rep}  {tag} M={M} N={N} K={K} b={batch_} conv={conv}")
