"""Times avsr_logmel_fwd on 8 x 30 s clips with HIP events (fp32 MFMA contraction: 1.06 GFLOP per clip)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mocov2_whisper_flamingo_amd import LogMelFrontend

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
wave = 0.1 * torch.randn(B, 480000, device=dev)
for variant in ("whisper", "melspec_norm"):
    fe = LogMelFrontend(variant, dev)
    for kw in (dict(), dict(dtype=torch.bfloat16, channels_last=True)):
        for _ in range(3):
            fe(wave, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fe(wave, **kw)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / 20
        flop = B * (3000 * 416 * 400 * 2 + 3000 * 208 * 80 * 2)
        print(f"{variant:13s} {kw} B={B}: {us:8.1f} us/call  {flop / us * 1e-6:6.1f} TFLOP/s (fp32 MFMA)  "
              f"{B / us * 1e6:9.0f} clips/s", flush=True)
