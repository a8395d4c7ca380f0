"""What a microsecond of kernel time on each encoder stream costs the pipelined step, measured on REAL data (a kernel
family replaced by zeros changes the data the following MFMA kernels multiply, and with it their power and clock):
the benchmark loop with the audio encoder cut to its first N layers, or with identity-shortcut blocks of the visual
trunk skipped (the tensors keep their shapes, so everything downstream runs unchanged).
usage: python scripts/marginal_probe.py audio=<layers> | visual=<i,j,...> | none  [bench args]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
what = sys.argv[1]
sys.argv = [sys.argv[0]] + sys.argv[2:] + ["--no-cpu-baseline", "--no-roofline", "--no-roofline-extra"]
from mocov2_whisper_flamingo_amd import whisper_encoder, moco_visual_frontend

if what.startswith("audio="):
    keep = int(what.split("=")[1])
    orig_pack = whisper_encoder.WhisperEncoder._pack

    def _pack(self, dtype):
        pk = dict(orig_pack(self, dtype))
        pk["layers"] = pk["layers"][:keep]
        return pk
    whisper_encoder.WhisperEncoder._pack = _pack
elif what.startswith("visual="):
    skip = {int(i) for i in what.split("=")[1].split(",")}
    orig_block = moco_visual_frontend.MoCoVisualFrontend.run_block

    def run_block(self, index, y, pk=None, pool=None):
        if index in skip:
            return y
        return orig_block(self, index, y, pk, pool)
    moco_visual_frontend.MoCoVisualFrontend.run_block = run_block
import bench
bench.main()
