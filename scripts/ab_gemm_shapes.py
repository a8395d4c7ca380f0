"""Interleaved timing of dense GEMM variants (scripts/variants/libavsr_*.so) on the Whisper encoder's products."""
import ctypes, glob, os, subprocess, sys
import torch
res = {}
for f in sorted(glob.glob('scripts/variants/libavsr_*.so')):
    out = subprocess.run([sys.executable, 'scripts/bench_gemm.py'], env=dict(os.environ, AVSR_HIP_LIB=os.path.abspath(f)),
                         capture_output=True, text=True).stdout
    print(os.path.basename(f))
    print("\n".join(l for l in out.splitlines() if l.startswith("gemm")))
