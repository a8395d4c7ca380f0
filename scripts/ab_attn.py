"""Interleaved timing of attention-forward variants (scripts/variants/libavsr_*.so) in one process."""
import ctypes, glob, os, sys
import torch
sys.path.insert(0, '.')
from mocov2_whisper_flamingo_amd import _lib
dev = torch.device('cuda:0')
libs = {}
for f in sorted(glob.glob('scripts/variants/libavsr_*.so')):
    l = ctypes.CDLL(os.path.abspath(f))
    res, args = _lib.SIGNATURES["avsr_attn_fwd"]
    l.avsr_attn_fwd.restype, l.avsr_attn_fwd.argtypes = res, args
    libs[os.path.basename(f)[9:-3]] = l
stream = torch.cuda.current_stream().cuda_stream
for B, H, T in [(8, 12, 1500), (8, 8, 1500), (16, 16, 1500)]:
    D = H * 64
    qkv = torch.randn(B, T, 3 * D, device=dev).to(torch.bfloat16)
    o = torch.empty(B, T, D, device=dev, dtype=torch.bfloat16)
    q, k, v = qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
    def run(l):
        rc = l.avsr_attn_fwd(stream, 1, 1, q.data_ptr(), 3 * D, k.data_ptr(), 3 * D, v.data_ptr(), 3 * D, o.data_ptr(), D, None, None,
                             B, H, T, T, 64, 0.125, 0.0, 0, 0)
        assert rc == 0
    out = {}
    for rnd in range(5):
        for name, l in libs.items():
            for _ in range(2): run(l)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): run(l)
            e1.record(); torch.cuda.synchronize()
            out.setdefault(name, []).append(e0.elapsed_time(e1) / 10 * 1e3)
    print(f"B={B} H={H} T={T}: " + "  ".join(f"{n} {sorted(t)[2]:6.1f} us" for n, t in out.items()))
