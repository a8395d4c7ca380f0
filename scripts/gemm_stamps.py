"""Phase timestamps (100 MHz) of workgroup 0 of the bf16 GEMM kernel, from the -DGEMM_STAMPS variant build."""
import ctypes, os, sys
import torch
sys.path.insert(0, '.')
from mocov2_whisper_flamingo_amd._lib import GemmDesc
dev = torch.device('cuda:0')
lib = ctypes.CDLL(os.path.abspath('scripts/variants/libavsr_stamps.so'))
lib.avsr_gemm_nt.restype = ctypes.c_int
lib.avsr_gemm_nt.argtypes = [ctypes.c_void_p, ctypes.POINTER(GemmDesc)]
lib.avsr_debug_read_stamps.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
for M, N, K in [(128, 128, 768), (12000, 2304, 768), (12000, 3072, 768), (12000, 768, 3072), (4096, 4096, 4096)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    for odt in (torch.bfloat16, torch.float32):
        out = torch.empty(M, N, device=dev, dtype=odt)
        d = GemmDesc()
        d.in_dtype, d.out_dtype = 1, (1 if odt == torch.bfloat16 else 0)
        d.M, d.N, d.K = M, N, K
        d.A, d.lda, d.B, d.ldb, d.C, d.ldc = a.data_ptr(), K, b.data_ptr(), K, out.data_ptr(), N
        d.batch, d.alpha = 1, 1.0
        res = []
        for it in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = lib.avsr_gemm_nt(None, ctypes.byref(d))
            e1.record()
            torch.cuda.synchronize()
            st = (ctypes.c_uint64 * 16)()
            lib.avsr_debug_read_stamps(st)
            res.append(([(st[i] - st[0]) * 10 for i in range(1, 8)] + ['cyc'] + [st[8 + i] - st[8] for i in range(1, 8)], e0.elapsed_time(e1) * 1e3))
        print(f"M={M} N={N} K={K} out={str(odt)[6:]}: ns since entry [setup, first tile landed, K loop done, epilogue issued, stores drained, 2nd epilogue issued, drained] "
              f"first call {res[0][0]} ({res[0][1]:.1f} us by events); later {res[-1][0]} ({res[-1][1]:.1f} us)", flush=True)
