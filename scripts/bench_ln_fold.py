"""Times the encoder's four products with and without the LayerNorm fold (and the LayerNorm kernel itself)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mocov2_whisper_flamingo_amd import ops

dev = torch.device("cuda:0")
M, D = 12000, 768
g = torch.Generator().manual_seed(1)
bf = torch.bfloat16


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


xs = torch.randn(M, D, generator=g).to(dev)
o = torch.randn(M, D, generator=g).to(dev).to(bf)
f = torch.randn(M, 4 * D, generator=g).to(dev).to(bf)
wo = (torch.randn(D, D, generator=g) * 0.03).to(dev).to(bf)
w2 = (torch.randn(D, 4 * D, generator=g) * 0.02).to(dev).to(bf)
bo = torch.zeros(D, device=dev)
gamma, beta = torch.ones(D, device=dev), torch.zeros(D, device=dev)
parts = torch.zeros(ops.row_lse_parts(D), M, 2, device=dev)
xb = torch.zeros(M, D, dtype=bf, device=dev)
print("layernorm              %.1f us" % timeit(lambda: ops.layernorm_fwd(xs, gamma, beta, out_dtype=bf)))
print("out-proj plain         %.1f us" % timeit(lambda: ops.gemm_nt(o, wo, bias=bo, residual=xs, out=xs)))
print("out-proj ln_out        %.1f us" % timeit(lambda: ops.gemm_nt(o, wo, bias=bo, residual=xs, out=xs, ln_out=(parts, xb))))
print("fc2 plain              %.1f us" % timeit(lambda: ops.gemm_nt(f, w2, bias=bo, residual=xs, out=xs)))
print("fc2 ln_out             %.1f us" % timeit(lambda: ops.gemm_nt(f, w2, bias=bo, residual=xs, out=xs, ln_out=(parts, xb))))
pl = torch.zeros(2, M, D, dtype=bf, device=dev)
P = (pl[0], pl[1])
print("out-proj f32 -> planes %.1f us" % timeit(lambda: ops.gemm_nt(o, wo, bias=bo, residual=xs, no_out=True, planes_out=P, ln_out=(parts, None))))
print("out-proj planes        %.1f us" % timeit(lambda: ops.gemm_nt(o, wo, bias=bo, no_out=True, planes_res=P, planes_out=P, ln_out=(parts, None))))
print("fc2 planes             %.1f us" % timeit(lambda: ops.gemm_nt(f, w2, bias=bo, no_out=True, planes_res=P, planes_out=P, ln_out=(parts, None))))
print("fc2 planes -> f32      %.1f us" % timeit(lambda: ops.gemm_nt(f, w2, bias=bo, planes_res=P, out=xs)))
xs.normal_()
ops.gemm_nt(o, wo, bias=bo, residual=xs, out=xs, ln_out=(parts, xb))
for name, N, act in (("qkv", 3 * D, ops.ACT_NONE), ("fc1", 4 * D, ops.ACT_GELU)):
    w = (torch.randn(N, D, generator=g) * 0.03).to(dev)
    b = torch.zeros(N, device=dev)
    w_g, colsum, bias_b = ops.ln_fold_weights(w, b, gamma, beta)
    wb = w.to(bf)
    out = torch.empty(M, N, dtype=bf, device=dev)
    print("%s plain              %.1f us" % (name, timeit(lambda: ops.gemm_nt(xb, wb, bias=b, act=act, out=out))))
    print("%s ln_in              %.1f us" % (name, timeit(lambda: ops.gemm_nt(xb, w_g, bias=bias_b, act=act, out=out, ln_in=(parts, colsum, 1e-5)))))
