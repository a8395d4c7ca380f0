"""What a kernel family really costs in the pipelined step: runs bench.py's loop with one family of launches
replaced by a no-op that returns a preallocated buffer (results are wrong on purpose; only the step time is read).
Standalone kernel times do not add up in this step - two encoder streams and the tail share the chip - so the marginal
cost of a family is measured by removing it.
usage: python scripts/ablate_step.py <family> [bench args]   family: none | attn | ln | bn_apply | maxpool | gemm_out |
       gemm_fc2 | gemm_qkv | gemm_fc1 | conv3x3 | conv1x1 | stem | tail | bn_finalize"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mocov2_whisper_flamingo_amd import ops

family = sys.argv[1]
sys.argv = [sys.argv[0]] + sys.argv[2:] + ["--no-cpu-baseline", "--no-roofline"]
_cache = {}


def cached(key, make):
    if key not in _cache:
        _cache[key] = make()
    return _cache[key]


orig = {n: getattr(ops, n) for n in ("attn_fwd", "layernorm_fwd", "bn_apply", "maxpool3x3s2", "gemm_nt", "stem3d_fwd")}

if family == "attn":
    def attn_fwd(q, k, v, H, **kw):
        if q.shape[1] < 1000:
            return orig["attn_fwd"](q, k, v, H, **kw)
        return cached(("attn", tuple(q.shape)), lambda: torch.zeros(q.shape[0], q.shape[1], H * 64, dtype=q.dtype, device=q.device))
    ops.attn_fwd = attn_fwd
elif family == "ln":
    def layernorm_fwd(x, gamma, beta, *, out_dtype=None, add=None, save_stats=False, eps=1e-5):
        if x.shape[0] < 5000 or save_stats:
            return orig["layernorm_fwd"](x, gamma, beta, out_dtype=out_dtype, add=add, save_stats=save_stats, eps=eps)
        return cached(("ln", tuple(x.shape), out_dtype), lambda: torch.zeros(x.shape, dtype=out_dtype or x.dtype, device=x.device))
    ops.layernorm_fwd = layernorm_fwd
elif family == "bn_apply":
    def bn_apply(x, scale, shift, *, residual=None, residual_affine=None, relu=True, out=None, out_dtype=None):
        return out if out is not None else x
    ops.bn_apply = bn_apply
elif family == "bn_finalize":
    def bn_finalize(*a, **k):
        return None
    ops.bn_finalize = bn_finalize
elif family == "maxpool":
    def maxpool(x, *, scale=None, shift=None, out_dtype=None):
        N, H, W, C = x.shape
        return cached(("mp", tuple(x.shape)), lambda: torch.zeros(N, (H + 1) // 2, (W + 1) // 2, C, dtype=out_dtype or x.dtype, device=x.device))
    ops.maxpool3x3s2 = maxpool
elif family == "stem":
    def stem(video, weight, out_dtype, **kw):
        B, T, _, H, W = video.shape
        return cached(("stem",), lambda: torch.zeros(B * T, (H + 6 - 3) // 2 + 1, (W + 6 - 3) // 2 + 1, 64, dtype=out_dtype, device=video.device))
    ops.stem3d_fwd = stem
elif family.startswith("gemm_") or family.startswith("conv"):
    def gemm_nt(a, b, **kw):
        conv = kw.get("conv")
        N, K = b.shape[-2], b.shape[-1]
        M = a.shape[-2] if conv is None else 0
        hit = False
        if conv is None and a.dim() == 2 and M >= 10000:
            hit = {"gemm_out": (N, K) == (768, 768), "gemm_fc2": (N, K) == (768, 3072), "gemm_qkv": N == 2304,
                   "gemm_fc1": N == 3072}.get(family, False)
        if conv is not None:
            hit = (family == "conv3x3" and conv[3] == 3) or (family == "conv1x1" and conv[3] == 1)
        if not hit or kw.get("stats_only"):
            return orig["gemm_nt"](a, b, **kw)
        if kw.get("out") is not None:
            return kw["out"]
        Mv = M if conv is None else a.numel() // (conv[0] * conv[1] * conv[2]) * conv[8] * conv[9]
        odt = kw.get("out_dtype") or a.dtype
        return cached((family, Mv, N, odt), lambda: torch.zeros(Mv, N, dtype=odt, device=a.device))
    ops.gemm_nt = gemm_nt
elif family == "tail":
    os.environ["AVSR_ABLATE_TAIL"] = "1"

import bench
if family == "tail":
    from mocov2_whisper_flamingo_amd import av_net
    # encoders only: the training step returns a constant loss that needs no tail
    def training_step(self, batch, label_smoothing=0.1, encoded=None):
        z = cached(("loss",), lambda: torch.zeros((), device=batch["audio"].device, requires_grad=True))
        for stream, ev in encoded["done"]:
            torch.cuda.current_stream().wait_event(ev)
        return z * 1.0, z, z, None
    av_net.AVNet.training_step = training_step
print("ablating:", family, file=sys.stderr)
bench.main()
