"""Flamingo-style tanh-gated cross-attention fusion on the HIP kernels, with the
reference's classes, constructor signatures and state-dict keys
(models/gate_cross_attention.py:8-73).  These blocks are trainable: forward and
backward run through the autograd wrappers in ``functional``.

Everything operates on ``[B, T, d]`` GPU tensors of the compute dtype.  A query
row whose keys are all masked attends to nothing (its attention output is the
out-projection bias) instead of producing NaN.
"""
import torch
import torch.nn as nn

from . import functional as Fn


class LayerNorm(nn.LayerNorm):
    """LayerNorm computed in fp32 and cast back to the input dtype
    (models/gate_cross_attention.py:8-10)."""

    def forward(self, x, add=None):
        shp = x.shape
        y = Fn.LayerNormFn.apply(x.reshape(-1, shp[-1]), self.weight, self.bias, add)
        return y.view(shp)


class GatedCrossAttentionBlock(nn.Module):
    def __init__(self, d_model, n_heads, dropout=0.1, enable_logging=False):
        super().__init__()
        self.enable_logging = enable_logging
        # parameter container with nn.MultiheadAttention's names: in_proj_weight, in_proj_bias, out_proj.*
        self.attn = nn.MultiheadAttention(embed_dim=d_model, num_heads=n_heads, dropout=dropout, batch_first=True)
        self.attn_ln = LayerNorm(d_model)
        self.ff_ln = LayerNorm(d_model)
        self.attn_gate = nn.Parameter(torch.tensor([0.]))
        self.ff_gate = nn.Parameter(torch.tensor([0.]))
        n_mlp = d_model * 4
        self.ff = nn.Sequential(nn.Linear(d_model, n_mlp), nn.GELU(), nn.Linear(n_mlp, d_model), nn.Dropout(dropout))
        self.d_model, self.n_heads, self.p_drop = d_model, n_heads, dropout
        self._cache = None          # WeightCache, set by the owner
        self._seed_fn = lambda tag: 0
        self._seed_delta_fn = lambda: None      # device word added to the dropout seeds at run time (graph replays)
        # fused: the block is ONE autograd node over avsr_gated_xattn_fwd / _bwd (element-wise steps in the GEMM
        # epilogues, one host call per direction); False = the kernel-by-kernel composition, kept for comparison
        self.fused = True

    def _params(self):
        a, f = self.attn, self.ff
        return (a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias, f[0].weight, f[0].bias,
                f[2].weight, f[2].bias, self.attn_ln.weight, self.attn_ln.bias, self.ff_ln.weight, self.ff_ln.bias,
                self.attn_gate, self.ff_gate)

    def _forward_fused(self, x, xa, mask, shared=None):
        B, T, d = x.shape
        cache = self._cache
        p = self.p_drop if self.training else 0.0
        Tk = xa.shape[1] if xa is not None else 1
        prm = self._params()
        weights = [cache.get(w) for w in (prm[0], prm[2], prm[4], prm[6])]
        vectors = (prm[1], prm[3], prm[5], prm[7]) + prm[8:]
        blk = Fn.ops.xattn_block_desc(x.dtype, B, T, Tk, d, self.n_heads, self.ff[0].weight.shape[0], p,
                                      self._seed_fn("attn") if p > 0 else 0, self._seed_fn("ff") if p > 0 else 0,
                                      weights, [v.detach() for v in vectors],
                                      seed_delta=self._seed_delta_fn() if p > 0 else None)
        km = None
        if mask is not None and xa is not None:     # a bool mask is read as bytes where it lies (no copy kernel)
            km = mask.view(torch.uint8) if (mask.dtype == torch.bool and mask.is_contiguous()) else mask.to(torch.uint8).contiguous()
        x2 = x.reshape(B * T, d)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        xa2 = None
        if xa is not None:
            xa2 = xa.reshape(B * Tk, d)
            xa2 = xa2 if xa2.is_contiguous() else xa2.contiguous()
            if shared is None:
                shared = Fn.SharedGrad(1)
        out = Fn.GatedXattnBlockFn.apply(x2, xa2, km, blk, shared, *prm)
        return out.view(B, T, d)

    def forward(self, x, xa=None, mask=None, _shared=None):
        """x [B,T,d]; xa [B,Tk,d] or None (skips attention); mask bool/uint8 [B,Tk], True = padded key."""
        if xa is not None and xa.dim() == 4 and xa.size(1) == 1:
            xa = xa.squeeze(1)
        if mask is not None and mask.dim() == 3 and mask.size(1) == 1:
            mask = mask.squeeze(1)
        B, T, d = x.shape
        H = self.n_heads
        if d // H != 64:
            raise ValueError("attention kernels are built for head_dim 64")
        p = self.p_drop if self.training else 0.0
        cache = self._cache
        if cache is None or cache.dtype != x.dtype:        # stand-alone block: its own weight copies in x's dtype
            cache = self._cache = Fn.WeightCache(x.dtype)
        if self.fused:
            return self._forward_fused(x, xa, mask, _shared)
        x2 = x.reshape(B * T, d)
        if xa is not None:
            Tk = xa.shape[1]
            km = None if mask is None else mask.to(torch.uint8).contiguous()
            wc, wct = cache.get(self.attn.in_proj_weight)
            w, b = self.attn.in_proj_weight, self.attn.in_proj_bias
            h = self.attn_ln(x2)
            q = Fn.linear(h, w[:d], b[:d], (wc[:d], wct[:, :d]), sink=(w, b, (0, d), 2))
            kv = Fn.linear(xa.reshape(B * Tk, d), w[d:], b[d:], (wc[d:], wct[:, d:]), sink=(w, b, (d, 3 * d), 2))
            kv = kv.view(B, Tk, 2 * d)
            a = Fn.AttnFn.apply(q.view(B, T, d), kv[:, :, :d], kv[:, :, d:], km, H, 64 ** -0.5, p,
                                self._seed_fn("attn"))
            ao = Fn.linear(a.view(B * T, d), self.attn.out_proj.weight, self.attn.out_proj.bias,
                           cache.get(self.attn.out_proj.weight))
            x2 = Fn.GatedResidualFn.apply(x2, ao, self.attn_gate, 0.0, 0)
        h = self.ff_ln(x2)
        f = Fn.linear(h, self.ff[0].weight, self.ff[0].bias, cache.get(self.ff[0].weight))
        f = Fn.GeluFn.apply(f)
        f = Fn.linear(f, self.ff[2].weight, self.ff[2].bias, cache.get(self.ff[2].weight))
        x2 = Fn.GatedResidualFn.apply(x2, f, self.ff_gate, p, self._seed_fn("ff"))
        return x2.view(B, T, d)


class GatedCrossModalFusion(nn.Module):
    def __init__(self, d_model, n_heads, n_layers, dropout=0.1, enable_logging=False, dtype=torch.bfloat16):
        super().__init__()
        self.enable_logging = enable_logging
        self.audio_proj = nn.Linear(d_model, d_model)
        self.video_proj = nn.Linear(d_model, d_model)
        self.layers = nn.ModuleList([GatedCrossAttentionBlock(d_model, n_heads, dropout, enable_logging=enable_logging)
                                     for _ in range(n_layers)])
        self.ln_post = LayerNorm(d_model)
        self.set_runtime(Fn.WeightCache(dtype), lambda tag: 0)

    def set_runtime(self, cache, seed_fn, seed_delta_fn=None):
        """Share one weight cache / dropout-seed source with the owning AVNet."""
        self._cache = cache
        for i, layer in enumerate(self.layers):
            layer._cache = cache
            layer._seed_fn = (lambda tag, i=i: seed_fn(f"fusion.{i}.{tag}"))
            if seed_delta_fn is not None:
                layer._seed_delta_fn = seed_delta_fn

    def forward(self, audio_features, video_features, audio_mask=None, video_mask=None, _skip=None):
        """``audio_mask`` is accepted and unused, as in the reference
        (models/gate_cross_attention.py:60-70).  ``_skip`` (internal): a functional.SkipGrad whose slots "b" / "c"
        hold gradients that reach the audio / video features along another path; the two input projections add
        them inside their dX products."""
        B, T, d = audio_features.shape
        Tk = video_features.shape[1]
        cdt = self._cache.dtype
        a = audio_features if audio_features.dtype == cdt else audio_features.to(cdt)
        v = video_features if video_features.dtype == cdt else video_features.to(cdt)
        sa = (_skip, "b") if (_skip is not None and a is audio_features) else None
        sv = (_skip, "c") if (_skip is not None and v is video_features) else None
        x = Fn.linear(a.reshape(B * T, d), self.audio_proj.weight, self.audio_proj.bias,
                      self._cache.get(self.audio_proj.weight), skip=sa).view(B, T, d)
        xa = Fn.linear(v.reshape(B * Tk, d), self.video_proj.weight, self.video_proj.bias,
                       self._cache.get(self.video_proj.weight), skip=sv).view(B, Tk, d)
        # xa feeds every block: their backward kernels add into one gradient buffer (see functional.SharedGrad)
        shared = None
        if torch.is_grad_enabled() and xa.requires_grad and all(layer.fused for layer in self.layers):
            shared = Fn.SharedGrad(len(self.layers))
        for layer in self.layers:
            x = layer(x, xa, video_mask, _shared=shared)
        return self.ln_post(x)
