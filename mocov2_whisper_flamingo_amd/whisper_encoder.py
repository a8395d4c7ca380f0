"""Whisper encoder on the HIP kernels, plus the parameter containers that keep
the reference's checkpoint key names (``whisper_model.encoder.*`` /
``whisper_model.decoder.*``; the reference builds them with
``WhisperModel.from_pretrained`` at models/av_net.py:22 and only ever runs the
encoder, models/av_net.py:78-81; the decoder lives in whisper_decoder.py).

The encoder is frozen in the reference (models/av_net.py:25-26), so there is
no backward here: ``WhisperEncoder.forward`` runs under ``no_grad``.
"""
import math
import os
from dataclasses import dataclass

import torch
import torch.nn as nn

from . import ops

WHISPER_SIZES = {
    # name: (d_model, layers, heads)  - ffn = 4*d_model, 80 mel bins, 1500 positions
    "tiny": (384, 4, 6),
    "base": (512, 6, 8),
    "small": (768, 12, 12),
    "medium": (1024, 24, 16),
    "large-v2": (1280, 32, 20),
}


@dataclass
class WhisperConfig:
    d_model: int = 768
    encoder_layers: int = 12
    encoder_attention_heads: int = 12
    decoder_layers: int = 12
    decoder_attention_heads: int = 12
    num_mel_bins: int = 80
    max_source_positions: int = 1500
    max_target_positions: int = 448
    vocab_size: int = 51865
    with_decoder: bool = True      # keep the (unused) decoder parameters for checkpoint compatibility

    @property
    def encoder_ffn_dim(self):
        return 4 * self.d_model

    @classmethod
    def from_name(cls, name, **kw):
        d, l, h = WHISPER_SIZES[name.replace("openai/whisper-", "")]
        return cls(d_model=d, encoder_layers=l, encoder_attention_heads=h, decoder_layers=l,
                   decoder_attention_heads=h, **kw)


def sinusoids(length, channels, max_timescale=10000.0):
    """Whisper's fixed encoder position table: [sin | cos] halves."""
    inc = math.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2))
    t = torch.arange(length).view(-1, 1) * inv.view(1, -1)
    return torch.cat([t.sin(), t.cos()], dim=1)


class _Attention(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.k_proj = nn.Linear(d, d, bias=False)
        self.v_proj = nn.Linear(d, d)
        self.q_proj = nn.Linear(d, d)
        self.out_proj = nn.Linear(d, d)


class _EncoderLayer(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.self_attn = _Attention(d)
        self.self_attn_layer_norm = nn.LayerNorm(d)
        self.fc1 = nn.Linear(d, 4 * d)
        self.fc2 = nn.Linear(4 * d, d)
        self.final_layer_norm = nn.LayerNorm(d)


class WhisperEncoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        d = cfg.d_model
        self.conv1 = nn.Conv1d(cfg.num_mel_bins, d, kernel_size=3, padding=1)
        self.conv2 = nn.Conv1d(d, d, kernel_size=3, stride=2, padding=1)
        self.embed_positions = nn.Embedding(cfg.max_source_positions, d)
        self.embed_positions.requires_grad_(False)
        with torch.no_grad():
            self.embed_positions.weight.copy_(sinusoids(cfg.max_source_positions, d))
        self.layers = nn.ModuleList([_EncoderLayer(d) for _ in range(cfg.encoder_layers)])
        self.layer_norm = nn.LayerNorm(d)
        self._packed = None
        # attn_split: let the attention forward cut the items of a half-empty last round of workgroups over the keys
        # (ops.attn_fwd, workspace=).  Faster when the encoder has the chip to itself (Whisper-small, batch 8: 92 ->
        # 87 us per layer); beside other streams, which fill those idle slots anyway, it is a small loss: off.
        self.attn_split = False
        self._attn_ws = ops.AttnWorkspace()     # the layers run one after another on one stream: one scratch for all
        self.attn_fp8 = False
        # ln_fold (bf16 compute dtype, d_model a multiple of 128): the LayerNorms between an out-projection / fc2 and
        # the QKV / fc1 product after it are not launched.  Between such products the fp32 residual stream lives as
        # two bf16 planes (hi = bf16(x), lo = bf16(x - hi): the same bytes, x to 2^-18): the producing product's
        # epilogue updates the planes in place and leaves the rows' per-tile {sum, sum of squares}; the consuming
        # product multiplies the hi plane - the un-normalised rows - by gain-folded weights and normalises in its
        # epilogue (ops.ln_fold_weights).  Per layer two launches and two passes over the stream less; rounding moves
        # from LayerNorm's output to its input.
        self.ln_fold = os.environ.get("AVSR_LN_FOLD", "1") != "0"

    # ---- weight packing -------------------------------------------------
    def _versions(self, dtype):
        return (dtype, sum(p._version for p in self.parameters()), next(self.parameters()).data_ptr(), self.ln_fold)

    def _pack(self, dtype):
        """One-time (per weight version) repacking into what the kernels read:
        conv weights as [out, tap*in], q/k/v fused into one [3D, D] matrix with the
        head_dim**-0.5 query scaling folded into W_q and b_q (0.125: exact)."""
        key = self._versions(dtype)
        if self._packed is not None and self._packed[0] == key:
            return self._packed[1]
        d = self.cfg.d_model
        hd = d // self.cfg.encoder_attention_heads
        if hd != 64:
            raise ValueError("attention kernels are built for head_dim 64")
        s = hd ** -0.5
        f32 = lambda t: t.detach().float().contiguous()
        pk = {
            "w1": self.conv1.weight.detach().permute(0, 2, 1).reshape(d, -1).to(dtype).contiguous(),
            "b1": f32(self.conv1.bias),
            "w2": self.conv2.weight.detach().permute(0, 2, 1).reshape(d, -1).to(dtype).contiguous(),
            "b2": f32(self.conv2.bias),
            "pos": f32(self.embed_positions.weight),
            "lnf": (f32(self.layer_norm.weight), f32(self.layer_norm.bias)),
            "layers": [],
        }
        for L in self.layers:
            a = L.self_attn
            wqkv = torch.cat([a.q_proj.weight.detach() * s, a.k_proj.weight.detach(), a.v_proj.weight.detach()], 0)
            bqkv = torch.cat([a.q_proj.bias.detach() * s, torch.zeros_like(a.q_proj.bias), a.v_proj.bias.detach()], 0)
            pk["layers"].append({
                "ln1": (f32(L.self_attn_layer_norm.weight), f32(L.self_attn_layer_norm.bias)),
                "wqkv": wqkv.to(dtype).contiguous(), "bqkv": f32(bqkv),
                "wo": a.out_proj.weight.detach().to(dtype).contiguous(), "bo": f32(a.out_proj.bias),
                "ln2": (f32(L.final_layer_norm.weight), f32(L.final_layer_norm.bias)),
                "wfc1": L.fc1.weight.detach().to(dtype).contiguous(), "bfc1": f32(L.fc1.bias),
                "wfc2": L.fc2.weight.detach().to(dtype).contiguous(), "bfc2": f32(L.fc2.bias),
            })
            if self.ln_fold and dtype == torch.bfloat16 and d % 128 == 0 and d <= 1280:
                lw = pk["layers"][-1]
                lw["qkv_ln"] = ops.ln_fold_weights(wqkv, bqkv, *lw["ln1"])
                lw["fc1_ln"] = ops.ln_fold_weights(L.fc1.weight, L.fc1.bias, *lw["ln2"])
        self._packed = (key, pk)
        return pk

    # ---- forward ----------------------------------------------------------
    @torch.no_grad()
    def forward(self, input_features, attention_mask=None, dtype=torch.bfloat16, channels_last=False, attn_fp8=None,
                rows=None):
        """input_features: [B, n_mels, 3000] (the HF layout) or, with
        ``channels_last``, [B, 3000, n_mels].  ``attention_mask`` is accepted and
        ignored exactly as the HF encoder does (hf:607-609).  Returns a 1-tuple
        ``([B, 1500, D] in the compute dtype,)`` so that ``encoder(x)[0]`` works.
        ``attn_fp8`` (default: the module's ``attn_fp8`` attribute): QK^T and PV of the self-attention
        on fp8 e4m3 MFMA (bf16 compute dtype only); everything else stays bf16.
        ``rows`` = n < 1500: the caller reads only the first n frames of every clip (AVNet truncates the audio
        features to the video length, models/av_net.py:91-96) and gets ``[B, n, D]``, bit for bit the first n
        frames of the full result: every layer but the last runs on all 1500 frames (its keys and values feed
        the kept frames' attention); in the last layer only keys and values are computed for all frames, while
        queries, attention output, out-projection, MLP and the final LayerNorm - all row-wise - run on the n
        kept frames."""
        cfg = self.cfg
        x = input_features
        if not channels_last:
            x = x.transpose(1, 2)
        B, T, C = x.shape
        expected = cfg.max_source_positions * 2
        if T != expected or C != cfg.num_mel_bins:
            raise ValueError(
                f"Whisper expects the mel input features to be of length {expected}, but found {T}. "
                f"Make sure to pad the input mel features to {expected}.")
        pk = self._pack(dtype)
        H, D = cfg.encoder_attention_heads, cfg.d_model
        # The two stem convolutions (kernel 3, padding 1; hf:570-571) as plain GEMMs: with one zero frame stored
        # before and after each clip, the three taps of output frame t are ONE contiguous run of 3*C values
        # starting at stored frame t (stride 1) or 2t (stride 2) - an ordinary row-major operand whose row stride
        # is C (or 2C) and whose rows overlap, batched over the clips.  No per-pixel address arithmetic.
        Tn = cfg.max_source_positions
        xp = torch.empty((B, T + 2, C), dtype=dtype, device=x.device)
        h1p = torch.empty((B, T + 2, D), dtype=dtype, device=x.device)
        ops.zero_regions(xp[:, 0], xp[:, T + 1], h1p[:, 0])          # the zero frames around each clip, one launch
        # [B, C, T] storage (the HF layout) is transposed to channels-last first
        xc = (x if x.is_contiguous() else x.contiguous()) if channels_last else ops.transpose2d(input_features, dtype=dtype)
        ops.cast_rows(xc.view(B, T * C), xp.view(B, (T + 2) * C)[:, C:C + T * C])
        a1 = torch.as_strided(xp, (B, T, 3 * C), ((T + 2) * C, C, 1))
        ops.gemm_nt(a1, pk["w1"], bias=pk["b1"], act=ops.ACT_GELU, out=h1p[:, 1:T + 1])
        # gelu(conv2) + positions, written straight into the fp32 residual stream
        a2 = torch.as_strided(h1p, (B, Tn, 3 * D), ((T + 2) * D, 2 * D, 1))
        xs = torch.empty((B, Tn, D), dtype=torch.float32, device=x.device)
        ops.gemm_nt(a2, pk["w2"], bias=pk["b2"], act=ops.ACT_GELU, residual=pk["pos"], res_period=Tn, out=xs)
        xs = xs.view(B * Tn, D)
        use_mfma = dtype == torch.bfloat16
        if (self.attn_fp8 if attn_fp8 is None else attn_fp8):
            if not use_mfma:
                raise ValueError("fp8 attention needs the bf16 compute dtype")
            use_mfma = "fp8"
        n_keep = Tn if rows is None else max(1, min(int(rows), Tn))
        fold = "qkv_ln" in pk["layers"][0] if pk["layers"] else False
        if fold:
            eps = self.layers[0].self_attn_layer_norm.eps
            xb = torch.empty((2, B * Tn, D), dtype=dtype, device=x.device)         # the stream's rows as hi / lo planes
            planes = (xb[0], xb[1])
            lnp = torch.empty((ops.row_lse_parts(D), B * Tn, 2), dtype=torch.float32, device=x.device)
        have = False                    # the stream is in the planes (and lnp holds its row statistics), not in xs
        last = len(pk["layers"]) - 1
        for li, lw in enumerate(pk["layers"]):
            pruned = n_keep < Tn and li == last
            # (the last layer is never folded: it runs on the kept rows only when the caller reads a prefix of the frames,
            # and ``rows=n`` promises the bits of the full result)
            if fold and li < last:
                if have:
                    w_g, colsum, bias_b = lw["qkv_ln"]
                    qkv = ops.gemm_nt(xb[0], w_g, bias=bias_b, ln_in=(lnp, colsum, eps)).view(B, Tn, 3 * D)
                else:
                    h = ops.layernorm_fwd(xs, *lw["ln1"], out_dtype=dtype)
                    qkv = ops.gemm_nt(h, lw["wqkv"], bias=lw["bqkv"]).view(B, Tn, 3 * D)
                o = ops.attn_fwd(qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:], H, scale=1.0, use_mfma=use_mfma,
                                 workspace=self._attn_ws if self.attn_split else None)
                ops.gemm_nt(o.view(B * Tn, D), lw["wo"], bias=lw["bo"], no_out=True, planes_out=planes, ln_out=(lnp, None),
                            **(dict(planes_res=planes) if have else dict(residual=xs)))
                w_g, colsum, bias_b = lw["fc1_ln"]
                f = ops.gemm_nt(xb[0], w_g, bias=bias_b, act=ops.ACT_GELU, ln_in=(lnp, colsum, eps))
                # the stream stays in the planes if its next reader is the next layer's folded QKV product; before the
                # last layer it goes back to fp32
                have = li + 1 < last
                if have:
                    ops.gemm_nt(f, lw["wfc2"], bias=lw["bfc2"], no_out=True, planes_res=planes, planes_out=planes,
                                ln_out=(lnp, None))
                else:
                    ops.gemm_nt(f, lw["wfc2"], bias=lw["bfc2"], planes_res=planes, out=xs)
                continue
            h = ops.layernorm_fwd(xs, *lw["ln1"], out_dtype=dtype)
            if pruned:
                kv = ops.gemm_nt(h, lw["wqkv"][D:], bias=lw["bqkv"][D:]).view(B, Tn, 2 * D)
                q = ops.gemm_nt(h.view(B, Tn, D)[:, :n_keep], lw["wqkv"][:D], bias=lw["bqkv"][:D])      # [B, n, D]
                o = ops.attn_fwd(q, kv[:, :, :D], kv[:, :, D:], H, scale=1.0, use_mfma=use_mfma,
                                 workspace=self._attn_ws if self.attn_split else None)
                xk = torch.empty((B, n_keep * D), dtype=torch.float32, device=x.device)                 # kept rows of the stream
                ops.cast_rows(xs.view(B, Tn * D)[:, :n_keep * D], xk)
                xs = xk.view(B * n_keep, D)
                ops.gemm_nt(o.view(B * n_keep, D), lw["wo"], bias=lw["bo"], residual=xs, out=xs)
                h = ops.layernorm_fwd(xs, *lw["ln2"], out_dtype=dtype)
                f = ops.gemm_nt(h, lw["wfc1"], bias=lw["bfc1"], act=ops.ACT_GELU)
                ops.gemm_nt(f, lw["wfc2"], bias=lw["bfc2"], residual=xs, out=xs)
                Tn = n_keep
                break
            qkv = ops.gemm_nt(h, lw["wqkv"], bias=lw["bqkv"]).view(B, Tn, 3 * D)
            o = ops.attn_fwd(qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:], H, scale=1.0, use_mfma=use_mfma,
                             workspace=self._attn_ws if self.attn_split else None)
            ops.gemm_nt(o.view(B * Tn, D), lw["wo"], bias=lw["bo"], residual=xs, out=xs)
            h = ops.layernorm_fwd(xs, *lw["ln2"], out_dtype=dtype)
            f = ops.gemm_nt(h, lw["wfc1"], bias=lw["bfc1"], act=ops.ACT_GELU)
            ops.gemm_nt(f, lw["wfc2"], bias=lw["bfc2"], residual=xs, out=xs)
        out = ops.layernorm_fwd(xs, *pk["lnf"], out_dtype=dtype)
        return (out.view(B, Tn, D),)


class WhisperModel(nn.Module):
    """Container with the HF attribute names the reference touches:
    ``.encoder``, ``.decoder`` (constructed by the reference, never called by it) and ``.config.d_model``."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.encoder = WhisperEncoder(config)
        if config.with_decoder:
            from .whisper_decoder import WhisperDecoder
            self.decoder = WhisperDecoder(config)
