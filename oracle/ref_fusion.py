"""Oracle (test infrastructure): tanh-gated cross-attention fusion, LayerNorm,
sinusoidal positional encoding.  fp32 CPU restatement of
``models/gate_cross_attention.py`` and ``models/utils.py`` of the reference.
"""
import math
import torch
import torch.nn.functional as F


def layer_norm_fp32(x, weight, bias, eps=1e-5):
    """``LayerNorm.forward`` (models/gate_cross_attention.py:8-10): normalise in
    fp32, cast back to the input dtype."""
    return F.layer_norm(x.float(), (x.shape[-1],), weight.float(), bias.float(), eps).type(x.dtype)


def positional_encoding_table(d_model, max_len=5000):
    """The ``pe`` buffer of ``PositionalEncoding`` (models/utils.py:12-18):
    interleaved sin/cos, shape [1, max_len, d_model]."""
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    denominator = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * denominator)
    pe[:, 1::2] = torch.cos(position * denominator)
    return pe.unsqueeze(0)


def positional_encoding(x, pe):
    """``PositionalEncoding.forward`` (models/utils.py:21-28)."""
    return x + pe[:, : x.size(1)]


def mha_forward(q_in, kv_in, in_w, in_b, out_w, out_b, n_heads, key_padding_mask=None,
                attn_keep=None, p_drop=0.0, masked_row="nan"):
    """``nn.MultiheadAttention(batch_first=True)`` as called at
    models/gate_cross_attention.py:40: packed in-projection, queries from
    ``q_in``, keys and values from ``kv_in``, boolean ``key_padding_mask``
    (True = ignore key), dropout on the attention weights.

    ``attn_keep`` is an explicit 0/1 keep tensor [B, h, Tq, Tk] so a device RNG
    can be replayed; kept weights are scaled by 1/(1-p).
    ``masked_row``: "nan" reproduces torch (softmax over an all -inf row);
    "zero" makes a fully masked query row attend to nothing (output = bias).
    """
    B, Tq, d = q_in.shape
    Tk = kv_in.shape[1]
    hd = d // n_heads
    wq, wk, wv = in_w[:d], in_w[d:2 * d], in_w[2 * d:]
    bq, bk, bv = in_b[:d], in_b[d:2 * d], in_b[2 * d:]
    q = F.linear(q_in, wq, bq).view(B, Tq, n_heads, hd).transpose(1, 2)
    k = F.linear(kv_in, wk, bk).view(B, Tk, n_heads, hd).transpose(1, 2)
    v = F.linear(kv_in, wv, bv).view(B, Tk, n_heads, hd).transpose(1, 2)
    s = torch.matmul(q * (hd ** -0.5), k.transpose(-1, -2))          # [B,h,Tq,Tk]
    if key_padding_mask is not None:
        s = s.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
    if masked_row == "zero":
        dead = torch.isinf(s).all(dim=-1, keepdim=True)
        p = torch.softmax(s.masked_fill(dead, 0.0), dim=-1).masked_fill(dead, 0.0)
    else:
        p = torch.softmax(s, dim=-1)
    if attn_keep is not None and p_drop > 0.0:
        p = p * attn_keep.to(p.dtype) / (1.0 - p_drop)
    o = torch.matmul(p, v).transpose(1, 2).reshape(B, Tq, d)
    return F.linear(o, out_w, out_b)


def gated_block(sd, pfx, x, xa, mask, n_heads, attn_keep=None, ff_keep=None, p_drop=0.0,
                masked_row="nan"):
    """``GatedCrossAttentionBlock.forward`` (models/gate_cross_attention.py:28-50)."""
    if xa is not None and xa.dim() == 4 and xa.size(1) == 1:
        xa = xa.squeeze(1)
    if mask is not None and mask.dim() == 3 and mask.size(1) == 1:
        mask = mask.squeeze(1)
    if xa is not None:
        h = layer_norm_fp32(x, sd[pfx + "attn_ln.weight"], sd[pfx + "attn_ln.bias"])
        a = mha_forward(h, xa, sd[pfx + "attn.in_proj_weight"], sd[pfx + "attn.in_proj_bias"],
                        sd[pfx + "attn.out_proj.weight"], sd[pfx + "attn.out_proj.bias"],
                        n_heads, mask, attn_keep, p_drop, masked_row)
        x = x + a * sd[pfx + "attn_gate"].tanh()
    h = layer_norm_fp32(x, sd[pfx + "ff_ln.weight"], sd[pfx + "ff_ln.bias"])
    f = F.linear(F.gelu(F.linear(h, sd[pfx + "ff.0.weight"], sd[pfx + "ff.0.bias"])),
                 sd[pfx + "ff.2.weight"], sd[pfx + "ff.2.bias"])
    if ff_keep is not None and p_drop > 0.0:
        f = f * ff_keep.to(f.dtype) / (1.0 - p_drop)
    return x + f * sd[pfx + "ff_gate"].tanh()


def fusion_forward(sd, pfx, audio, video, n_heads, video_mask=None, drop=None, p_drop=0.0,
                   masked_row="nan"):
    """``GatedCrossModalFusion.forward`` (models/gate_cross_attention.py:60-73).
    ``audio_mask`` is accepted by the reference and never used, so it has no
    counterpart here.  ``drop`` = {"attn": [keep per layer], "ff": [keep per layer]}."""
    x = F.linear(audio, sd[pfx + "audio_proj.weight"], sd[pfx + "audio_proj.bias"])
    xa = F.linear(video, sd[pfx + "video_proj.weight"], sd[pfx + "video_proj.bias"])
    i = 0
    while (pfx + f"layers.{i}.attn_gate") in sd:
        ak = drop["attn"][i] if drop else None
        fk = drop["ff"][i] if drop else None
        x = gated_block(sd, pfx + f"layers.{i}.", x, xa, video_mask, n_heads, ak, fk, p_drop,
                        masked_row)
        i += 1
    return layer_norm_fp32(x, sd[pfx + "ln_post.weight"], sd[pfx + "ln_post.bias"])
