"""Oracle (test infrastructure): the Whisper encoder the reference calls at
models/av_net.py:78-81.  fp32 CPU restatement of ``WhisperEncoder.forward``
(transformers ``modeling_whisper.py``, pinned 4.36.2 in the reference's
environment.yml; 5.15 in this image - the encoder arithmetic is unchanged).
"""
import torch
import torch.nn.functional as F


def whisper_attention(sd, pfx, x, n_heads):
    """``WhisperAttention.forward`` without mask or cache: q is scaled by
    head_dim**-0.5 before the matmul, k_proj has no bias."""
    B, T, D = x.shape
    hd = D // n_heads
    q = F.linear(x, sd[pfx + "q_proj.weight"], sd[pfx + "q_proj.bias"]) * (hd ** -0.5)
    k = F.linear(x, sd[pfx + "k_proj.weight"])
    v = F.linear(x, sd[pfx + "v_proj.weight"], sd[pfx + "v_proj.bias"])
    q = q.view(B, T, n_heads, hd).transpose(1, 2)
    k = k.view(B, T, n_heads, hd).transpose(1, 2)
    v = v.view(B, T, n_heads, hd).transpose(1, 2)
    p = torch.softmax(torch.matmul(q, k.transpose(2, 3)), dim=-1)
    o = torch.matmul(p, v).transpose(1, 2).reshape(B, T, D)
    return F.linear(o, sd[pfx + "out_proj.weight"], sd[pfx + "out_proj.bias"])


def whisper_encoder_layer(sd, pfx, x, n_heads):
    """``WhisperEncoderLayer.forward``: pre-LN attention and pre-LN GELU MLP."""
    D = x.shape[-1]
    h = F.layer_norm(x, (D,), sd[pfx + "self_attn_layer_norm.weight"],
                     sd[pfx + "self_attn_layer_norm.bias"])
    x = x + whisper_attention(sd, pfx + "self_attn.", h, n_heads)
    h = F.layer_norm(x, (D,), sd[pfx + "final_layer_norm.weight"], sd[pfx + "final_layer_norm.bias"])
    h = F.gelu(F.linear(h, sd[pfx + "fc1.weight"], sd[pfx + "fc1.bias"]))
    return x + F.linear(h, sd[pfx + "fc2.weight"], sd[pfx + "fc2.bias"])


def whisper_encoder(sd, pfx, mel, n_heads, return_hidden=False):
    """mel [B, n_mels, 3000] -> [B, 1500, D].  The attention mask argument of the
    reference call is ignored by the encoder, so it does not appear here.
    Raises ValueError on a wrong frame count, as the original does."""
    n_pos = sd[pfx + "embed_positions.weight"].shape[0]
    expected = n_pos * 2  # conv1 stride 1, conv2 stride 2
    if mel.shape[-1] != expected:
        raise ValueError(
            f"Whisper expects the mel input features to be of length {expected}, "
            f"but found {mel.shape[-1]}.")
    x = F.gelu(F.conv1d(mel, sd[pfx + "conv1.weight"], sd[pfx + "conv1.bias"], padding=1))
    x = F.gelu(F.conv1d(x, sd[pfx + "conv2.weight"], sd[pfx + "conv2.bias"], stride=2, padding=1))
    x = x.permute(0, 2, 1) + sd[pfx + "embed_positions.weight"]
    hidden = [x]
    i = 0
    while (pfx + f"layers.{i}.fc1.weight") in sd:
        x = whisper_encoder_layer(sd, pfx + f"layers.{i}.", x, n_heads)
        hidden.append(x)
        i += 1
    D = x.shape[-1]
    x = F.layer_norm(x, (D,), sd[pfx + "layer_norm.weight"], sd[pfx + "layer_norm.bias"])
    return (x, hidden) if return_hidden else x


def whisper_decoder_attention(sd, pfx, x, kv, n_heads, causal=False, key_keep=None):
    """``WhisperAttention.forward`` as the decoder uses it (transformers ``modeling_whisper.py:284-356``):
    queries from ``x``, keys/values from ``kv`` (``x`` itself for the self-attention, the encoder output for
    ``encoder_attn``); q scaled by head_dim**-0.5 before the matmul, k_proj without bias.  ``causal`` adds the
    lower-triangular mask aligned to the END of the key sequence (queries are its last Tq positions, the cached
    decoding case); ``key_keep`` [B,Tk] bool (True = attend) is the decoder ``attention_mask``."""
    B, Tq, D = x.shape
    Tk = kv.shape[1]
    hd = D // n_heads
    q = F.linear(x, sd[pfx + "q_proj.weight"], sd[pfx + "q_proj.bias"]) * (hd ** -0.5)
    k = F.linear(kv, sd[pfx + "k_proj.weight"])
    v = F.linear(kv, sd[pfx + "v_proj.weight"], sd[pfx + "v_proj.bias"])
    q = q.view(B, Tq, n_heads, hd).transpose(1, 2)
    k = k.view(B, Tk, n_heads, hd).transpose(1, 2)
    v = v.view(B, Tk, n_heads, hd).transpose(1, 2)
    s = torch.matmul(q, k.transpose(2, 3))
    neg = torch.finfo(s.dtype).min
    if causal:
        qi = torch.arange(Tq).view(Tq, 1) + (Tk - Tq)
        s = s.masked_fill(torch.arange(Tk).view(1, Tk) > qi, neg)
    if key_keep is not None:
        s = s.masked_fill(~key_keep.view(B, 1, 1, Tk), neg)
    o = torch.matmul(torch.softmax(s, dim=-1), v).transpose(1, 2).reshape(B, Tq, D)
    return F.linear(o, sd[pfx + "out_proj.weight"], sd[pfx + "out_proj.bias"])


def whisper_decoder_layer(sd, pfx, x, enc, n_heads, key_keep=None):
    """``WhisperDecoderLayer.forward`` (transformers ``modeling_whisper.py:416-505``), eval mode: pre-LN causal
    self-attention, pre-LN cross-attention to the encoder output, pre-LN GELU MLP."""
    D = x.shape[-1]
    ln = lambda t, name: F.layer_norm(t, (D,), sd[pfx + name + ".weight"], sd[pfx + name + ".bias"])
    h = ln(x, "self_attn_layer_norm")
    x = x + whisper_decoder_attention(sd, pfx + "self_attn.", h, h, n_heads, causal=True, key_keep=key_keep)
    if enc is not None:
        h = ln(x, "encoder_attn_layer_norm")
        x = x + whisper_decoder_attention(sd, pfx + "encoder_attn.", h, enc, n_heads)
    h = ln(x, "final_layer_norm")
    h = F.gelu(F.linear(h, sd[pfx + "fc1.weight"], sd[pfx + "fc1.bias"]))
    return x + F.linear(h, sd[pfx + "fc2.weight"], sd[pfx + "fc2.bias"])


def whisper_decoder(sd, pfx, input_ids, enc, n_heads, attention_mask=None):
    """``WhisperDecoder.forward`` (transformers ``modeling_whisper.py:649-797``) without a cache: token embedding
    (embed_scale 1) + learned positions ``embed_positions.weight[:T]``, the layers, the final LayerNorm.
    ``attention_mask`` [B,T] (1 = token, 0 = padding) masks keys of the self-attention only."""
    T = input_ids.shape[1]
    x = F.embedding(input_ids, sd[pfx + "embed_tokens.weight"]) + sd[pfx + "embed_positions.weight"][:T]
    keep = None if attention_mask is None else attention_mask.bool()
    i = 0
    while (pfx + f"layers.{i}.fc1.weight") in sd:
        x = whisper_decoder_layer(sd, pfx + f"layers.{i}.", x, enc, n_heads, keep)
        i += 1
    D = x.shape[-1]
    return F.layer_norm(x, (D,), sd[pfx + "layer_norm.weight"], sd[pfx + "layer_norm.bias"])
