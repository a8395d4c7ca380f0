"""Oracle (test infrastructure): the whole audio-visual path - fp32 CPU
restatement of ``AVNet.forward`` (models/av_net.py:67-114) and of the loss the
training loop puts on its logits (train.py:95-128,143-165).
"""
import torch
import torch.nn.functional as F

from .ref_fusion import layer_norm_fp32, positional_encoding, fusion_forward
from .ref_whisper import whisper_encoder
from .ref_visual import visual_frontend

# keys autograd reaches in the reference: everything except the frozen Whisper
# encoder and the frozen visual frontend (av_net.py:25-26,45-46); the Whisper
# decoder and fusion_scalar require grad but never receive one.
TRAINABLE_PREFIXES = ("audio_proj.", "audio_ln.", "video_proj.", "video_ln.",
                      "fusion_module.", "decoder.linear.")


def make_padding_mask(lengths, max_length):
    """``AVNet.make_padding_mask`` (av_net.py:111-114): True = padded."""
    return torch.arange(max_length).expand(lengths.size(0), max_length) >= lengths.unsqueeze(1)


def avnet_forward(sd, input_batch, whisper_heads, fusion_heads, training=False, new_stats=None,
                  drop=None, p_drop=0.0, masked_row="nan", return_parts=False, long_form=False):
    """``AVNet.forward``.  ``input_batch`` = (audio, audio_mask, video, video_mask,
    video_len).  ``video_len`` is clamped in place, as the reference does.
    ``long_form`` (no counterpart in the reference, whose encoder rejects anything but 3000
    frames): audio of k x 3000 frames is cut into k windows, each goes through the
    reference encoder on its own, and the outputs are concatenated in time."""
    audio, audio_mask, video, _video_mask, video_len = input_batch
    if audio.shape[2] == 80 and audio.shape[1] % 3000 == 0:
        audio = audio.transpose(1, 2)
    if long_form and audio.shape[2] > 3000:
        enc = torch.cat([whisper_encoder(sd, "whisper_model.encoder.", w, whisper_heads)
                         for w in audio.split(3000, dim=2)], dim=1)
    else:
        enc = whisper_encoder(sd, "whisper_model.encoder.", audio, whisper_heads)
    a = layer_norm_fp32(F.linear(enc, sd["audio_proj.weight"], sd["audio_proj.bias"]),
                        sd["audio_ln.weight"], sd["audio_ln.bias"])
    a = positional_encoding(a, sd["pos_enc_audio.pe"])
    vis = visual_frontend(sd, "visual_model.", video, video_len, training, new_stats)
    v = layer_norm_fp32(F.linear(vis, sd["video_proj.weight"], sd["video_proj.bias"]),
                        sd["video_ln.weight"], sd["video_ln.bias"])
    v = positional_encoding(v, sd["pos_enc_video.pe"])
    min_len = min(a.size(1), v.size(1))
    a, v = a[:, :min_len], v[:, :min_len]
    video_len.clamp_(max=min_len)
    fused = fusion_forward(sd, "fusion_module.", a, v, fusion_heads,
                           make_padding_mask(video_len, min_len), drop, p_drop, masked_row)
    out = fused + a + v
    logits = F.linear(out, sd["decoder.linear.weight"], sd["decoder.linear.bias"])
    if return_parts:
        return logits, {"encoder": enc, "visual": vis, "audio_feat": a, "video_feat": v,
                        "fused": fused, "pre_head": out}
    return logits


def ctc_ce_loss(logits, target_ids, input_lengths, target_lengths, label_smoothing=0.1):
    """``_compute_ctc_loss`` + ``_compute_ce_loss`` (train.py:95-128): CTC(blank 0,
    mean, zero_infinity) on log_softmax(logits), plus label-smoothed CE between the
    first min(T, L) frame logits and the first min(T, L) target ids."""
    V = logits.shape[-1]
    log_probs = F.log_softmax(logits, dim=-1).transpose(0, 1)
    ctc = F.ctc_loss(log_probs, target_ids, input_lengths, target_lengths, blank=0,
                     reduction="mean", zero_infinity=True)
    t_min = min(logits.shape[1], target_ids.shape[1])
    ce = F.cross_entropy(logits[:, :t_min, :].reshape(-1, V), target_ids[:, :t_min].reshape(-1),
                         ignore_index=-100, label_smoothing=label_smoothing)
    return ctc, ce


def training_step(sd, batch, whisper_heads, fusion_heads, training=False, label_smoothing=0.1,
                  drop=None, p_drop=0.0, masked_row="nan", long_form=False):
    """``AVSRModule.training_step`` (train.py:143-170) followed by backward.
    Returns (loss, ctc, ce, logits, grads) with grads keyed like ``sd`` for the
    parameters autograd reaches."""
    leaf = dict(sd)
    names = [k for k in sd if k.startswith(TRAINABLE_PREFIXES) and sd[k].is_floating_point()]
    for k in names:
        leaf[k] = sd[k].detach().clone().requires_grad_(True)
    inp = (batch["audio"], batch["audio_mask"], batch["video"], batch["video_mask"],
           batch["video_lengths"].clone())
    new_stats = {} if training else None
    logits = avnet_forward(leaf, inp, whisper_heads, fusion_heads, training, new_stats, drop,
                           p_drop, masked_row, long_form=long_form)
    ctc, ce = ctc_ce_loss(logits, batch["target_ids"], batch["audio_lengths"],
                          batch["target_lengths"].reshape(-1), label_smoothing)
    loss = ctc + ce
    grads = torch.autograd.grad(loss, [leaf[k] for k in names], allow_unused=True)
    return (loss.detach(), ctc.detach(), ce.detach(), logits.detach(),
            {k: g for k, g in zip(names, grads) if g is not None}, new_stats)
