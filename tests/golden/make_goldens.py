"""Generate the golden vectors under tests/golden/ (run ONLY in the build container, where the
reference checkout is mounted at /root/reference; nothing of the reference travels with the repo).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_goldens.py

What is pinned, and by what:
  fusion.npz    - the reference's own GatedCrossModalFusion / LayerNorm (models/gate_cross_attention.py)
                  forward + backward on seeded weights (gates non-zero), ragged key mask, eval mode, fp32.
  head_pe.npz   - the reference's PositionalEncoding and Decoder (models/utils.py) + the loss arithmetic of
                  train.py:95-128 (core torch CTC / CE calls, restated here because train.py itself needs
                  pytorch_lightning, which is not installed).
  whisper.npz   - transformers' WhisperEncoder (the class models/av_net.py:78 calls), eager attention, random
                  init from the same seeded recipe: a micro config in full, Whisper-tiny as row samples + sums.
  whisper_decoder.npz - transformers' WhisperDecoder (the ``whisper_model.decoder`` the reference constructs at
                  models/av_net.py:22 and never calls), eager attention, same seeded recipe: micro config in full,
                  with and without a padded decoder attention_mask.
  logmel.npz    - transformers' WhisperFeatureExtractor built from the reference's TW_tokenizer/
                  preprocessor_config.json, on a seeded 10 s waveform.
Inputs are regenerated from seeds on both sides (tests/helpers.py); only outputs are stored.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from tests.helpers import seeded_values  # noqa: E402


def load_seeded(module, seed):
    sd = module.state_dict()
    vals = seeded_values({k: (tuple(v.shape), v.dtype) for k, v in sd.items()}, seed)
    module.load_state_dict(vals, strict=False)
    return module


def fusion_golden():
    from models.gate_cross_attention import GatedCrossModalFusion
    d, H, N, B, T = 512, 8, 3, 2, 75
    m = load_seeded(GatedCrossModalFusion(d, H, N, dropout=0.1), 13).eval()
    g = torch.Generator().manual_seed(2)
    a = torch.randn(B, T, d, generator=g).requires_grad_(True)
    v = torch.randn(B, T, d, generator=g).requires_grad_(True)
    mask = torch.zeros(B, T, dtype=torch.bool)
    mask[1, 40:] = True
    dout = torch.randn(B, T, d, generator=g)
    y = m(a, v, None, mask)
    y.backward(dout)
    out = {"y": y.detach().numpy(), "da": a.grad.numpy(), "dv": v.grad.numpy()}
    for k, p in m.named_parameters():
        out["gnorm/" + k] = np.float64(p.grad.double().norm().item())
        out["gsum/" + k] = np.float64(p.grad.double().sum().item())
    np.savez_compressed(os.path.join(HERE, "fusion.npz"), **out)


def head_pe_golden():
    import torch.nn.functional as F
    from models.utils import Decoder, PositionalEncoding
    from models.gate_cross_attention import LayerNorm
    d, V, B, T, L = 128, 503, 2, 6, 4
    pe = PositionalEncoding(d)
    dec = load_seeded(Decoder(d, V), 31)
    ln = load_seeded(LayerNorm(d), 32)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(B, T, d, generator=g)
    logits = dec(pe(ln(x)))
    targets = torch.randint(1, V, (B, L), generator=g)
    il, tl = torch.full((B,), T), torch.full((B,), L)
    lp = F.log_softmax(logits, -1).transpose(0, 1)
    ctc = F.ctc_loss(lp, targets, il, tl, blank=0, reduction="mean", zero_infinity=True)
    ce = F.cross_entropy(logits[:, :L].reshape(-1, V), targets.reshape(-1), ignore_index=-100, label_smoothing=0.1)
    np.savez_compressed(os.path.join(HERE, "head_pe.npz"), pe_slice=pe.pe[0, :64, :].numpy(),
                        logits=logits.detach().numpy(), argmax=logits.argmax(-1).numpy(),
                        ctc=np.float64(ctc.item()), ce=np.float64(ce.item()))


def whisper_golden():
    from transformers import WhisperConfig, WhisperModel
    out = {}
    for name, (D, Ly, H, B) in {"micro": (128, 2, 2, 1), "tiny": (384, 4, 6, 2)}.items():
        cfg = WhisperConfig(d_model=D, encoder_layers=Ly, encoder_attention_heads=H, encoder_ffn_dim=4 * D,
                            decoder_layers=1, decoder_attention_heads=H, decoder_ffn_dim=4 * D, num_mel_bins=80,
                            vocab_size=51865, attn_implementation="eager")
        enc = load_seeded(WhisperModel(cfg).encoder, 7).eval()
        g = torch.Generator().manual_seed(3)
        mel = torch.randn(B, 80, 3000, generator=g)
        with torch.no_grad():
            y = enc(mel)[0]
        if name == "micro":
            out["micro/y"] = y.numpy()
        else:
            out["tiny/first"] = y[:, :8].numpy()
            out["tiny/last"] = y[:, -8:].numpy()
            out["tiny/sum"] = np.float64(y.double().sum().item())
            out["tiny/abssum"] = np.float64(y.double().abs().sum().item())
    np.savez_compressed(os.path.join(HERE, "whisper.npz"), **out)


def whisper_decoder_golden():
    from transformers import WhisperConfig, WhisperModel
    D, Ly, H, B, T, Te, V = 128, 2, 2, 2, 12, 40, 503
    cfg = WhisperConfig(d_model=D, encoder_layers=1, encoder_attention_heads=H, encoder_ffn_dim=4 * D,
                        decoder_layers=Ly, decoder_attention_heads=H, decoder_ffn_dim=4 * D, num_mel_bins=80,
                        vocab_size=V, max_target_positions=64, pad_token_id=0, bos_token_id=1, eos_token_id=2,
                        decoder_start_token_id=1, attn_implementation="eager")
    dec = load_seeded(WhisperModel(cfg).decoder, 17).eval()
    g = torch.Generator().manual_seed(9)
    ids = torch.randint(3, V, (B, T), generator=g)
    enc = torch.randn(B, Te, D, generator=g)
    am = torch.ones(B, T, dtype=torch.long)
    am[1, 9:] = 0
    with torch.no_grad():
        y = dec(input_ids=ids, encoder_hidden_states=enc, use_cache=False)[0]
        ym = dec(input_ids=ids, attention_mask=am, encoder_hidden_states=enc, use_cache=False)[0]
    np.savez_compressed(os.path.join(HERE, "whisper_decoder.npz"), y=y.numpy(), y_masked=ym.numpy())


def logmel_golden():
    from transformers import WhisperFeatureExtractor
    fe = WhisperFeatureExtractor.from_pretrained("/root/reference/TW_tokenizer")
    g = torch.Generator().manual_seed(5)
    wave = (0.1 * torch.randn(160000, generator=g)).numpy()
    feats = fe(wave, sampling_rate=16000, return_tensors="np")["input_features"][0]      # [80, 3000]
    np.savez_compressed(os.path.join(HERE, "logmel.npz"), feats_sub=feats[:, ::7].astype(np.float32),
                        total=np.float64(feats.astype(np.float64).sum()))


if __name__ == "__main__":
    torch.manual_seed(0)
    fusion_golden()
    head_pe_golden()
    whisper_golden()
    whisper_decoder_golden()
    logmel_golden()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
