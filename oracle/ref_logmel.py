"""Oracle (test infrastructure): the two mel front ends of the reference.

* ``whisper_logmel``: ``WhisperFeatureExtractor._torch_extract_fbank_features``
  (transformers feature_extraction_whisper.py:135-168; reference call site
  datamodule/av_dataset.py:47-51).  Pinned against ``transformers`` in tests.
* ``torchaudio_melspec_norm``: ``torchaudio.transforms.MelSpectrogram(16 kHz,
  n_fft 400, hop 160, 80 mels)`` followed by a whole-tensor layer_norm
  (datamodule/transforms.py:115-121,131).  PARITY UNPINNED: torchaudio is not
  installed; restated from its documented defaults (HTK mel scale, no filter
  normalisation, power 2, periodic hann, centre + reflect padding).
"""
import math
import numpy as np
import torch
import torch.nn.functional as F


def _hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    min_log_hz, min_log_mel, logstep = 1000.0, 15.0, 27.0 / np.log(6.4)
    mel = 3.0 * f / 200.0
    log_region = f >= min_log_hz
    mel = np.where(log_region, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) * logstep, mel)
    return mel


def _mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    min_log_hz, min_log_mel, logstep = 1000.0, 15.0, np.log(6.4) / 27.0
    f = 200.0 * m / 3.0
    log_region = m >= min_log_mel
    return np.where(log_region, min_log_hz * np.exp(logstep * (m - min_log_mel)), f)


def mel_filters_slaney(n_freqs=201, n_mels=80, f_min=0.0, f_max=8000.0, sr=16000):
    """Triangular filters on the Slaney mel scale with Slaney area normalisation,
    [n_freqs, n_mels] (the bank WhisperFeatureExtractor builds at
    feature_extraction_whisper.py:95-103)."""
    fft_freqs = np.linspace(0, sr // 2, n_freqs)
    mel_pts = np.linspace(_hz_to_mel_slaney(f_min), _hz_to_mel_slaney(f_max), n_mels + 2)
    hz_pts = _mel_to_hz_slaney(mel_pts)
    fdiff = np.diff(hz_pts)
    slopes = hz_pts[None, :] - fft_freqs[:, None]
    down = -slopes[:, :-2] / fdiff[:-1]
    up = slopes[:, 2:] / fdiff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    enorm = 2.0 / (hz_pts[2:n_mels + 2] - hz_pts[:n_mels])
    return (fb * enorm[None, :]).astype(np.float32)


def mel_filters_htk(n_freqs=201, n_mels=80, f_min=0.0, f_max=8000.0, sr=16000):
    """torchaudio ``melscale_fbanks(norm=None, mel_scale='htk')``, [n_freqs, n_mels]."""
    all_freqs = torch.linspace(0, sr // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.max(torch.zeros(1), torch.min(down, up))


def power_stft(wave, n_fft=400, hop=160):
    """|STFT|^2 with a periodic hann window, centre + reflect padding:
    [B, n_fft//2+1, 1 + len//hop]."""
    window = torch.hann_window(n_fft)
    st = torch.stft(wave, n_fft, hop, window=window, return_complex=True)
    return st.abs() ** 2


def whisper_logmel(wave):
    """wave [B, n_samples] f32 -> [B, 80, n_samples//160] log-mel in Whisper's
    normalisation: log10(max(mel, 1e-10)), floor at (per-clip max - 8), (x+4)/4."""
    mag = power_stft(wave)[..., :-1]
    fb = torch.from_numpy(mel_filters_slaney())
    mel = fb.T @ mag
    log_spec = torch.clamp(mel, min=1e-10).log10()
    mx = log_spec.amax(dim=(1, 2), keepdim=True)
    log_spec = torch.maximum(log_spec, mx - 8.0)
    return (log_spec + 4.0) / 4.0


def torchaudio_melspec_norm(wave, target_len=3000):
    """One clip per row: MelSpectrogram -> pad/trim to ``target_len`` frames ->
    [T, 80] -> layer_norm over the whole [T, 80] tensor with eps 1e-8
    (transforms.py:134-142, eval pipeline without noise)."""
    mag = power_stft(wave)
    mel = (mag.transpose(1, 2) @ mel_filters_htk()).transpose(1, 2)       # [B,80,frames]
    t = mel.shape[-1]
    if t < target_len:
        mel = F.pad(mel, (0, target_len - t))
    else:
        mel = mel[..., :target_len]
    x = mel.transpose(1, 2)                                               # [B,T,80]
    return torch.stack([F.layer_norm(xi, xi.shape, eps=1e-8) for xi in x])


def spec_augment(mel_ft, frame_starts, bin_starts, frame_width=48):
    """``AudioTransform._apply_spec_augment`` as the reference EXECUTES it.  The pipeline calls it at
    transforms.py:127 on the MelSpectrogram output [80, frames]; the transpose to [frames, 80] only comes two
    steps later (:129).  The function indexes that tensor as (time x frequency), so on the [80, frames] tensor
    its "freq" loop (:152-164: width 48, start in [0, size(1) - 48), columns ``[:, s:s+48]``) zeroes two bands of
    48 FRAMES, and its "time" loop (:167-177: width size(0) // 8 = 10, start in [0, 80 - 10), rows
    ``[s:s+10, :]``) zeroes two bands of 10 MEL BINS - both before pad/trim (:128).  ``frame_starts`` /
    ``bin_starts`` are the random starts, in that order of drawing (negative = band skipped)."""
    x = mel_ft.clone()
    n_bins, frames = x.shape
    for s0 in frame_starts:
        if s0 >= 0 and frames - frame_width > 0:
            x[:, s0:s0 + frame_width] = 0
    bw = n_bins // 8
    for s0 in bin_starts:
        if s0 >= 0 and n_bins - bw > 0:
            x[s0:s0 + bw, :] = 0
    return x


def add_noise(waveform, noise, snr_db):
    """``torchaudio.functional.add_noise`` (torchaudio 2.4.1, absent here: restated from its documented formula -
    PARITY UNPINNED): scales ``noise`` so that each row of ``waveform`` reaches the requested SNR."""
    e_sig = torch.linalg.vector_norm(waveform, ord=2, dim=-1) ** 2
    e_noise = torch.linalg.vector_norm(noise, ord=2, dim=-1) ** 2
    original_snr_db = 10 * (torch.log10(e_sig) - torch.log10(e_noise))
    scale = 10 ** ((original_snr_db - snr_db) / 20.0)
    return waveform + scale.unsqueeze(-1) * noise


def torchaudio_melspec_train(wave, bin_starts, frame_starts, noise_segments, snr_db, target_len=3000):
    """``AudioTransform`` training pipeline (transforms.py:122-132), one clip per row, with the random draws
    given: MelSpectrogram -> SpecAugment on the [80, frames] tensor (see ``spec_augment``) -> pad/trim to 3000
    frames -> transpose -> AddNoise (transforms.py:53-59 transposes back: the mel rows are treated as audio
    channels) -> whole-clip layer norm."""
    mag = power_stft(wave)
    mel = (mag.transpose(1, 2) @ mel_filters_htk()).transpose(1, 2)       # [B,80,frames]
    out = []
    for b in range(mel.shape[0]):
        x = spec_augment(mel[b], frame_starts[b], bin_starts[b])         # [80, frames]
        t = x.shape[-1]
        x = F.pad(x, (0, target_len - t)) if t < target_len else x[..., :target_len]
        x = add_noise(x, noise_segments[b].view(1, -1), torch.tensor([float(snr_db[b])]))
        x = x.t()
        out.append(F.layer_norm(x, x.shape, eps=1e-8))
    return torch.stack(out)
