"""Mel front ends of the reference's data path, on the GPU.

``LogMelFrontend("whisper")`` stands in for ``WhisperFeatureExtractor`` as called at
datamodule/av_dataset.py:47-51 (transformers feature_extraction_whisper.py:135-168):
a waveform of up to 30 s is read as zero-padded to 480 000 samples and becomes
``[B, 80, 3000]`` log-mel features.  ``LogMelFrontend("melspec_norm")`` stands in for
``AudioTransform``'s evaluation pipeline (datamodule/transforms.py:115-121,134-142):
torchaudio's ``MelSpectrogram`` -> pad/trim to 3000 frames -> whole-clip layer norm,
returned as ``[B, 3000, 80]`` like the pipeline's transpose does;
``LogMelFrontend("melspec_norm").train(...)`` is its training pipeline
(transforms.py:122-132): SpecAugment bands (48 frames / 10 mel bins, as the reference applies
them to the [80, frames] tensor) and babble noise at a random SNR between
the mel bank and the layer norm, with the random draws made on the host as the
reference does.
"""
import random

import torch

from . import ops

_VARIANTS = {"whisper": ops.MEL_WHISPER, "melspec_norm": ops.MEL_SPEC_LN}


class LogMelFrontend:
    N_SAMPLES = 480000
    N_FRAMES = 3000

    def __init__(self, variant="whisper", device="cuda"):
        if variant not in _VARIANTS:
            raise ValueError(f"variant must be one of {sorted(_VARIANTS)}")
        self.variant = variant
        self.code = _VARIANTS[variant]
        self.device = torch.device(device)
        self.tables = ops.logmel_tables(self.code, self.device)

    def __call__(self, wave, *, dtype=torch.float32, channels_last=None):
        """wave: [n] or [B, n] fp32 on the GPU.  ``channels_last`` defaults to the layout the
        replaced pipeline produces: False ([B,80,T]) for "whisper", True ([B,T,80]) for
        "melspec_norm"; AVNet.forward accepts both (models/av_net.py:75-76)."""
        squeeze = wave.dim() == 1
        if squeeze:
            wave = wave.unsqueeze(0)
        if channels_last is None:
            channels_last = self.code == ops.MEL_SPEC_LN
        if self.code == ops.MEL_WHISPER:
            if wave.shape[1] > self.N_SAMPLES:
                wave = wave[:, :self.N_SAMPLES]                       # the extractor truncates to 30 s
            out = ops.logmel_fwd(wave.contiguous(), self.tables, self.code, n_samples=self.N_SAMPLES,
                                 n_frames=self.N_FRAMES, out_dtype=dtype, channels_last=channels_last)
        else:
            out = ops.logmel_fwd(wave.contiguous(), self.tables, self.code, n_frames=self.N_FRAMES,
                                 out_dtype=dtype, channels_last=channels_last)
        return out[0] if squeeze else out

    SNR_LEVELS = (-5, 0, 5, 10, 15, 20, 999999)          # transforms.py:50

    BIN_BAND = 80 // 8           # transforms.py:167 on the [80, frames] tensor: size(0) // 8 mel bins
    FRAME_BAND = 48              # transforms.py:152,164 on the same tensor: 48 of its columns = frames

    def draw(self, batch, n_samples, noise_len, rng=random):
        """The random choices of one training batch, in the reference's order per clip.  SpecAugment runs on the
        [80, frames] mel (transforms.py:127, before the transpose at :129), so its first loop (:157-164) draws two
        starts in [0, frames - 48) for bands of 48 frames and its second (:171-177) two starts in [0, 80 - 10)
        for bands of 10 mel bins, over the un-padded frame count; then AddNoise draws a start offset into the
        noise recording and an SNR level (:55-57).  Returns (bin_starts, frame_starts, offsets, snr)."""
        frames = 1 + n_samples // 160
        fs, ts, off, snr = [], [], [], []
        for _ in range(batch):
            room = frames - self.FRAME_BAND
            ts.append([rng.randrange(0, room) if room > 0 else -1 for _ in range(2)])
            fs.append([rng.randrange(0, 80 - self.BIN_BAND) for _ in range(2)])
            off.append(rng.randint(0, noise_len - self.N_FRAMES))
            snr.append(float(rng.choice(self.SNR_LEVELS)))
        return fs, ts, off, snr

    def train(self, wave, noise_recording, *, draws=None, rng=random, dtype=torch.float32, channels_last=True):
        """wave [B, n] fp32 on the GPU, ``noise_recording`` a 1-D fp32 GPU tensor of at least 3000 values (the
        reference slices 3000 consecutive samples of babble_noise.wav and adds them to the mel rows).
        ``draws`` = (mel-bin band starts, frame band starts, noise_offsets, snr_db) overrides the random
        choices."""
        if self.code != ops.MEL_SPEC_LN:
            raise ValueError('the training pipeline belongs to the "melspec_norm" variant')
        B, n = wave.shape
        fs, ts, off, snr = draws if draws is not None else self.draw(B, n, noise_recording.numel(), rng)
        dev = wave.device
        noise = torch.stack([noise_recording[o:o + self.N_FRAMES] for o in off]).contiguous()
        return ops.logmel_augment_fwd(wave.contiguous(), self.tables, noise=noise,
                                      snr_db=torch.tensor(snr, dtype=torch.float32, device=dev),
                                      freq_start=torch.tensor(fs, dtype=torch.int32, device=dev),
                                      time_start=torch.tensor(ts, dtype=torch.int32, device=dev),
                                      freq_width=self.BIN_BAND, time_width=self.FRAME_BAND,
                                      n_frames=self.N_FRAMES, out_dtype=dtype, channels_last=channels_last)
