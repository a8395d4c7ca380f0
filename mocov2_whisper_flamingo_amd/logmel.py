"""Mel front ends of the reference's data path, on the GPU.

``LogMelFrontend("whisper")`` stands in for ``WhisperFeatureExtractor`` as called at
datamodule/av_dataset.py:47-51 (transformers feature_extraction_whisper.py:135-168):
a waveform of up to 30 s is read as zero-padded to 480 000 samples and becomes
``[B, 80, 3000]`` log-mel features.  ``LogMelFrontend("melspec_norm")`` stands in for
``AudioTransform``'s evaluation pipeline (datamodule/transforms.py:115-121,134-142):
torchaudio's ``MelSpectrogram`` -> pad/trim to 3000 frames -> whole-clip layer norm,
returned as ``[B, 3000, 80]`` like the pipeline's transpose does.
"""
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
