"""MI355X-native (gfx950) audio-visual speech-recognition forward/backward path
behind the reference's ``AVNet`` module surface.  Compute runs in hand-written
HIP kernels (``csrc/``) reached through a C ABI (``include/avsr_hip.h``);
PyTorch supplies device memory, streams and ``torch.distributed`` only."""
from . import _lib, ops  # noqa: F401
from .av_net import AVNet
from .logmel import LogMelFrontend
from .transforms import VideoTransform
from .gate_cross_attention import GatedCrossAttentionBlock, GatedCrossModalFusion, LayerNorm
from .moco_visual_frontend import MoCoVisualFrontend
from .utils import Decoder, PositionalEncoding
from .whisper_encoder import WhisperConfig, WhisperEncoder, WhisperModel
from .whisper_decoder import WhisperDecoder

__all__ = ["AVNet", "GatedCrossAttentionBlock", "GatedCrossModalFusion", "LayerNorm", "LogMelFrontend", "MoCoVisualFrontend",
           "Decoder", "PositionalEncoding", "VideoTransform", "WhisperConfig", "WhisperDecoder", "WhisperEncoder", "WhisperModel", "ops"]
