"""MI355X-native (gfx950) audio-visual speech-recognition forward/backward path
behind the reference's ``AVNet`` module surface.  Compute runs in hand-written
HIP kernels (``csrc/``) reached through a C ABI (``include/avsr_hip.h``);
PyTorch supplies device memory, streams and ``torch.distributed`` only."""
from . import _lib, ops  # noqa: F401

__all__ = ["_lib", "ops"]
