"""CPU oracle for the audio-visual forward/backward path.

TEST INFRASTRUCTURE ONLY.  This package is a plain-PyTorch, fp32, CPU restatement
of the reference's algorithm.  It exists so that the HIP path can be checked
against something independent of it.  Only ``tests/``, ``__graft_entry__.smoke``
and the ``cpu_baseline`` leg of ``bench.py`` may import it; the product package
never does, and fails loudly when its HIP library is missing.

Every function works on a flat ``state_dict`` whose keys are the reference's own
checkpoint names (``whisper_model.encoder.layers.0.fc1.weight``,
``visual_model.MoCoModel.layer1.0.conv1.weight``, ...), so the same tensors can be
loaded into the product module, the oracle and (in the build container only)
the reference's importable classes.

Pinning status
--------------
* fusion / LayerNorm / PositionalEncoding / Decoder: pinned against the
  reference's own classes (``models/gate_cross_attention.py``,
  ``models/utils.py``) through ``tests/golden/*.npz``.
* Whisper encoder, Whisper decoder and Whisper log-mel: pinned against the
  ``transformers`` implementation the reference constructs / calls
  (``models/av_net.py:22,78``), live in ``tests/test_oracle_cpu.py`` and
  through goldens.
* ResNet-50 trunk, 3D stem, torchaudio MelSpectrogram variant and its add_noise: PARITY UNPINNED -
  ``torchvision`` / ``torchaudio`` are not installed, and the reference holds no
  fixtures for them; the restatement follows the published architecture and is
  self-checked against ``torch.nn.functional`` convolution / batch-norm.
"""
from .ref_fusion import (layer_norm_fp32, positional_encoding, gated_block,
                         fusion_forward, mha_forward)
from .ref_whisper import whisper_encoder, whisper_decoder
from .ref_visual import visual_frontend
from .ref_avnet import avnet_forward, ctc_ce_loss, training_step, make_padding_mask
from .ref_logmel import whisper_logmel, torchaudio_melspec_norm, torchaudio_melspec_train
from .ref_video import video_transform_train, color_jitter
