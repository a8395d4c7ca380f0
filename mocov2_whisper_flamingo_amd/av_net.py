"""``AVNet`` with the reference's constructor / forward surface and state-dict keys
(models/av_net.py:14-114), running on the HIP kernels.

Drop-in use: ``AVNet(modal, MoCofile, reqInpLen, modelargs, vocab_size)`` then
``logits = model((audio, audio_mask, video, video_mask, video_len))``.
Keyword-only extensions (defaults keep the reference call valid):
``whisper_config`` (the reference hard-codes "openai/whisper-small" and fetches
it by name; there is no network here, so the architecture is instantiated from
a config with random weights, or from ``whisper_state`` tensors), ``dtype`` (the
compute dtype: ``torch.bfloat16`` or ``torch.float32``), ``device``,
``attention`` ("bf16" or "fp8": the products of the Whisper self-attention on
e4m3 MFMA) and ``long_form`` (audio of k x 3000 mel frames is encoded as k
independent 30 s windows whose outputs are concatenated in time; off, such
input raises ValueError exactly as the reference's encoder does).

There is no CPU path: the module's parameters must live on a GPU and the HIP
library must be built, otherwise the first call raises.
"""
import logging
import os
import zlib

import torch
import torch.nn as nn

from . import functional as Fn
from . import ops
from .gate_cross_attention import GatedCrossModalFusion, LayerNorm
from .moco_visual_frontend import MoCoVisualFrontend
from .utils import Decoder, PositionalEncoding
from .whisper_encoder import WhisperConfig, WhisperModel

logger = logging.getLogger(__name__)


class AVNet(nn.Module):
    def __init__(self, modal, MoCofile, reqInpLen, modelargs, vocab_size, enable_logging=False, *,
                 whisper_config="small", whisper_state=None, dtype=torch.bfloat16, device=None, seed=0,
                 attention="bf16", long_form=False):
        super().__init__()
        self.enable_logging = enable_logging
        self.modal = modal
        self.reqInpLen = reqInpLen
        self.compute_dtype = dtype
        if dtype not in (torch.bfloat16, torch.float32):
            raise ValueError("dtype must be torch.bfloat16 or torch.float32")

        cfg = whisper_config if isinstance(whisper_config, WhisperConfig) else WhisperConfig.from_name(whisper_config)
        self.whisper_model = WhisperModel(cfg)
        if whisper_state is not None:
            self.whisper_model.load_state_dict(whisper_state, strict=False)
        for p in self.whisper_model.encoder.parameters():      # models/av_net.py:25-26
            p.requires_grad = False
        if attention not in ("bf16", "fp8") or (attention == "fp8" and dtype != torch.bfloat16):
            raise ValueError('attention must be "bf16" or "fp8" (fp8 with the bf16 compute dtype only)')
        self.whisper_model.encoder.attn_fp8 = attention == "fp8"
        self.long_form = bool(long_form)

        d_model = modelargs[0]
        self.audio_proj = nn.Linear(cfg.d_model, d_model)
        self.audio_ln = LayerNorm(d_model)

        self.visual_model = MoCoVisualFrontend(enable_logging=enable_logging)
        if MoCofile is not None:                               # models/av_net.py:40-42
            state = torch.load(MoCofile, map_location="cpu", weights_only=True)
            self.visual_model.load_state_dict(state, strict=False)
        for p in self.visual_model.parameters():               # models/av_net.py:45-46
            p.requires_grad = False
        self.visual_model.compute_dtype = dtype

        self.video_proj = nn.Linear(2048, d_model)
        self.video_ln = LayerNorm(d_model)
        self.fusion_module = GatedCrossModalFusion(d_model=d_model, n_heads=modelargs[1], n_layers=modelargs[2] // 2,
                                                   dropout=modelargs[5], enable_logging=enable_logging, dtype=dtype)
        self.fusion_scalar = nn.Parameter(torch.tensor([1.0]))  # present in checkpoints, unused in forward
        self.pos_enc_audio = PositionalEncoding(d_model, enable_logging=enable_logging)
        self.pos_enc_video = PositionalEncoding(d_model, enable_logging=enable_logging)
        self.decoder = Decoder(d_model=d_model, vocab_size=vocab_size, enable_logging=enable_logging, dtype=dtype)

        self._cache = Fn.WeightCache(dtype)
        self.decoder._cache = self._cache
        self._seed, self._step = int(seed), 0
        self.overlap_encoders = True        # run the visual frontend on a second stream next to the audio encoder
        # the last encoder layer computes queries / MLP / final norm only for the frames the truncation keeps
        # (identical results; see WhisperEncoder.forward)
        self.prune_unread_frames = True
        self._side = None
        self._aside = None
        self._graphs = {}
        self.max_encoder_graphs = 3             # live encoder recordings (train / validation / test shapes)
        # one recording per encoder, replayed on two streams (AVSR_SPLIT_ENC_GRAPHS=0: both encoders in one recording,
        # whose two branches the runtime largely takes in turn: 7.33 ms per benchmark step against 7.26)
        self.split_encoder_graphs = os.environ.get("AVSR_SPLIT_ENC_GRAPHS", "1") != "0"
        self.video_in_place = os.environ.get("AVSR_VIDEO_IN_PLACE", "1") != "0"
        self._capture_slot = None
        self._seed_delta = None             # device record of a graphed training step (dp.GraphedTail), else None
        self.fusion_module.set_runtime(self._cache, self._dropout_seed, lambda: self._seed_delta)
        if device is not None:
            self.to(device)

    # ---- runtime helpers --------------------------------------------------
    SEED_STEP_MUL = 0xBF58476D1CE4E5B9      # a step further = this much more on every site's seed (mod 2^64)

    def _dropout_seed(self, tag):
        """A distinct 64-bit stream id per (construction seed, forward call, site)."""
        mix = self._seed * 0x9E3779B97F4A7C15 + self._step * self.SEED_STEP_MUL + zlib.crc32(tag.encode())
        return mix & 0xFFFFFFFFFFFFFFFF

    @property
    def _side_stream(self):
        dev = torch.cuda.current_device()
        if self._side is None or self._side[0] != dev:
            self._side = (dev, torch.cuda.Stream(device=dev))
        return self._side[1]

    def make_padding_mask(self, lengths, max_length):
        """True = padded position (models/av_net.py:111-114)."""
        return (torch.arange(max_length, device=lengths.device).expand(lengths.size(0), max_length)
                >= lengths.unsqueeze(1))

    # ---- the two frozen encoders ---------------------------------------------
    @property
    def _audio_stream(self):
        dev = torch.cuda.current_device()
        if self._aside is None or self._aside[0] != dev:
            self._aside = (dev, torch.cuda.Stream(device=dev))
        return self._aside[1]

    def encode(self, inputBatch, ahead=False, graph=False):
        """Runs the frozen Whisper encoder and the frozen visual frontend on ``inputBatch`` (the 5-tuple of
        ``forward``) and returns an opaque handle that ``forward`` / ``training_step`` accept as ``encoded=``.

        ``graph=True`` replays the encoders from a HIP graph captured on first use for this input shape and
        module mode (about 250 kernel launches on two streams become one graph launch; the frozen encoders take
        no per-step scalar arguments, so the recording stays valid while their weights do).

        Nothing in the encoders depends on the trainable weights, so a training loop can issue the encoders of
        batch i+1 *before* the trainable tail (fusion, head, loss, backward, optimizer) of batch i:
        ``ahead=True`` puts the audio encoder and the visual frontend on two streams of their own, where they
        queue behind the previous batch's encoders and run beside the current batch's tail, whose many small
        kernels leave most of the chip idle."""
        audioBatch, audioMask, videoBatch, videoMask, videoLen = inputBatch
        if not audioBatch.is_cuda:
            raise ops._lib.AvsrHipError("AVNet inputs must be on the GPU: this package has no CPU path")
        if graph:
            return self._encode_graph(inputBatch, ahead)
        # The two frozen encoders are independent until the fusion: the visual frontend runs on a second HIP
        # stream so its many short kernels fill the CUs the audio encoder's GEMM tails leave idle (and vice versa).
        main = torch.cuda.current_stream()
        side = self._side_stream if (self.overlap_encoders or ahead) else None
        audio = self._audio_stream if ahead else main
        if side is not None:
            side.wait_stream(main)                                          # the inputs are ready on `main`
            with torch.cuda.stream(side):
                vis = self._run_visual(inputBatch)                          # [B,Tv,2048]
        if ahead:
            audio.wait_stream(main)
        with torch.cuda.stream(audio):
            enc = self._run_audio(inputBatch)
        if side is None:
            vis = self._run_visual(inputBatch)
        done = []
        if side is not None:
            done.append((side, side.record_event()))
        if ahead:
            done.append((audio, audio.record_event()))
        return {"enc": enc, "vis": vis, "done": done, "inputs": inputBatch}

    def _run_visual(self, inputBatch):
        return self.visual_model(inputBatch[2], inputBatch[4], video_slot=self._capture_slot)

    @staticmethod
    def _video_in_place(videoBatch):
        """The batch as the stem can read it where it lies: fp32, contiguous, 16-byte aligned (else such a copy)."""
        if videoBatch.dtype == torch.float32 and videoBatch.is_contiguous():
            return videoBatch if videoBatch.data_ptr() % 16 == 0 else videoBatch.clone()
        return videoBatch.float().contiguous()

    def _run_audio(self, inputBatch):
        audioBatch, videoBatch = inputBatch[0], inputBatch[2]
        cdt = self.compute_dtype
        # AUDIO: [B,3000,80] is already channels-last, the layout the conv kernel wants
        channels_last = audioBatch.shape[2] == 80 and audioBatch.shape[1] % 3000 == 0
        frames = audioBatch.shape[1] if channels_last else audioBatch.shape[2]
        windows = frames // 3000 if (self.long_form and frames % 3000 == 0) else 1
        if windows > 1:                                                 # k x 30 s: k encoder windows per clip
            if not channels_last:
                audioBatch, channels_last = ops.transpose2d(audioBatch, dtype=cdt), True
            audioBatch = audioBatch.contiguous().view(audioBatch.shape[0] * windows, 3000, 80)
        # only the first min(1500, T_video) encoder frames survive the truncation of models/av_net.py:91-96
        n_video = videoBatch.shape[2] if (videoBatch.dim() == 6 and videoBatch.shape[1] == 1) else videoBatch.shape[1]
        keep = n_video if (self.prune_unread_frames and windows == 1) else None
        enc = self.whisper_model.encoder(audioBatch, attention_mask=None, dtype=cdt, channels_last=channels_last,
                                         rows=keep)[0]
        if windows > 1:
            enc = enc.view(enc.shape[0] // windows, windows * enc.shape[1], enc.shape[2])
        return enc

    def _capture_encoders(self, inputBatch, stream, main):
        """Static input buffers, one eager pass (weight packing, lazily set kernel attributes), then the capture."""
        audioBatch, _, videoBatch, _, videoLen = inputBatch
        # The video (by far the largest input) is not copied into a buffer of the recording: the stem reads its address
        # from a device word that every replay sets to the caller's current batch.
        slot = torch.zeros(1, dtype=torch.int64, device=audioBatch.device) if self.video_in_place else None
        stream.wait_stream(main)
        with torch.cuda.stream(stream):
            train_bn = self.visual_model.training
            pending = self.visual_model._tracked_pending
            running = [b.clone() for b in self.visual_model.buffers()] if train_bn else []
            try:
                while True:
                    video = self._video_in_place(videoBatch) if slot is not None else videoBatch.clone()
                    st_in = (audioBatch.clone(), None, video, None, videoLen.clone())
                    self._capture_slot = slot
                    try:
                        if slot is not None:
                            ops.store_u64(slot, video.data_ptr())
                        self.encode(st_in)
                        break
                    except ops._lib.AvsrHipError:
                        if slot is None:
                            raise
                        slot = None                 # a stem form that cannot take the indirection: fixed input buffer
                stream.synchronize()
                for b, saved in zip(self.visual_model.buffers(), running):  # the warm-up must not count as a step
                    b.copy_(saved)
                if self.split_encoder_graphs:
                    # one recording per encoder, replayed on two streams: two launch queues instead of one
                    g = (torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph())
                    with torch.cuda.graph(g[0], stream=stream, capture_error_mode="thread_local"):
                        enc = self._run_audio(st_in)
                    side = self._side_stream
                    side.wait_stream(stream)
                    with torch.cuda.graph(g[1], stream=side, capture_error_mode="thread_local"):
                        vis = self._run_visual(st_in)
                    stream.wait_stream(side)
                else:
                    g = torch.cuda.CUDAGraph()
                    # thread_local: a communicator's watchdog thread may touch the runtime while this thread records
                    with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
                        out = self.encode(st_in)
                        cur = torch.cuda.current_stream()
                        for _side, ev in out["done"]:                        # join the visual stream into the capture
                            cur.wait_event(ev)
                    enc, vis = out["enc"], out["vis"]
            finally:
                self._capture_slot = None
                self.visual_model._tracked_pending = pending         # neither pass above was a training step
        # the recorded kernels read the packed weight copies of this moment: the recording keeps them alive, so a later
        # re-pack (another module mode) cannot hand their memory to someone else
        keep = (self.visual_model._packed, self.whisper_model.encoder._packed)
        if slot is not None:
            st_in = (st_in[0], None, None, None, st_in[4])           # the first batch's video is not the recording's to hold
        return (g, st_in, enc, vis, keep, slot)

    def _encode_graph(self, inputBatch, ahead):
        audioBatch, _, videoBatch, _, videoLen = inputBatch
        key = (tuple(audioBatch.shape), audioBatch.dtype, tuple(videoBatch.shape), videoBatch.dtype, self.training,
               self.prune_unread_frames,
               self.visual_model.training, self.compute_dtype, self.long_form, self.whisper_model.encoder.attn_fp8,
               sum(p._version for p in self.whisper_model.encoder.parameters()),
               sum(p._version for p in self.visual_model.parameters()), torch.cuda.current_device(),
               # eval mode folds the running statistics into the recording: it is only as fresh as they are
               None if self.visual_model.training else self.visual_model.stats_version())
        main = torch.cuda.current_stream()
        stream = self._audio_stream
        rec = self._graphs.get(key)
        if rec is None:
            try:
                rec = self._capture_encoders(inputBatch, stream, main)
            except RuntimeError as e:       # e.g. a runtime that refuses the capture: keep working, launch by launch
                logger.warning("HIP graph capture of the frozen encoders failed (%s); launching them eagerly", e)
                rec = False
            # a recording's pool holds every encoder activation (a few GB): keep the most recent few, so a loop that
            # alternates training and validation (another module mode, often another batch shape) re-records neither
            while len(self._graphs) >= max(1, self.max_encoder_graphs):
                self._graphs.pop(next(iter(self._graphs)))
            self._graphs[key] = rec
        else:
            self._graphs[key] = self._graphs.pop(key)                            # most recently used last
        if rec is False:
            return self.encode(inputBatch, ahead=ahead)
        g, st_in, enc_s, vis_s, _keep, slot = rec
        split = isinstance(g, tuple)
        # the stream the visual recording runs on: its inputs (video or its address, the lengths) are written in THAT
        # stream's order, behind the previous replay that still reads them
        vstream = self._side_stream if split else stream
        video = self._video_in_place(videoBatch) if slot is not None else None
        if self.visual_model.training:
            self.visual_model._stats_epoch += 1                                  # the replay updates the running statistics
            self.visual_model._tracked_pending += 1
        stream.wait_stream(main)                                                 # the inputs are ready on `main`
        if split:
            vstream.wait_stream(main)

        def visual_inputs():
            if slot is None:
                st_in[4].copy_(videoLen, non_blocking=True)
                st_in[2].copy_(videoBatch, non_blocking=True)
            else:                                             # the video's address and the lengths in one small launch
                if videoLen.is_cuda and videoLen.dtype == torch.int64 and videoLen.is_contiguous() and st_in[4].is_contiguous():
                    ops.store_u64(slot, video.data_ptr(), copy=(st_in[4], videoLen))
                else:
                    st_in[4].copy_(videoLen, non_blocking=True)
                    ops.store_u64(slot, video.data_ptr())
                video.record_stream(vstream)

        with torch.cuda.stream(stream):
            st_in[0].copy_(audioBatch, non_blocking=True)
            if split:
                g[0].replay()
                enc = enc_s.clone()                           # the next replay overwrites the recording's outputs
            else:
                visual_inputs()
                g.replay()
                enc, vis = enc_s.clone(), vis_s.clone()
            done = [(stream, stream.record_event())]
        if split:
            with torch.cuda.stream(vstream):
                visual_inputs()
                g[1].replay()
                vis = vis_s.clone()
                done.append((vstream, vstream.record_event()))
        if not ahead:
            for _, e in done:
                main.wait_event(e)
        return {"enc": enc, "vis": vis, "done": done, "inputs": inputBatch}

    # ---- the shared trunk of forward() and training_step() -----------------
    def _features(self, inputBatch, encoded=None):
        audioBatch, audioMask, videoBatch, videoMask, videoLen = inputBatch
        self._step += 1
        if encoded is None:
            encoded = self.encode(inputBatch)
        elif encoded["inputs"][0] is not audioBatch or encoded["inputs"][2] is not videoBatch:
            raise ValueError("`encoded` was computed for a different batch")
        main = torch.cuda.current_stream()
        enc, vis = encoded["enc"], encoded["vis"]
        for stream, ev in encoded["done"]:
            main.wait_event(ev)
        if not torch.cuda.is_current_stream_capturing():
            for tns in (enc, vis):
                tns.record_stream(main)
        ready = getattr(self, "_weights_ready", None)     # compute-dtype weight copies rebuilt on another stream
        if ready is not None:
            main.wait_event(ready)
            self._weights_ready = None
        B, Ta, D = enc.shape
        Tv = vis.shape[1]
        min_len = min(Ta, Tv)
        # audio_proj / audio_ln / positional add are row-wise, so truncating to min_len first
        # gives the same rows as the reference's project-then-slice (models/av_net.py:83-94)
        a_in = enc[:, :min_len].reshape(B * min_len, D) if min_len < Ta else enc.reshape(B * Ta, D)
        a = Fn.linear(a_in, self.audio_proj.weight, self.audio_proj.bias, self._cache.get(self.audio_proj.weight))
        a = self.audio_ln(a, self.pos_enc_audio.table(min_len)).view(B, min_len, -1)
        v_in = vis[:, :min_len].reshape(B * min_len, 2048) if min_len < Tv else vis.reshape(B * Tv, 2048)
        v = Fn.linear(v_in, self.video_proj.weight, self.video_proj.bias, self._cache.get(self.video_proj.weight))
        v = self.video_ln(v, self.pos_enc_video.table(min_len)).view(B, min_len, -1)
        # videoLen.clamp_(max=min_len) in place, as the reference does, and make_padding_mask, in one launch
        if videoLen.dtype == torch.int64 and videoLen.is_contiguous():
            pad_mask = ops.length_mask(videoLen, min_len)
        else:
            videoLen.clamp_(max=min_len)
            pad_mask = self.make_padding_mask(videoLen, min_len)
        # a and v feed the fusion's input projections AND the sum below: the sum's backward parks their gradient and
        # the projections' dX products add it in their epilogues (no separate accumulation kernels)
        same_dtype = a.dtype == self._cache.dtype and v.dtype == self._cache.dtype
        skip = Fn.SkipGrad() if (torch.is_grad_enabled() and a.requires_grad and v.requires_grad and same_dtype) else None
        fused = self.fusion_module(a, v, audio_mask=audioMask[:, :min_len] if audioMask is not None else None,
                                   video_mask=pad_mask, _skip=skip)
        d = a.shape[-1]
        out = Fn.Add3Fn.apply(fused.reshape(B * min_len, d), a.reshape(B * min_len, d), v.reshape(B * min_len, d), skip)
        return out, B, min_len

    def forward(self, inputBatch, encoded=None):
        """inputBatch = (audio [B,3000,80] or [B,80,3000], audio_mask, video [B,T,3,H,W],
        video_mask, video_len int64 [B]) -> fp32 logits [B, min(1500,T), vocab].
        ``encoded``: the handle :meth:`encode` returned for this batch (encoders already issued)."""
        out, B, T = self._features(inputBatch, encoded)
        lin = self.decoder.linear
        wc, wct = self._cache.get(lin.weight)
        logits = Fn.LinearFn.apply(out, lin.weight, lin.bias, wc, wct, torch.float32, wct.shape[1],
                                   (lin.weight, lin.bias, None, 1))
        return logits.view(B, T, -1)

    @staticmethod
    def batch_inputs(batch):
        """The 5-tuple ``forward`` takes, from the data module's batch dict (datamodule/data_module.py:142-152)."""
        return (batch["audio"], batch["audio_mask"], batch["video"], batch["video_mask"], batch["video_lengths"])

    def training_step(self, batch, label_smoothing=0.1, encoded=None):
        """Counterpart of ``AVSRModule.training_step`` (train.py:143-170): forward,
        CTC + label-smoothed CE with the head, log-softmax and both losses fused.
        Returns (loss, ctc, ce, logits).  ``encoded``: see :meth:`encode`."""
        inp = self.batch_inputs(batch)
        out, B, T = self._features(inp, encoded)
        lin = self.decoder.linear
        wc, wct = self._cache.get(lin.weight)
        ctc, ce, logits, total = Fn.HeadLossFn.apply(out, lin.weight, lin.bias, wc, wct, batch["target_ids"],
                                                     batch["audio_lengths"], batch["target_lengths"].reshape(-1).contiguous(),
                                                     B, T, label_smoothing)
        return total, ctc, ce, logits             # total = ctc + ce (train.py:165), summed by the loss kernel

    @torch.no_grad()
    def decode_ids(self, logits):
        """``torch.argmax(logits, -1)`` of ``_decode_predictions`` (train.py:137)."""
        return ops.argmax_rows(logits)
