"""Data-parallel training runtime: flat parameter / gradient buffers, bucketed
gradient all-reduce over RCCL overlapped with the backward pass, global-norm
clipping and one fused AdamW launch.

This is the MI355X counterpart of what the reference gets implicitly from
Lightning's ``DDPStrategy(find_unused_parameters=True)`` + ``gradient_clip_val``
+ ``torch.optim.AdamW`` (train.py:218-225,312-322).  Differences by design:

* only parameters that can receive a gradient are registered (the reference's
  buckets also carry the 153.6 M never-used Whisper-decoder weights and
  ``fusion_scalar``; AdamW skips ``grad is None`` parameters, so leaving them out
  changes no result and removes 80 % of the all-reduce volume);
* gradients live in ONE flat fp32 buffer cut into a few large buckets ordered
  the way backward produces them (head first), so each bucket is one RCCL
  all-reduce on a side stream that starts while the rest of backward still
  runs; xGMI is a point-to-point mesh, so few large messages beat many small ones;
* clipping needs no extra collective: after the all-reduce every rank holds the
  same averaged gradient and computes the same norm locally.

``dp_mode="rs_ag"`` replaces the all-reduce by its two halves with the optimizer between them: one
reduce-scatter of the flat gradient buffer (every rank receives the summed gradient of ITS 1/world shard), the
global norm from one 4-byte all-reduce of the shards' sums of squares, clip + AdamW on the owned shard only
(the optimizer's read-modify-write pass and both moment buffers shrink by the world size), and one all-gather
of the updated fp32 parameters.  On the xGMI mesh both halves are direct one-hop exchanges that use all seven
links of a GPU at once; the bytes on the links equal a ring all-reduce's.  The parameter trajectory is the
all-reduce mode's (same sums, same optimizer arithmetic; the global norm is summed shard by shard).
"""
import os

import torch
import torch.distributed as dist

from . import ops

# Debug switch: hand gloo the device tensors themselves when ranks share a GPU (see FlatTrainer._reduce_bucket).
GLOO_DEVICE_TENSORS = os.environ.get("AVSR_GLOO_DEVICE_TENSORS", "0") == "1"
# matrices from this many elements on are overwritten by their gradient kernel instead of cleared and accumulated into
# (FlatTrainer.overwrite); AVSR_OVERWRITE_GRAD=0 turns that off
OVERWRITE_MIN_ELEMS = (1 << 22) if os.environ.get("AVSR_OVERWRITE_GRAD", "1") != "0" else (1 << 62)


def trainable_parameters(model):
    """(name, parameter) pairs autograd reaches in AVNet, in registration order."""
    skip = ("whisper_model.", "visual_model.")
    return [(n, p) for n, p in model.named_parameters()
            if p.requires_grad and not n.startswith(skip) and n != "fusion_scalar"]


class FlatTrainer:
    """Owns flat fp32 parameter / gradient / AdamW-moment buffers for the trainable
    tail of an AVNet and runs reduce -> clip -> AdamW on them.

    Works with or without an initialised process group (world size 1: no collective).
    """

    def __init__(self, model, *, lr=1e-3, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.01, max_grad_norm=1.0,
                 bucket_mb=64, process_group=None, overlap=True, force_collective=False, grad_stream=True, fused_step=False,
                 accumulate=1, dp_mode="allreduce"):
        self.model = model
        self.named = trainable_parameters(model)
        if not self.named:
            raise ValueError("model has no trainable parameters")
        dev = self.named[0][1].device
        self.device = dev
        self.on_gpu = dev.type == "cuda"
        self.lr, self.betas, self.eps, self.wd, self.max_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        # force_collective: issue the all-reduces even at world size 1 (a rehearsal of the collective call path)
        self.collective = self.world > 1 or (force_collective and dist.is_available() and dist.is_initialized())
        self.step_count = 0
        if dp_mode not in ("allreduce", "rs_ag"):
            raise ValueError('dp_mode must be "allreduce" or "rs_ag"')
        self.dp_mode = dp_mode
        self.rank = dist.get_rank(process_group) if self.collective else 0
        if dp_mode == "rs_ag" and overlap and self.collective:
            raise ValueError('dp_mode="rs_ag" exchanges the whole buffer after backward: construct with overlap=False')
        # fused_step: the mean over ranks and the next step's zero_grad are folded into the optimizer kernels (one
        # read-modify-write pass and one fill pass over the flat gradient buffer less per step).  Between
        # reduce_gradients() and step() the buffer then holds the SUM over ranks, and after step() it is zero.
        self.fused_step = bool(fused_step)
        self._pre_scale = 1.0
        self._clean = False
        # accumulate = k (accumulate_grad_batches, config.py:42 / train.py:322): k backward passes - use
        # ``backward(loss)``, which divides the loss by k as Lightning does - feed one optimizer step; the buckets are
        # all-reduced during (overlap) or after the k-th pass only
        self.accumulate = int(accumulate)
        if self.accumulate < 1:
            raise ValueError("accumulate must be >= 1")
        self._passes = 0              # backward passes since the last zero_grad
        self._in_backward = False

        # backward reaches parameters roughly in reverse registration order: lay the flat buffer out in
        # that order so buckets fill front to back as backward proceeds
        order = list(reversed(self.named))
        sizes = [p.numel() for _, p in order]
        # 16-byte alignment for every parameter view
        offs, total = [], 0
        for s in sizes:
            offs.append(total)
            total += (s + 3) // 4 * 4
        self.used = total
        self.shard = None
        if dp_mode == "rs_ag":
            # equal shards of whole 16-byte groups; the padding at the end of the buffer stays zero
            parts = self.world if self.collective else 1
            per = (total + 4 * parts - 1) // (4 * parts) * 4
            total = per * parts
            self.shard = (self.rank * per, (self.rank + 1) * per)
        self.flat_param = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        # rs_ag: the optimizer state exists for the owned shard only; the reduce-scatter lands in a buffer of its own
        n_state = total if self.shard is None else self.shard[1] - self.shard[0]
        self.exp_avg = torch.zeros(n_state, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n_state, dtype=torch.float32, device=dev)
        self.grad_shard = torch.zeros(n_state, dtype=torch.float32, device=dev) if self.shard is not None else None
        self.slices = {}
        with torch.no_grad():
            for (n, p), o, s in zip(order, offs, sizes):
                self.flat_param[o:o + s].copy_(p.detach().reshape(-1))
                p.data = self.flat_param[o:o + s].view(p.shape)
                p.grad = self.flat_grad[o:o + s].view(p.shape)
                self.slices[n] = (o, s)
        # buckets: contiguous element ranges of about bucket_mb each
        cap = max(1, int(bucket_mb * 1024 * 1024 // 4))
        self.buckets, start, pending = [], 0, []
        for (n, p), o, s in zip(order, offs, sizes):
            pending.append(p)
            end = o + (s + 3) // 4 * 4
            if end - start >= cap:
                self.buckets.append({"range": (start, end), "params": pending, "left": 0})
                start, pending = end, []
        if pending:
            self.buckets.append({"range": (start, total), "params": pending, "left": 0})
        self._bucket_of = {id(p): b for b in self.buckets for p in b["params"]}
        # Large matrices whose gradient is produced by ONE kernel per step and whose module says so (the vocabulary
        # head, utils.Decoder: 26.5 M of the 38 M elements) need neither the optimizer's clearing write nor the producer's read of the cleared buffer: in the
        # fused mode without accumulation the producer OVERWRITES their range (functional._param_grads_here reads the
        # flag) and the optimizer leaves it as it is - 8 bytes per element and step less.  Their .grad then holds the
        # last step's gradient, not zeros, between step() and the next backward; a step in which the producer did
        # not run clears the range before the norm is taken.
        self.overwrite = []
        if self.fused_step and self.accumulate == 1 and self.shard is None and self.on_gpu:
            single = {id(q) for m in model.modules() if hasattr(m, "single_kernel_gradients")
                      for q in m.single_kernel_gradients()}
            for (n, p), o, s in zip(order, offs, sizes):
                if s >= OVERWRITE_MIN_ELEMS and id(p) in single:
                    p._avsr_overwrite_grad = True
                    p._avsr_grad_written = False
                    self.overwrite.append((p, o, (s + 3) // 4 * 4))
        # optimizer segments: (lo, hi, clear the gradient after use)
        self._segments, pos = [], 0
        for _, o, s in sorted(self.overwrite, key=lambda e: e[1]):
            if o > pos:
                self._segments.append((pos, o, True))
            self._segments.append((o, o + s, False))
            pos = o + s
        if pos < total:
            self._segments.append((pos, total, True))
        self.overlap = overlap and self.collective
        self.comm_stream = torch.cuda.Stream(device=dev) if (self.on_gpu and self.overlap) else None
        self._works = []
        # the backward kernels accumulate straight into the flat gradient buffer (functional.grad_sink);
        # gradients that still arrive through autograd keep the ordinary post-accumulate hook
        # ... on a stream of their own, beside the dX chain of backward
        self.grad_stream = torch.cuda.Stream(device=dev) if (self.on_gpu and grad_stream) else None
        from . import functional as Fn
        Fn.GRAD_STREAM = self.grad_stream
        for _, p in self.named:
            p._avsr_direct_grad = self.on_gpu
            p._avsr_grad_hook = self._on_grad_ready if self.overlap else None
        if self.overlap:
            for _, p in self.named:
                p.register_post_accumulate_grad_hook(self._on_grad_ready)
        self._scratch = torch.zeros(3, dtype=torch.float32, device=dev)   # sumsq, clip scale, norm
        self._arm()

    # ---- gradient reduction -------------------------------------------------
    def _arm(self):
        for b in self.buckets:
            b["left"] = len(b["params"])
            b["launched"] = False
        self._seen = set()
        self._passes = 0
        self._in_backward = False

    def _gloo_on_gpu(self):
        return self.on_gpu and dist.get_backend(self.group) == "gloo"

    def _reduce_bucket(self, b):
        lo, hi = b["range"]
        view = self.flat_grad[lo:hi]
        b["launched"] = True
        if self._gloo_on_gpu() and not GLOO_DEVICE_TENSORS:
            # gloo with ranks that share one GPU (rehearsals of the multi-rank path): exchange a host copy,
            # synchronously.  c10d's gloo backend stages device tensors through streams it takes from the
            # HIGH-PRIORITY pool; with two processes on one GPU a high-priority queue that waits on the same
            # process's backlogged normal-priority queues stalled for good here - the same stall a high-priority
            # tail stream produced in this configuration.  RCCL's stream has normal priority and every rank has
            # its own GPU, so the path below is not affected.
            if self.grad_stream is not None:
                torch.cuda.current_stream().wait_stream(self.grad_stream)
            host = view.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            view.copy_(host)
            return
        if self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            if self.grad_stream is not None:
                self.comm_stream.wait_stream(self.grad_stream)
            with torch.cuda.stream(self.comm_stream):
                w = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            w = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._works.append(w)

    def _backward_done(self):
        self._in_backward = False

    def _on_grad_ready(self, p):
        if not self._in_backward:
            # first gradient of a backward pass: count the pass and have the engine tell us when it ends
            self._in_backward = True
            self._passes += 1
            torch.autograd.Variable._execution_engine.queue_callback(self._backward_done)
            if self._passes > self.accumulate:
                # the buckets of this step are already being all-reduced in place: more gradient added now would
                # race with the collective and never be reduced
                raise RuntimeError(
                    f"backward pass {self._passes} since zero_grad(), but FlatTrainer(accumulate={self.accumulate}) "
                    "all-reduces during pass number `accumulate`; construct it with accumulate=k to add up k "
                    "micro-batches per optimizer step")
        if self._passes < self.accumulate:
            return                                    # not the last micro-batch: accumulate locally, no exchange
        # a parameter can be announced twice in one pass (by the kernels' own notification and by autograd's
        # post-accumulate hook, which also runs for a gradient that was handed back as None): count it once
        if id(p) in self._seen:
            return
        self._seen.add(id(p))
        b = self._bucket_of[id(p)]
        b["left"] -= 1
        if b["left"] == 0:
            self._reduce_bucket(b)

    def backward(self, loss):
        """Back-propagate one micro-batch: ``loss / accumulate`` (Lightning divides the loss by
        accumulate_grad_batches, so the summed gradient is the mean over the micro-batches).  Returns True when
        this was the last micro-batch of the optimizer step (time for reduce_gradients() and step())."""
        done = self._passes if self.overlap else self._manual_passes()
        (loss / self.accumulate if self.accumulate > 1 else loss).backward()
        if not self.overlap:
            self._passes = done + 1
        return self._passes >= self.accumulate

    def _manual_passes(self):
        if self._passes >= self.accumulate:
            raise RuntimeError(f"more than accumulate={self.accumulate} backward passes since zero_grad()")
        return self._passes

    def zero_grad(self):
        self.join_gradients()
        if self._clean:              # the last optimizer step left the buffer zeroed
            self._clean = False
        else:
            self.flat_grad.zero_()
        for n, p in self.named:          # keep .grad pointing into the flat buffer
            if p.grad is None:
                o, s = self.slices[n]
                p.grad = self.flat_grad[o:o + s].view(p.shape)
        self._arm()

    def reduce_gradients(self):
        """Finish the all-reduce of every bucket (launching those the hooks did not) and
        turn the sums into means."""
        self.join_gradients()
        if self.shard is not None:
            self._reduce_scatter()
            return
        if not self.collective:
            return
        if not self.overlap:
            # no hooks: every gradient is in place and the current stream has joined the gradient stream, so the
            # buckets are reduced right here, in stream order (the collective library's own stream does the
            # work; nothing of this object's runs beside it)
            staged = self._gloo_on_gpu() and not GLOO_DEVICE_TENSORS      # see _reduce_bucket
            for b in self.buckets:
                lo, hi = b["range"]
                view = self.flat_grad[lo:hi]
                if staged:      # gloo (CPU rehearsals of the multi-rank path): exchange a host copy
                    host = view.cpu()
                    dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                    view.copy_(host)
                else:
                    dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
            self._to_mean()
            return
        for b in self.buckets:
            if not b["launched"]:                     # unused parameter in the bucket, or fewer passes than `accumulate`
                self._reduce_bucket(b)
        for w in self._works:
            w.wait()
        self._works.clear()
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        self._to_mean()

    def _reduce_scatter(self):
        """rs_ag: the summed gradient of this rank's shard -> ``grad_shard`` (one collective over the whole buffer)."""
        lo, hi = self.shard
        if not self.collective:
            self.grad_shard.copy_(self.flat_grad[lo:hi])
        elif self._gloo_on_gpu() and not GLOO_DEVICE_TENSORS:      # see _reduce_bucket: host copies on a shared GPU
            host, out = self.flat_grad.cpu(), torch.empty(hi - lo)
            dist.reduce_scatter_tensor(out, host, op=dist.ReduceOp.SUM, group=self.group)
            self.grad_shard.copy_(out)
        else:
            dist.reduce_scatter_tensor(self.grad_shard, self.flat_grad, op=dist.ReduceOp.SUM, group=self.group)
        self._to_mean()

    def _all_gather_params(self):
        """rs_ag: every rank's updated shard -> the whole flat parameter buffer on every rank."""
        if not self.collective:
            return
        lo, hi = self.shard
        if self._gloo_on_gpu() and not GLOO_DEVICE_TENSORS:
            host = torch.empty(self.flat_param.numel())
            dist.all_gather_into_tensor(host, self.flat_param[lo:hi].cpu(), group=self.group)
            self.flat_param.copy_(host)
        else:
            mine = self.flat_param[lo:hi]
            if dist.get_backend(self.group) != "nccl":
                mine = mine.clone()           # RCCL gathers in place (input = output + rank * count); others get a copy
            dist.all_gather_into_tensor(self.flat_param, mine, group=self.group)

    def _to_mean(self):
        if self.fused_step:
            self._pre_scale = 1.0 / self.world       # applied by the clip / AdamW kernels
        elif self.world > 1:
            (self.flat_grad if self.shard is None else self.grad_shard).mul_(1.0 / self.world)

    def join_gradients(self):
        """Make the current stream wait for the parameter-gradient kernels queued on ``grad_stream``."""
        if self.grad_stream is not None:
            torch.cuda.current_stream().wait_stream(self.grad_stream)

    # ---- optimizer ----------------------------------------------------------
    def _settle_overwritten(self):
        """A range its producer overwrites holds the previous step's gradient until then: if the producer did not
        run in this step (the loss did not reach the parameter), the gradient is zero - make it so."""
        for p, o, s in self.overwrite:
            if not p._avsr_grad_written:
                self.flat_grad[o:o + s].zero_()
            p._avsr_grad_written = False

    def step(self, lr=None):
        """Global-norm clip (train.py:321) + AdamW (train.py:219-225) on the flat buffers."""
        self.join_gradients()
        self.step_count += 1
        lr = self.lr if lr is None else lr
        scr = self._scratch
        scr.zero_()
        self._settle_overwritten()
        if self.shard is None:
            ops.sumsq_accumulate(self.flat_grad, scr[0:1])
            ops.clip_scale(scr[0:1], self.max_norm, scr[1:2], scr[2:3], pre_scale=self._pre_scale)
            for lo, hi, clear in self._segments:
                ops.adamw_step(self.flat_param[lo:hi], self.flat_grad[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi],
                               self.step_count, lr=lr, betas=self.betas, eps=self.eps, weight_decay=self.wd,
                               grad_scale=scr[1:2], clear_grad=self.fused_step and clear)
            self._clean = self.fused_step
        else:
            # the norm of the whole gradient = the shards' sums of squares added up (4 bytes over the links)
            lo, hi = self.shard
            ops.sumsq_accumulate(self.grad_shard, scr[0:1])
            if self.collective:
                if self._gloo_on_gpu() and not GLOO_DEVICE_TENSORS:
                    host = scr[0:1].cpu()
                    dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                    scr[0:1].copy_(host)
                else:
                    dist.all_reduce(scr[0:1], op=dist.ReduceOp.SUM, group=self.group)
            ops.clip_scale(scr[0:1], self.max_norm, scr[1:2], scr[2:3], pre_scale=self._pre_scale)
            ops.adamw_step(self.flat_param[lo:hi], self.grad_shard, self.exp_avg, self.exp_avg_sq, self.step_count, lr=lr,
                           betas=self.betas, eps=self.eps, weight_decay=self.wd, grad_scale=scr[1:2])
            self._all_gather_params()
            self._clean = False               # the flat gradient buffer still holds this step's local gradients
        self._pre_scale = 1.0
        # parameters changed in place behind torch's back: drop the compute-dtype weight copies
        cache = getattr(self.model, "_cache", None)
        if cache is not None:
            cache.bump()
            if self.grad_stream is not None:
                # rebuild the bf16 W / W^T copies now, beside the next step's frozen encoders; the next zero_grad /
                # backward / step joins this stream, and the forward tail waits for it through `weights_ready`
                cache.refresh(self.grad_stream)
                self.weights_ready = torch.cuda.Event()
                self.weights_ready.record(self.grad_stream)
                self.model._weights_ready = self.weights_ready
        return scr[2:3]

    def grad_norm(self):
        return self._scratch[2:3]


def one_cycle_lr(step, total_steps, max_lr=1e-3, pct_start=0.1, div_factor=25.0, final_div_factor=1e4):
    """torch.optim.lr_scheduler.OneCycleLR(anneal_strategy='linear', three_phase=False) as configured
    at train.py:230-238: the learning rate to use for optimizer step number ``step`` (0-based)."""
    initial, min_lr = max_lr / div_factor, max_lr / div_factor / final_div_factor
    up_end = float(pct_start * total_steps) - 1.0
    down_end = float(total_steps) - 1.0
    if step <= up_end:
        pct = step / up_end if up_end > 0 else 1.0
        return initial + pct * (max_lr - initial)
    pct = (step - up_end) / (down_end - up_end)
    return max_lr + pct * (min_lr - max_lr)


class GraphedTail:
    """The trainable tail of a training step - fusion, vocabulary head, both losses, backward, the gradient
    exchange, global-norm clip, AdamW and the refresh of the compute-dtype weight copies
    (train.py:143-170,218-246) - replayed from HIP graphs instead of being enqueued kernel by kernel: the tail
    is a dependent chain of about 150 short kernels, and enqueuing it costs the host more than the GPU takes to
    run it.

    ``step(batch, encoded)`` = ``trainer.zero_grad(); model.training_step(batch, encoded=encoded);
    loss.backward(); trainer.reduce_gradients(); trainer.step(lr)``.  The first ``warmup`` calls run exactly
    that, eagerly; the next call records two graphs over static copies of the tail's inputs (forward + backward,
    and clip + AdamW + weight refresh; the gradient all-reduce runs between them, outside any graph) and every
    later call replays them.  What changes from step to step - the dropout seeds, the learning rate, AdamW's
    bias corrections - is read by the recorded kernels from a 32-byte device record that one small launch
    rewrites before each replay (``ops.step_params``).

    Requirements: ``FlatTrainer(fused_step=True, overlap=False, grad_stream=False, dp_mode="allreduce")`` (the
    gradient buffer is left zeroed by the optimizer kernel, every kernel of the tail runs on the caller's
    stream), fused cross-attention blocks, and a fixed batch shape.  The returned loss tensors and logits are
    the recording's static buffers: the next call overwrites them.
    """

    def __init__(self, trainer, warmup=2):
        if not (trainer.fused_step and not trainer.overlap and trainer.grad_stream is None and trainer.shard is None):
            raise ValueError("GraphedTail needs FlatTrainer(fused_step=True, overlap=False, grad_stream=False, "
                             'dp_mode="allreduce")')
        if trainer.accumulate != 1:
            raise ValueError("GraphedTail records one backward pass per optimizer step (accumulate=1)")
        self.trainer, self.model = trainer, trainer.model
        self.warmup = int(warmup)
        self.calls = 0
        self.rec = None
        self.failed = None
        self.eager_steps = 0                 # steps after the recording whose shapes did not fit it
        import torch
        self._one = torch.ones((), dtype=torch.float32, device=trainer.device)

    # ---- the two halves, as they are recorded -------------------------------------------------------------
    def _forward_backward(self, batch, encoded):
        loss, ctc, ce, logits = self.model.training_step(batch, encoded=encoded)
        loss.backward(gradient=self._one)         # a resident 1.0: no fill kernel for the seed gradient
        return loss, ctc, ce, logits

    def _optimizer(self, record):
        tr = self.trainer
        scr = tr._scratch                    # [sum of squares (zero on entry), clip factor, norm]
        ops.sumsq_accumulate(tr.flat_grad, scr[0:1])
        ops.clip_scale(scr[0:1], tr.max_norm, scr[1:2], scr[2:3], pre_scale=1.0 / tr.world)
        for i, (lo, hi, clear) in enumerate(tr._segments):
            ops.adamw_step_dev(tr.flat_param[lo:hi], tr.flat_grad[lo:hi], tr.exp_avg[lo:hi], tr.exp_avg_sq[lo:hi], record,
                               betas=tr.betas, eps=tr.eps, weight_decay=tr.wd, grad_scale=scr[1:2], clear_grad=clear,
                               reset=scr[0:1] if i == len(tr._segments) - 1 else None)
        self.model._cache.refresh_inplace()

    def _eager(self, batch, encoded, lr):
        tr = self.trainer
        tr.zero_grad()
        out = self._forward_backward(batch, encoded)
        tr.reduce_gradients()
        tr.step(lr)
        return out

    # ---- recording ----------------------------------------------------------------------------------------
    def _record(self, batch, encoded):
        import torch
        model, tr = self.model, self.trainer
        dev = tr.device
        st = {k: torch.empty_like(batch[k]) for k in ("video_lengths", "target_ids", "audio_lengths", "target_lengths",
                                                       "audio_mask", "video_mask")}
        # the tail never reads the raw audio / video (the encoders did): one-element stand-ins carry the identity check
        st["audio"] = torch.empty((1,), dtype=batch["audio"].dtype, device=dev)
        st["video"] = torch.empty((1,), dtype=batch["video"].dtype, device=dev)
        s_enc = {"enc": torch.empty_like(encoded["enc"]), "vis": torch.empty_like(encoded["vis"]), "done": [],
                 "inputs": model.batch_inputs(st)}
        record = torch.zeros(4, dtype=torch.int64, device=dev)          # the 32-byte step record
        self._fill(st, s_enc, batch, encoded)
        step0, count0 = model._step, tr.step_count
        tr.join_gradients()
        if tr._clean:
            tr._clean = False
        else:
            tr.flat_grad.zero_()
        tr._scratch.zero_()
        model._cache.refresh()                # every compute-dtype weight copy current, allocated outside the recording
        model._seed_delta = record
        torch.cuda.current_stream().synchronize()
        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g1, capture_error_mode="thread_local"):
                outs = self._forward_backward(st, s_enc)
            pool = g1.pool()
            with torch.cuda.graph(g2, pool=pool, capture_error_mode="thread_local"):
                self._optimizer(record)
        finally:
            model._seed_delta = None
            model._step = step0               # the recording consumed a step number without running a step
        self.rec = {"g1": g1, "g2": g2, "st": st, "enc": s_enc, "record": record, "outs": outs, "step_cap": step0 + 1}
        tr.step_count = count0

    def _fits(self, batch, encoded):
        rec = self.rec
        return (all(batch[k].shape == v.shape and batch[k].dtype == v.dtype for k, v in rec["st"].items()
                    if k not in ("audio", "video"))
                and encoded["enc"].shape == rec["enc"]["enc"].shape and encoded["vis"].shape == rec["enc"]["vis"].shape)

    @staticmethod
    def _fill(st, s_enc, batch, encoded):
        for k, dst in st.items():
            if k not in ("audio", "video"):
                dst.copy_(batch[k], non_blocking=True)
        s_enc["enc"].copy_(encoded["enc"], non_blocking=True)
        s_enc["vis"].copy_(encoded["vis"], non_blocking=True)

    # ---- one step -----------------------------------------------------------------------------------------
    def step(self, batch, encoded, lr=None):
        """One optimizer step on ``batch`` whose frozen encoders were issued as ``encoded`` (AVNet.encode).
        Returns (loss, ctc, ce, logits)."""
        import torch
        model, tr = self.model, self.trainer
        self.calls += 1
        if self.rec is None and self.failed is None and self.calls > self.warmup:
            main = torch.cuda.current_stream()
            for stream, ev in encoded["done"]:
                main.wait_event(ev)
            try:
                self._record(batch, encoded)
            except RuntimeError as e:           # a runtime that refuses the capture: keep training, launch by launch
                self.failed = repr(e)
        if self.rec is None:
            return self._eager(batch, encoded, lr)
        rec = self.rec
        if encoded["inputs"][0] is not batch["audio"] or encoded["inputs"][2] is not batch["video"]:
            raise ValueError("`encoded` was computed for a different batch")
        if not self._fits(batch, encoded):
            # another shape than the recorded one (the short last batch of an epoch, a longer transcript): this step
            # runs launch by launch, on the buffers the recording keeps reading
            self.eager_steps += 1
            out = self._eager(batch, encoded, lr)
            self.model._cache.refresh_inplace()       # the weight copies stay where the recorded kernels read them
            tr._scratch.zero_()                       # the recorded optimizer expects its sum of squares cleared
            return out
        main = torch.cuda.current_stream()
        for stream, ev in encoded["done"]:
            main.wait_event(ev)
        for tns in (encoded["enc"], encoded["vis"]):
            tns.record_stream(main)
        self._fill(rec["st"], rec["enc"], batch, encoded)
        model._step += 1
        tr.step_count += 1
        delta = ((model._step - rec["step_cap"]) * model.SEED_STEP_MUL) & 0xFFFFFFFFFFFFFFFF
        ops.step_params(rec["record"], delta, tr.lr if lr is None else lr, tr.betas, tr.step_count)
        rec["g1"].replay()
        tr.reduce_gradients()                 # the all-reduce (world > 1), in stream order between the two graphs
        rec["g2"].replay()
        tr._pre_scale = 1.0
        tr._clean = True
        batch["video_lengths"].copy_(rec["st"]["video_lengths"], non_blocking=True)    # clamped in place, as forward does
        return rec["outs"]
