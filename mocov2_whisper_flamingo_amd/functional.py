"""``torch.autograd.Function`` wrappers for the trainable tail of AVNet
(projections, LayerNorms, the gated cross-attention fusion, the vocabulary head
and its losses).  Forward and backward both run in the HIP kernels behind
``ops``; autograd only chains them and accumulates parameter gradients.

All activations are 2-D ``[rows, features]`` (or ``[B, T, features]`` for
attention) tensors of the compute dtype; parameter gradients are fp32.
"""
import os

import torch
from torch.autograd import Function

from . import ops


class WeightCache:
    """Compute-dtype copies of trainable parameters, refreshed when the
    parameter's version counter moves (an optimizer step or a state-dict load).
    ``get(p)`` -> (W as [N,K] in cdt, W^T as [K, N rounded up to 8] in cdt)."""

    def __init__(self, dtype):
        self.dtype = dtype
        self._store = {}
        self.epoch = 0

    def bump(self):
        """Invalidate every copy (the weights were updated in place outside autograd)."""
        self.epoch += 1

    def refresh(self, stream=None):
        """Rebuild every cached copy now, on ``stream`` (off the critical path: right after the optimizer step the
        next step's frozen encoders run for milliseconds before the first trainable weight is read).  The caller
        makes the consuming stream wait for ``stream``."""
        params = [ent[3] for ent in self._store.values()]
        if stream is None:
            for p in params:
                self.get(p)
            return
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            for p in params:
                self.get(p)

    def get(self, p):
        key = id(p)
        hit = self._store.get(key)
        ver = (p._version, p.data_ptr(), self.epoch)
        if hit is not None and hit[0] == ver:
            return hit[1], hit[2]
        if hit is not None:                       # the old copies may still be read by kernels in flight on another stream
            for t in (hit[1], hit[2]):
                if t.is_cuda:
                    t.record_stream(torch.cuda.current_stream())
        with torch.no_grad():
            w = p.detach()
            # big vocabulary heads: pad N so the dX product can split its K (= N) 16 ways in 64-element steps
            pad = 1024 if w.shape[0] > 16384 else 8
            if self.dtype == torch.float32:
                wc, wct = w, ops.transpose2d(w, dtype=self.dtype, pad_to=pad)
            else:                      # compute-dtype copy and its transpose in one pass over the fp32 master
                wct, wc = ops.transpose2d(w, dtype=self.dtype, pad_to=pad, with_copy=True)
        self._store[key] = (ver, wc, wct, p)
        return wc, wct

    def refresh_inplace(self):
        """Recompute every cached copy INTO its existing buffers on the current stream, and mark them current.
        For an owner that updates the parameters in place and keeps every reader on this one stream (a training
        step recorded in a HIP graph: the recorded kernels keep reading the same addresses)."""
        with torch.no_grad():
            for key, (ver, wc, wct, p) in list(self._store.items()):
                w = p.detach()
                pad = 1024 if w.shape[0] > 16384 else 8
                if self.dtype == torch.float32:
                    ops.transpose2d(w, dtype=self.dtype, pad_to=pad, out=wct)
                    wc = w
                else:
                    ops.transpose2d(w, dtype=self.dtype, pad_to=pad, with_copy=True, out=wct, out_copy=wc)
                self._store[key] = ((p._version, p.data_ptr(), self.epoch), wc, wct, p)

    def clear(self):
        self._store.clear()


def grad_sink(p, rows=None):
    """fp32 buffer the backward kernels may ACCUMULATE the gradient of parameter ``p`` (or of its row
    range ``rows``) into directly, or None (= hand the gradient back to autograd).  A parameter owner
    opts in with ``p._avsr_direct_grad = True`` and a preallocated fp32 ``p.grad`` (``dp.FlatTrainer``
    does); this removes one temporary, one fill and one add kernel per parameter and step.  Only valid
    under ``loss.backward()`` - ``torch.autograd.grad`` callers must leave the flag unset."""
    if p is None or not getattr(p, "_avsr_direct_grad", False):
        return None
    g = p.grad
    if g is None or g.dtype != torch.float32 or not g.is_contiguous():
        return None
    return g if rows is None else g[rows[0]:rows[1]]


def grad_done(p, parts=1):
    """Tell the parameter's owner that its gradient (``parts`` pieces per step) is complete: stands in
    for the post-accumulate-grad hook autograd would have run."""
    hook = getattr(p, "_avsr_grad_hook", None) if p is not None else None
    if hook is None:
        return
    if parts > 1:
        left = getattr(p, "_avsr_parts_left", parts) - 1
        p._avsr_parts_left = left if left > 0 else parts
        if left > 0:
            return
    hook(p)


# Stream for parameter-gradient work that goes straight into sink buffers: it is off the critical path of
# backward (only dX feeds the next node), so it can run beside it.  Set by the gradient buffers' owner, who
# must make its own stream wait for this one before reading the gradients (dp.FlatTrainer does both).
GRAD_STREAM = None


def _param_grads(dy, x, has_bias, sink):
    """dW = dY^T X and db = column sums of dY: returned, or accumulated into the sink buffers."""
    pw, pb, rows, parts = sink if sink is not None else (None, None, None, 1)
    gs = GRAD_STREAM
    if gs is not None and grad_sink(pw, rows) is not None and (not has_bias or grad_sink(pb, rows) is not None):
        main = torch.cuda.current_stream()
        gs.wait_stream(main)                                # dy and x are complete on the backward stream
        dy.record_stream(gs)
        x.record_stream(gs)
        with torch.cuda.stream(gs):
            return _param_grads_here(dy, x, has_bias, pw, pb, rows, parts)
    return _param_grads_here(dy, x, has_bias, pw, pb, rows, parts)


def _param_grads_here(dy, x, has_bias, pw, pb, rows, parts):
    if ops.gemm_tn_ok(dy, x):                               # bf16: one pass over dY and X as they lie in memory
        gw = grad_sink(pw, rows)
        gb = grad_sink(pb, rows) if has_bias else None
        db = None
        if has_bias and gb is None:
            db = torch.empty(dy.shape[1], dtype=torch.float32, device=dy.device)
        if gw is not None:
            # a parameter whose owner promises ONE gradient kernel per step (dp.FlatTrainer.overwrite) is written, not
            # read-added-written: the buffer then needs no clearing either
            over = rows is None and parts == 1 and getattr(pw, "_avsr_overwrite_grad", False)
            ops.gemm_tn(dy, x, out=gw, accumulate=not over, colsum=gb if gb is not None else db,
                        accumulate_colsum=gb is not None)
            if over:
                pw._avsr_grad_written = True
            dw = None
            grad_done(pw, parts)
        else:
            dw = ops.gemm_tn(dy, x, colsum=gb if gb is not None else db, accumulate_colsum=gb is not None)
        if gb is not None:
            grad_done(pb, parts)
        return dw, db
    dyt = ops.transpose2d(dy, pad_to=8)                     # [N, Mp]
    xt = ops.transpose2d(x, pad_to=8)                       # [K, Mp]
    gw = grad_sink(pw, rows)
    if gw is not None:
        over = rows is None and parts == 1 and getattr(pw, "_avsr_overwrite_grad", False)
        ops.gemm_nt(dyt, xt, out=gw, residual=None if over else gw)
        if over:
            pw._avsr_grad_written = True
        dw = None
    else:
        dw = ops.gemm_nt(dyt, xt, out_dtype=torch.float32)  # [N, K]
    db = None
    if has_bias:
        gb = grad_sink(pb, rows)
        if gb is not None:
            ops.colsum(dy, out=gb, accumulate=True)
        else:
            db = ops.colsum(dy)
    if gw is not None:
        grad_done(pw, parts)
    if has_bias and db is None:
        grad_done(pb, parts)
    return dw, db


def _dx_gemm(dy, wct, extra=None):
    """dX = dY @ W (+ extra).  Few output tiles with a very long reduction (the vocabulary head: M x 512 over
    K = 52k) would leave the chip idle; split K over 16 batched GEMMs and sum the partials."""
    K = dy.shape[1]
    if K >= 16384 and K % (16 * 64) == 0 and dy.shape[0] * wct.shape[0] <= 1 << 20:
        dx = ops.gemm_nt_splitk(dy, wct, 16)
        return dx if extra is None else dx + extra
    if extra is not None and extra.dtype == dy.dtype and extra.is_contiguous() and extra.shape == (dy.shape[0], wct.shape[0]):
        return ops.gemm_nt(dy, wct, residual=extra)              # the sum rides in the product's epilogue
    dx = ops.gemm_nt(dy, wct)
    return dx if extra is None else dx + extra


class LinearFn(Function):
    """y = x W^T + b  (every nn.Linear of the tail).  ``w``/``b`` are (views of)
    the fp32 parameters and only route gradients; ``wc``/``wct`` are the cached
    compute-dtype copies the kernels read."""

    @staticmethod
    def forward(ctx, x, w, b, wc, wct, out_dtype, ld_out, sink=None, skip=None):
        """``sink`` = (weight parameter, bias parameter, row range or None, pieces per step): where the
        parameter gradients may be accumulated directly (see ``grad_sink``).  ``skip`` = (SkipGrad, slot): a
        gradient that reaches ``x`` along another path and is added to dx inside the dX product (see SkipGrad)."""
        N = wc.shape[0]
        ctx.sink = sink
        ctx.skip = skip
        if ld_out is None:
            y = ops.gemm_nt(x, wc, bias=b, out_dtype=out_dtype)
        else:   # padded leading dimension: the result is a [M, N] view of a [M, ld_out] buffer
            buf = torch.empty((x.shape[0], ld_out), dtype=out_dtype or x.dtype, device=x.device)
            y = ops.gemm_nt(x, wc, bias=b, out=buf[:, :N])
        ctx.save_for_backward(x, wct)
        ctx.has_bias = b is not None
        ctx.N = N
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wct = ctx.saved_tensors
        M, N = x.shape[0], ctx.N
        cdt = x.dtype
        Np = wct.shape[1]
        if dy.dtype != cdt or dy.stride(-1) != 1 or (Np != N and dy.stride(0) < Np):
            pad = torch.zeros((M, Np), dtype=cdt, device=x.device) if Np != N else torch.empty((M, N), dtype=cdt,
                                                                                               device=x.device)
            pad[:, :N].copy_(dy)
            dy = pad[:, :N]
        dyk = dy if Np == N else torch.as_strided(dy, (M, Np), (dy.stride(0), 1))   # padding columns are zero
        extra = ctx.skip[0].take(ctx.skip[1]) if ctx.skip is not None else None
        dx = _dx_gemm(dyk, wct, extra) if ctx.needs_input_grad[0] else None
        dw, db = _param_grads(dy, x, ctx.has_bias, ctx.sink)
        return dx, dw, db, None, None, None, None, None, None


class LayerNormFn(Function):
    """LayerNorm in fp32 with an optional additive table (positional encoding)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, add):
        y, mean, rstd = ops.layernorm_fwd(x, gamma, beta, add=add, save_stats=True)
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.params = (gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        pg, pb = ctx.params
        sg, sb = grad_sink(pg), grad_sink(pb)
        direct = sg is not None and sb is not None
        dg = sg if direct else torch.zeros_like(gamma)          # the kernel accumulates into both
        db = sb if direct else torch.zeros_like(gamma)
        dy = dy if dy.stride(-1) == 1 else dy.contiguous()
        dx = ops.layernorm_bwd(dy, x, gamma, mean, rstd, dg, db)
        if direct:
            grad_done(pg)
            grad_done(pb)
            return dx, None, None, None
        return dx, dg, db, None


class GeluFn(Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.gelu_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.gelu_bwd(x, dy.contiguous())


class GatedResidualFn(Function):
    """out = x + tanh(gate) * dropout(y)."""

    @staticmethod
    def forward(ctx, x, y, gate, p_drop, seed):
        ctx.save_for_backward(y, gate)
        ctx.p_drop, ctx.seed = p_drop, seed
        ctx.gate_param = gate
        return ops.gated_residual_fwd(x, y, gate, p_drop, seed)

    @staticmethod
    def backward(ctx, dout):
        y, gate = ctx.saved_tensors
        dout = dout.contiguous()
        sink = grad_sink(ctx.gate_param)
        dgate = sink if sink is not None else torch.zeros_like(gate)     # the kernel accumulates
        dy = ops.gated_residual_bwd(dout, y, gate, dgate, ctx.p_drop, ctx.seed)
        if sink is not None:
            grad_done(ctx.gate_param)
            return dout, dy, None, None, None
        return dout, dy, dgate, None, None


class AttnFn(Function):
    """softmax(q k^T * scale + key mask) v with attention dropout, per head."""

    @staticmethod
    def forward(ctx, q, k, v, key_mask, H, scale, p_drop, seed):
        mfma = q.dtype == torch.bfloat16                    # bf16: MFMA flash kernels; fp32: exact fp32-math kernels
        o, lse = ops.attn_fwd(q, k, v, H, key_mask=key_mask, scale=scale, use_mfma=mfma, save_lse=True, p_drop=p_drop,
                              seed=seed)
        ctx.save_for_backward(q, k, v, o, lse, key_mask)
        ctx.cfg = (H, scale, p_drop, seed)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse, key_mask = ctx.saved_tensors
        H, scale, p_drop, seed = ctx.cfg
        dq, dk, dv = ops.attn_bwd(q, k, v, o, do, lse, H, key_mask=key_mask, scale=scale,
                                  use_mfma=q.dtype == torch.bfloat16, p_drop=p_drop, seed=seed)
        return dq, dk, dv, None, None, None, None, None


class SharedGrad:
    """Gradient buffer of a tensor that feeds several nodes (the fusion's ``xa`` feeds every block): the nodes'
    backward kernels add into ONE buffer and the last of them to run hands it to autograd, instead of autograd
    summing one tensor per consumer with add kernels of its own."""

    def __init__(self, consumers):
        self.consumers = self.left = consumers
        self.buf = None

    def arrived(self, buf):
        """One consumer has added its share into ``buf``; returns the sum once all have, else None."""
        self.buf = buf
        self.left -= 1
        if self.left > 0:
            return None
        self.left, self.buf = self.consumers, None        # ready for another backward over a retained graph
        return buf


class GatedXattnBlockFn(Function):
    """One GatedCrossAttentionBlock (models/gate_cross_attention.py:28-50) as a single autograd node over the
    avsr_gated_xattn_fwd / _bwd entry points.  ``params`` are the block's 14 parameters in the order
    in_proj_weight, in_proj_bias, out_proj.weight, out_proj.bias, ff.0.weight, ff.0.bias, ff.2.weight, ff.2.bias,
    attn_ln.weight, attn_ln.bias, ff_ln.weight, ff_ln.bias, attn_gate, ff_gate - they only route gradients;
    the kernels read the descriptor's compute-dtype copies."""

    @staticmethod
    def forward(ctx, x, xa, key_mask, blk, shared, *params):
        out, saved = ops.gated_xattn_fwd(blk, x, xa, key_mask)
        ctx.save_for_backward(x, xa, key_mask, saved)
        ctx.blk, ctx.shared, ctx.params = blk, shared, params
        return out

    @staticmethod
    def backward(ctx, dout):
        x, xa, key_mask, saved = ctx.saved_tensors
        params = ctx.params
        sinks = [grad_sink(p) for p in params]
        direct = all(s is not None for s in sinks)
        grads = sinks if direct else [torch.zeros(p.shape, dtype=torch.float32, device=x.device) for p in params]
        shared = ctx.shared if xa is not None else None
        dout = dout if dout.is_contiguous() else dout.contiguous()
        dx, dxa = ops.gated_xattn_bwd(ctx.blk, x, xa, key_mask, saved, dout, grads,
                                      dxa=shared.buf if shared is not None else None)
        if shared is not None:                      # hand the summed gradient over once, from the last consumer
            dxa = shared.arrived(dxa)
        if direct:
            for p in params:
                grad_done(p)
            return (dx, dxa, None, None, None) + (None,) * len(params)
        return (dx, dxa, None, None, None) + tuple(grads)


class SkipGrad:
    """Gradient of a tensor that feeds a sum AND a linear layer (AVNet: ``out = fused + a + v`` with ``a`` and ``v``
    also the inputs of the fusion's two projections): instead of autograd adding the two contributions with a kernel
    of its own, the sum's backward parks its gradient here and the linear layer's dX product adds it in its epilogue.
    The sum is the last consumer in forward order, so its backward always runs first."""

    def __init__(self):
        self.slots = {}

    def put(self, slot, grad):
        self.slots[slot] = grad

    def take(self, slot):
        return self.slots.pop(slot, None)


class Add3Fn(Function):
    @staticmethod
    def forward(ctx, a, b, c, skip=None):
        """``skip``: a SkipGrad that takes the gradients of ``b`` and ``c`` (slots "b", "c") instead of autograd."""
        ctx.skip = skip
        return ops.add3(a, b, c)

    @staticmethod
    def backward(ctx, d):
        if ctx.skip is not None:
            ctx.skip.put("b", d)
            ctx.skip.put("c", d)
            return d, None, None, None
        return d, d, d, None


# Off by default: at the benchmark's head (600 x 51865 x 512) the partials cost the product's epilogue 26 us and save the
# loss kernels 31 us - nothing in the step (same-box A/B 7.48 vs 7.48 ms).  AVSR_HEAD_LSE_FROM_PRODUCT=1 turns it on.
HEAD_LSE_FROM_PRODUCT = os.environ.get("AVSR_HEAD_LSE_FROM_PRODUCT", "0") == "1"


class HeadLossFn(Function):
    """Vocabulary head + CTC + label-smoothed CE in one node (models/utils.py:40 and
    train.py:95-128).  Returns (ctc, ce, logits); the logits output is a
    non-differentiable view - gradients reach the head only through the two
    losses, and d(loss)/d(logits) is written once, in the compute dtype, straight
    into the buffer the two backward GEMMs read."""

    @staticmethod
    def forward(ctx, x, w, b, wc, wct, targets, input_lengths, target_lengths, B, T, label_smoothing):
        V = wc.shape[0]
        ctx.sink = (w, b, None, 1)
        ld = wct.shape[1]
        buf = torch.empty((B * T, ld), dtype=torch.float32, device=x.device)
        # optional (bf16): the product's epilogue leaves every row's log-sum-exp partials per 128-column tile, so the
        # loss kernels never stream the 124 MB of logits in the forward
        part = None
        if HEAD_LSE_FROM_PRODUCT and x.dtype == torch.bfloat16:
            part = torch.empty((ops.row_lse_parts(V), B * T, 4), dtype=torch.float32, device=x.device)
        ops.gemm_nt(x, wc, bias=b, out=buf[:, :V], row_part=part)
        logits = buf.view(B, T, ld)[:, :, :V]
        out3, ws = ops.head_loss_fwd(logits, targets, input_lengths, target_lengths, label_smoothing, row_part=part)
        ctx.save_for_backward(x, wct, buf, ws, targets, input_lengths, target_lengths)
        ctx.cfg = (B, T, V, ld, label_smoothing)
        ctx.mark_non_differentiable(logits)
        ctx.set_materialize_grads(False)          # no zero tensors for the outputs nobody differentiates (logits: 124 MB)
        return out3[0], out3[1], logits, out3[3]

    @staticmethod
    def backward(ctx, g_ctc, g_ce, _g_logits, g_total):
        x, wct, buf, ws, targets, input_lengths, target_lengths = ctx.saved_tensors
        B, T, V, ld, ls = ctx.cfg
        logits = buf.view(B, T, ld)[:, :, :V]

        def upstream(g):          # d(loss)/d(ctc) = its own gradient + the total's; usually only the total's arrives
            parts = [t.reshape(1).float() for t in (g, g_total) if t is not None]
            if not parts:
                return torch.zeros(1, device=x.device)
            return (parts[0] if len(parts) == 1 else parts[0] + parts[1]).contiguous()

        gc, ge = upstream(g_ctc), upstream(g_ce)
        dl = ops.head_loss_bwd(logits, targets, input_lengths, target_lengths, ws, x.dtype, label_smoothing=ls,
                               up_ctc=gc, up_ce=ge, ld_out=ld)                       # [B*T, ld], pad columns zero
        dx = _dx_gemm(dl, wct)
        dw, db = _param_grads(dl[:, :V], x, True, ctx.sink)
        return dx, dw, db, None, None, None, None, None, None, None, None


def linear(x, w, b, cache_entry, out_dtype=None, ld_out=None, sink=None, skip=None):
    """``w`` / ``b`` parameters (sink defaults to them) or row slices of parameters (pass ``sink``)."""
    wc, wct = cache_entry
    if sink is None and isinstance(w, torch.nn.Parameter):
        sink = (w, b if isinstance(b, torch.nn.Parameter) else None, None, 1)
    return LinearFn.apply(x, w, b, wc, wct, out_dtype, ld_out, sink, skip)
