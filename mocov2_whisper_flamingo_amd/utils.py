"""``PositionalEncoding`` and ``Decoder`` with the reference's constructor
signatures and state-dict keys (models/utils.py:8-44).  The arithmetic runs in
the HIP kernels: the positional table is added inside the LayerNorm kernel that
precedes it in AVNet, and the head is the epilogue-fused GEMM."""
import math

import torch
import torch.nn as nn

from . import functional as Fn
from . import ops


def sinusoid_table(d_model, max_len=5000):
    """pe[p, 2i] = sin(p * exp(-2i ln(1e4)/d)), pe[p, 2i+1] = cos(...)  (models/utils.py:12-18)."""
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    denominator = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * denominator)
    pe[:, 1::2] = torch.cos(position * denominator)
    return pe.unsqueeze(0)


class PositionalEncoding(nn.Module):
    def __init__(self, d_model, max_len=5000, enable_logging=False):
        super().__init__()
        self.enable_logging = enable_logging
        self.register_buffer("pe", sinusoid_table(d_model, max_len))

    def table(self, seq_len):
        """fp32 [seq_len, d] slice for the fused LayerNorm + add kernel."""
        return self.pe[0, :seq_len]

    def forward(self, x):
        """Stand-alone use: x + pe[:, :T] for a [B,T,d] tensor on the GPU (models/utils.py:21-28)."""
        if x.dim() != 3 or x.shape[1] > self.pe.shape[1] or x.shape[2] != self.pe.shape[2]:
            raise ValueError(f"expected [B, T <= {self.pe.shape[1]}, {self.pe.shape[2]}], got {tuple(x.shape)}")
        return _AddTableFn.apply(x, self.pe[0, :x.shape[1]])


class _AddTableFn(torch.autograd.Function):
    """x[b] + table for every batch item; the table is read in place (no expanded copy)."""

    @staticmethod
    def forward(ctx, x, table):
        return ops.add_table(x, table.contiguous())

    @staticmethod
    def backward(ctx, g):
        return g, None


class Decoder(nn.Module):
    """Frame-level vocabulary head: one Linear(d_model, vocab_size) (models/utils.py:30-44)."""

    def __init__(self, d_model, vocab_size, enable_logging=False, dtype=torch.bfloat16):
        super().__init__()
        self.enable_logging = enable_logging
        self.linear = nn.Linear(d_model, vocab_size)
        self._cache = Fn.WeightCache(dtype)         # AVNet replaces it with the cache it shares with the fusion

    def single_kernel_gradients(self):
        """Parameters whose gradient comes from ONE kernel launch per backward pass (LinearFn's weight-gradient product):
        a gradient-buffer owner may let that kernel overwrite the buffer instead of accumulating into a cleared one
        (dp.FlatTrainer.overwrite)."""
        return [self.linear.weight]

    def forward(self, x):
        """x [B, T, d_model] (or [rows, d_model]) on the GPU -> fp32 logits [B, T, vocab] (models/utils.py:36-44).
        The logits live in a buffer whose row stride is padded; the returned tensor is a view of it."""
        lin = self.linear
        cdt = self._cache.dtype
        x2 = x.reshape(-1, x.shape[-1])
        x2 = x2 if x2.dtype == cdt else x2.to(cdt)
        wc, wct = self._cache.get(lin.weight)
        logits = Fn.LinearFn.apply(x2, lin.weight, lin.bias, wc, wct, torch.float32, wct.shape[1],
                                   (lin.weight, lin.bias, None, 1))
        return logits.view(*x.shape[:-1], -1)
