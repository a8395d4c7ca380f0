"""``PositionalEncoding`` and ``Decoder`` with the reference's constructor
signatures and state-dict keys (models/utils.py:8-44).  The arithmetic runs in
the HIP kernels: the positional table is added inside the LayerNorm kernel that
precedes it in AVNet, and the head is the epilogue-fused GEMM."""
import math

import torch
import torch.nn as nn

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
        """Stand-alone use: x + pe[:, :T] for a [B,T,d] tensor on the GPU."""
        B, T, d = x.shape
        pe = self.pe[0, :T].to(x.dtype).expand(B, T, d).contiguous()
        return ops.add3(x.contiguous(), pe)


class Decoder(nn.Module):
    """Frame-level vocabulary head: one Linear(d_model, vocab_size)."""

    def __init__(self, d_model, vocab_size, enable_logging=False):
        super().__init__()
        self.enable_logging = enable_logging
        self.linear = nn.Linear(d_model, vocab_size)
