"""Positional encodings (the module surface of the reference's positional_encodings.py).

Contract shared by all four: `cls(d_model, max_len)` and `module(x[S, B, E]) -> x + table[:S, None, :]`.
Every BASELINE configuration is permutation invariant and uses `NoPositionalEncoding` (train.py:42) -- a no-op, so
the HIP embedding kernel feeds the encoder stack directly.  The other three add a [S, 1, E] table to the embedding
with one broadcast add before it enters the HIP stack.  Buffer / parameter names (`pe`, `positional_embeddings`) and
the order in which the random generator is consumed (one normal draw at construction, one `randperm` per scrambled
forward) are those of the reference, so its checkpoints and seeds carry over.
"""
import math

import torch
from torch import nn


class _AdditiveTable(nn.Module):
    """x + rows(S)[:, None, :] for a subclass-defined [S, E] table."""

    def rows(self, seq_len):
        raise NotImplementedError

    def forward(self, x):
        return self.rows(x.shape[0]).unsqueeze(1) + x


class NoPositionalEncoding(nn.Module):
    def __init__(self, d_model, max_len=None):
        super().__init__()

    def forward(self, x):
        return x


class PositionalEncoding(_AdditiveTable):
    """The fixed sinusoid table of 'Attention is all you need' (reference :21-34): even columns sin, odd columns cos
    of position / 10000^(column / d_model).  Stored as buffer `pe` of shape [max_len, 1, d_model]."""

    def __init__(self, d_model, max_len=5000):
        super().__init__()
        inv_wavelength = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        angle = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1) * inv_wavelength       # [max_len, d_model / 2]
        table = torch.stack([torch.sin(angle), torch.cos(angle)], dim=-1).reshape(max_len, -1)   # interleave sin / cos
        self.register_buffer('pe', table[:, :d_model].unsqueeze(1).contiguous())

    def rows(self, seq_len):
        return self.pe[:seq_len, 0]


class LearnedPositionalEncoding(_AdditiveTable):
    """One trainable vector per position, N(0, 1 / d_model) at construction (reference :37-49)."""

    def __init__(self, d_model, max_len=5000):
        super().__init__()
        self.max_seq_len = max_len
        self.positional_embeddings = nn.Parameter(torch.empty(max_len, d_model).normal_(mean=0, std=d_model ** -0.5))

    def rows(self, seq_len):
        assert seq_len <= self.positional_embeddings.shape[0], 'seq_len can be at most max_len.'
        return self.positional_embeddings[:seq_len]


class PairedScrambledPositionalEncodings(LearnedPositionalEncoding):
    """The learned table read through a fresh random permutation on every call (reference :52-62).  What is permuted
    are the rows of the table viewed as [max_len, d_model / 2, 2] -- the reference's "pairs" -- so whole position
    vectors move; one permutation serves the full batch."""

    def rows(self, seq_len):
        table = self.positional_embeddings
        assert seq_len <= table.shape[0], 'seq_len can be at most max_len.'
        assert table.shape[0] % 2 == 0, 'Please specify an even max_len.'
        grouped = table.view(table.shape[0], -1, 2)
        order = torch.randperm(grouped.shape[0])
        return grouped[order].view_as(table)[:seq_len]
