"""Positional encodings (reference positional_encodings.py).

Protocol: __init__(d_model, max_len=...), forward(x[S,B,E]) -> x + pe.  The permutation-invariant
PFN setup of every BASELINE config uses NoPositionalEncoding (train.py:42); that case is a no-op and
the HIP embedding kernel feeds the stack directly.  The others are PyTorch plumbing applied to
the embedding before it enters the HIP stack.
"""
import math

import torch
from torch import nn


class NoPositionalEncoding(nn.Module):
    def __init__(self, d_model, max_len=None):
        super().__init__()

    def forward(self, x):
        return x


class PositionalEncoding(nn.Module):
    """Fixed sinusoids (reference positional_encodings.py:21-34)."""

    def __init__(self, d_model, max_len=5000):
        super().__init__()
        pos = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        freq = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        table = torch.zeros(max_len, d_model)
        table[:, 0::2] = torch.sin(pos * freq)
        table[:, 1::2] = torch.cos(pos * freq)
        self.register_buffer('pe', table.unsqueeze(1))  # [max_len, 1, d_model]

    def forward(self, x):
        return self.pe[:x.size(0), :] + x


class LearnedPositionalEncoding(nn.Module):
    """One learned vector per position (reference positional_encodings.py:37-49)."""

    def __init__(self, d_model, max_len=5000):
        super().__init__()
        self.max_seq_len = max_len
        self.positional_embeddings = nn.Parameter(torch.empty(max_len, d_model))
        nn.init.normal_(self.positional_embeddings, mean=0, std=d_model ** -0.5)

    def _table(self, seq_len):
        assert seq_len <= len(self.positional_embeddings), 'seq_len can be at most max_len.'
        return self.positional_embeddings[:seq_len]

    def forward(self, x):
        seq_len, bs, d_model = x.shape
        return self._table(seq_len).unsqueeze(1).expand(seq_len, bs, d_model) + x


class PairedScrambledPositionalEncodings(LearnedPositionalEncoding):
    """Learned table whose consecutive pairs are randomly permuted on every call (reference
    positional_encodings.py:52-62)."""

    def forward(self, x):
        seq_len, bs, d_model = x.shape
        table = self.positional_embeddings
        assert seq_len <= len(table), 'seq_len can be at most max_len.'
        assert len(table) % 2 == 0, 'Please specify an even max_len.'
        pairs = table.view(len(table), -1, 2)
        scrambled = pairs[torch.randperm(len(pairs))].view(*table.shape)[:seq_len]
        return scrambled.unsqueeze(1).expand(seq_len, bs, d_model) + x
