"""Optional decoder heads (reference decoders.py). The default decoder (Linear-GELU-Linear,
transformer.py:23) runs in the HIP stack on the test rows only; these two alternatives are kept as
PyTorch modules and are applied to the stack's final hidden states (no BASELINE config uses them).
"""
import random

import torch
from torch import nn


class ScaledDecoder(nn.Module):
    """Logits divided by a learned, softmax-mixed temperature (reference decoders.py:6-20)."""
    TEMPERATURES = (1., 1.4, 1.7, 2., 5., 10., 20., 40., 80., 160.)

    def __init__(self, ninp, nhid, nout):
        super().__init__()
        self.linear = nn.Linear(ninp, nhid)
        self.linear1 = nn.Linear(nhid, nout)
        self.linear2 = nn.Linear(nhid, len(self.TEMPERATURES))

    def forward(self, x):
        hidden = nn.functional.gelu(self.linear(x))
        temps = self.linear2(hidden).softmax(-1) @ torch.tensor(self.TEMPERATURES, device=x.device)
        if random.random() > .99:
            print(temps.shape, temps[:, :2])
        return self.linear1(hidden) / temps.unsqueeze(-1)


class FixedScaledDecoder(nn.Module):
    """Default decoder MLP divided by one learned scalar (reference decoders.py:22-29)."""

    def __init__(self, ninp, nhid, nout):
        super().__init__()
        self.mapper = nn.Sequential(nn.Linear(ninp, nhid), nn.GELU(), nn.Linear(nhid, nout))
        self.T = nn.Parameter(torch.ones(10000) / 10000)

    def forward(self, x):
        return self.mapper(x) / self.T.sum()
