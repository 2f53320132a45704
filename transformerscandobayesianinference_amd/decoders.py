"""Alternative decoder heads (the module surface of the reference's decoders.py).  The default head
(Linear-GELU-Linear, transformer.py:23) runs inside the HIP stack on the test rows only; these two are ordinary
modules applied to the stack's final hidden states -- no BASELINE configuration selects them.  Sub-module names
(`linear`, `linear1`, `linear2`, `mapper`, `T`) are the reference's, so its checkpoints load.
"""
import random

import torch
from torch import nn
from torch.nn import functional as F

_TEMPERATURE_GRID = (1., 1.4, 1.7, 2., 5., 10., 20., 40., 80., 160.)


class ScaledDecoder(nn.Module):
    """Logits over a learned temperature: a third head produces softmax weights over a fixed grid of temperatures
    and the logits are divided by their weighted mean (reference :6-20)."""

    def __init__(self, ninp, nhid, nout):
        super().__init__()
        self.linear = nn.Linear(ninp, nhid)                       # trunk
        self.linear1 = nn.Linear(nhid, nout)                      # logits
        self.linear2 = nn.Linear(nhid, len(_TEMPERATURE_GRID))    # temperature mixture

    def forward(self, x):
        trunk = F.gelu(self.linear(x))
        grid = torch.tensor(_TEMPERATURE_GRID, device=x.device)
        temperature = torch.softmax(self.linear2(trunk), dim=-1) @ grid
        random.random()   # the reference draws one number per call here (for an occasional debug print): same `random` stream
        return self.linear1(trunk) / temperature.unsqueeze(-1)


class FixedScaledDecoder(nn.Module):
    """The default MLP head divided by ONE learned scalar, stored as the sum of a 10000-vector initialised to
    1 / 10000 each (reference :22-29)."""

    def __init__(self, ninp, nhid, nout):
        super().__init__()
        self.mapper = nn.Sequential(nn.Linear(ninp, nhid), nn.GELU(), nn.Linear(nhid, nout))
        self.T = nn.Parameter(torch.full((10000,), 1e-4))

    def forward(self, x):
        return self.mapper(x) / self.T.sum()
