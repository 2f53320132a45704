"""Input encoders (reference encoders.py). `Linear` is the one on the hot path: when both the x-
and the y-encoder are plain `nn.Linear`, TransformerModel fuses them into the HIP embedding kernel
(csrc/rowwise.hip: embed_fwd_kernel). Any other encoder module runs in PyTorch and its output is
handed to the stack as a pre-embedded [S,B,E] tensor.
"""
import math

from torch import nn

Linear = nn.Linear  # reference encoders.py:8


class Normalize(nn.Module):
    """(x - mean) / std. The reference *uses* this name in get_normalized_uniform_encoder
    (encoders.py:18) without defining it; defined here so the helper works."""

    def __init__(self, mean, std):
        super().__init__()
        self.mean, self.std = mean, std

    def forward(self, x):
        return (x - self.mean) / self.std


def get_normalized_uniform_encoder(encoder_creator):
    """Wrap an encoder fed with U[0,1] samples so its input has zero mean / unit std (reference encoders.py:10-18)."""
    return lambda in_dim, out_dim: nn.Sequential(Normalize(.5, math.sqrt(1 / 12)), encoder_creator(in_dim, out_dim))


class CanEmb(nn.Embedding):
    """Per-feature categorical embedding, concatenated over features (reference encoders.py:22-30)."""

    def __init__(self, num_features, num_embeddings, embedding_dim, *args, **kwargs):
        assert embedding_dim % num_features == 0
        super().__init__(num_embeddings, embedding_dim // num_features, *args, **kwargs)

    def forward(self, x):
        emb = super().forward(x)
        return emb.view(*emb.shape[:-2], -1)


def get_Canonical(num_classes):
    return lambda num_features, emsize: CanEmb(num_features, num_classes, emsize)
