"""Bayesian linear regression prior (reference priors/ridge.py): x ~ U[0,1)^{T x B x F}, weights m_b ~ N(0, 0.1^2 I),
y = x . m + N(0, noisy_std^2).  SURVEY.md 8(f) row 4: a cheap prior kept as tensor arithmetic on the training device
(three elementwise launches per batch, nothing to fuse); it exists so `train.py ridge` and notebooks that import
`priors.ridge` keep working.  The transformer it feeds runs through the HIP stack like every other prior's.
"""
import time

import torch

from transformerscandobayesianinference_amd.priors.utils import get_batch_to_dataloader
from transformerscandobayesianinference_amd.utils import default_device


@torch.no_grad()
def get_batch(batch_size, seq_len, num_features, noisy_std=.1, device=default_device):
    """Reference :10-16 (which draws on the CPU; here directly on `device`).  Returns (x[T,B,F], y[T,B], y_non_noisy[T,B])."""
    m = torch.randn(batch_size, num_features, device=device) * .1
    x = torch.rand(seq_len, batch_size, num_features, device=device)
    y_non_noisy = (x * m.unsqueeze(0)).sum(-1)
    y = y_non_noisy + torch.randn_like(y_non_noisy) * noisy_std
    return x, y, y_non_noisy


DataLoader = get_batch_to_dataloader(get_batch)
DataLoader.num_outputs = 1


@torch.no_grad()
def evaluate(x, y, y_non_noisy, alpha=0.):
    """The ridge-regression baseline of the reference (:22-34): for every t, fit ridge regression with an intercept
    (sklearn `Ridge(alpha)` semantics: centred data, penalty on the weights only, minimum-norm solution when the
    system is singular) on the first t points of each dataset and score the prediction at point t against the
    noise-free target.  All datasets of the batch are solved together; returns (mean squared error per t with the
    reference's leading 0., seconds)."""
    start = time.time()
    xb, yb, tb = x.transpose(0, 1).double(), y.transpose(0, 1).double(), y_non_noisy.transpose(0, 1).double()
    B, T, F = xb.shape
    eye = torch.eye(F, dtype=xb.dtype, device=xb.device)
    losses = [torch.zeros((), dtype=xb.dtype, device=xb.device)]
    for t in range(1, T):
        xm, ym = xb[:, :t].mean(1, keepdim=True), yb[:, :t].mean(1, keepdim=True)
        xc, yc = xb[:, :t] - xm, yb[:, :t] - ym
        gram = xc.transpose(1, 2) @ xc + alpha * eye
        w = torch.linalg.pinv(gram, hermitian=True) @ (xc.transpose(1, 2) @ yc.unsqueeze(-1))     # [B,F,1]
        pred = ((xb[:, t:t + 1] - xm) @ w).reshape(B) + ym.reshape(B)
        losses.append(((pred - tb[:, t]) ** 2).mean())
    return torch.stack(losses).float().cpu(), time.time() - start
