"""Binary-classification priors made from regression priors (the module surface of the reference's
priors/binarized_regression.py): every target becomes a coin flip with success probability sigmoid(y).
The regression draw is the HIP GP sampler of `fast_gp` / `fast_gp_mix`; squashing and coin flips are two elementwise
launches on the same device.
"""
import torch

from transformerscandobayesianinference_amd.priors import fast_gp, fast_gp_mix
from transformerscandobayesianinference_amd.priors.utils import get_batch_to_dataloader


class _Binarized:
    """`get_batch` of a regression prior with labels ~ Bernoulli(sigmoid(y)) in place of y -- and of the target, which
    these priors do not distinguish from y (`assert_on=True` verifies that assumption, reference :8-10)."""

    def __init__(self, regression_get_batch):
        self.regression_get_batch = regression_get_batch

    def __call__(self, *args, assert_on=False, **kwargs):
        x, y, target_y = self.regression_get_batch(*args, **kwargs)
        if assert_on and y is not target_y:
            raise AssertionError('y == target_y is assumed by this function')
        labels = torch.sigmoid(y).bernoulli()
        return x, labels, labels


def regression_prior_to_binary(get_batch_function):
    return _Binarized(get_batch_function)


def _loader(get_batch_function):
    cls = get_batch_to_dataloader(regression_prior_to_binary(get_batch_function))
    cls.num_outputs = 1
    return cls


Binarized_fast_gp_dataloader = _loader(fast_gp.get_batch)
Binarized_fast_gp_mix_dataloader = _loader(fast_gp_mix.get_batch)
