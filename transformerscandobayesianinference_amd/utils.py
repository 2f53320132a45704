"""Host-side helpers of the PFN training loop: LR schedules, eval-position samplers, SeqBN.

API parity with the reference `utils.py` (schedules :10-51, get_openai_lr :54-56, samplers
:59-73, SeqBN :76-86, set_locals_in_self :89-92, default_device :95, StoreDictKeyPair :99-113).
Pure host Python / PyTorch plumbing; nothing here is on the GPU hot path.
"""
import argparse
import ast
import math
import random

import torch
from torch import nn
from torch.optim.lr_scheduler import LambdaLR


def _warmup_factor(step, num_warmup_steps):
    return float(step) / float(max(1, num_warmup_steps))


def get_cosine_schedule_with_warmup(optimizer, num_warmup_steps, num_training_steps, num_cycles=0.5, last_epoch=-1):
    """Linear warm-up 0 -> 1 over `num_warmup_steps`, then a cosine down to 0 (reference utils.py:10-22).

    As in the reference, the factor at step 0 is 0, and `train()` steps the scheduler once per
    epoch -- so the whole first epoch trains with lr == 0."""

    def factor(step):
        if step < num_warmup_steps:
            return _warmup_factor(step, num_warmup_steps)
        done = float(step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * done)))

    return LambdaLR(optimizer, factor, last_epoch)


def get_linear_schedule_with_warmup(optimizer, num_warmup_steps, num_training_steps, last_epoch=-1):
    """Linear warm-up then linear decay to 0 (reference utils.py:25-51)."""

    def factor(step):
        if step < num_warmup_steps:
            return _warmup_factor(step, num_warmup_steps)
        return max(0.0, float(num_training_steps - step) / float(max(1, num_training_steps - num_warmup_steps)))

    return LambdaLR(optimizer, factor, last_epoch)


def get_openai_lr(transformer_model):
    """The scaling-law learning rate used when `lr=None` (reference utils.py:54-56)."""
    n = sum(p.numel() for p in transformer_model.parameters())
    return 0.003239 - 0.0001395 * math.log(n)


def get_weighted_single_eval_pos_sampler(max_len):
    """P(i) proportional to 1/(max_len - i) for i in [0, max_len) (reference utils.py:59-65)."""
    population = range(max_len)
    weights = [1 / (max_len - i) for i in population]
    return lambda: random.choices(population, weights)[0]


def get_uniform_single_eval_pos_sampler(max_len):
    """Uniform over [0, max_len) (reference utils.py:68-73)."""
    population = range(max_len)
    return lambda: random.choices(population)[0]


class SeqBN(nn.Module):
    """BatchNorm1d over all (position, batch) tokens (reference utils.py:76-86). PyTorch plumbing:
    no BASELINE config enables `input_normalization`."""

    def __init__(self, d_model):
        super().__init__()
        self.bn = nn.BatchNorm1d(d_model)
        self.d_model = d_model

    def forward(self, x):
        assert self.d_model == x.shape[-1]
        return self.bn(x.reshape(-1, self.d_model)).view(*x.shape)


def set_locals_in_self(locals_):
    """`set_locals_in_self(locals())` inside __init__ stores every argument on self (reference utils.py:89-92)."""
    obj = locals_['self']
    for name, value in locals_.items():
        if name != 'self':
            setattr(obj, name, value)


default_device = 'cuda:0' if torch.cuda.is_available() else 'cpu:0'


class StoreDictKeyPair(argparse.Action):
    """`--flag K1=V1 K2=V2` -> dict. Values are parsed with ast.literal_eval (the reference uses
    eval(), utils.py:109, which executes arbitrary CLI text)."""

    def __init__(self, option_strings, dest, nargs=None, **kwargs):
        self._nargs = nargs
        super().__init__(option_strings, dest, nargs=nargs, **kwargs)

    def __call__(self, parser, namespace, values, option_string=None):
        parsed = {}
        for item in values:
            key, _, raw = item.partition('=')
            try:
                parsed[key] = ast.literal_eval(raw)
            except (ValueError, SyntaxError):
                parsed[key] = raw
        setattr(namespace, self.dest, parsed)
        print('dict values: {}'.format(parsed))
