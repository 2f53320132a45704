"""Prior plumbing (reference priors/utils.py): the get_batch -> DataLoader factory (:14-42), the
scalar hyper-parameter samplers (:64-70) and the small tensor helpers used by priors.mlp (:73-100).
Plotting helpers of the reference (:45-62) are out of scope.
"""
import random

import numpy as np
import scipy.stats as stats
import torch
from torch import nn

from transformerscandobayesianinference_amd.priors.prior import PriorDataLoader
from transformerscandobayesianinference_amd.utils import set_locals_in_self


def get_batch_to_dataloader(get_batch_method_):
    """Wrap `get_batch(batch_size, seq_len, num_features, ...) -> (x, y, target_y)` into a loader class."""

    class DL(PriorDataLoader):
        get_batch_method = get_batch_method_

        def __init__(self, num_steps, fuse_x_y=False, **get_batch_kwargs):
            set_locals_in_self(locals())
            # class-level defaults (e.g. DataLoader.num_outputs = 1) apply when the kwarg is absent
            self.num_features = get_batch_kwargs.get('num_features') or self.num_features
            self.num_outputs = get_batch_kwargs.get('num_outputs') or self.num_outputs
            print('DataLoader.__dict__', self.__dict__)

        @staticmethod
        def gbm(*args, fuse_x_y=True, **kwargs):
            x, y, target_y = get_batch_method_(*args, **kwargs)
            if not fuse_x_y:
                return (x, y), target_y
            shifted_y = torch.cat([torch.zeros_like(y[:1]), y[:-1]], 0).unsqueeze(-1).float()
            return torch.cat([x, shifted_y], -1), target_y

        def __len__(self):
            return self.num_steps

        def __iter__(self):
            draw = lambda: self.gbm(**self.get_batch_kwargs, fuse_x_y=self.fuse_x_y)
            if getattr(self, 'prefetch', False) and torch.cuda.is_available():
                return prefetch_on_side_stream(draw, self.num_steps)
            return iter(draw() for _ in range(self.num_steps))

    return DL


def _tensors(obj):
    if torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, (tuple, list)):
        for o in obj:
            yield from _tensors(o)


def prefetch_on_side_stream(draw, num_steps):
    """Generator over `num_steps` batches whose draw for step t+1 is enqueued on a side HIP stream
    before step t is handed out, so the (latency-bound) prior sampler overlaps the training step of
    the previous batch instead of serialising with it (the reference samples synchronously,
    SURVEY.md 3.3).  The consumer's stream waits on the batch's event; tensors are marked as used
    by the consumer stream so the caching allocator cannot recycle them early."""
    side = torch.cuda.Stream()

    def enqueue():
        with torch.cuda.stream(side):
            batch = draw()
            done = torch.cuda.Event()
            done.record(side)
        return batch, done

    pending = enqueue() if num_steps > 0 else None
    for step in range(num_steps):
        batch, done = pending
        pending = enqueue() if step + 1 < num_steps else None
        main = torch.cuda.current_stream()
        main.wait_event(done)
        for t in _tensors(batch):
            if t.is_cuda:
                t.record_stream(main)
        yield batch


def trunc_norm_sampler_f(mu, sigma):
    return lambda: stats.truncnorm((0 - mu) / sigma, (1 - mu) / sigma, loc=mu, scale=sigma).rvs(1)[0]


def beta_sampler_f(a, b):
    return lambda: np.random.beta(a, b)


def gamma_sampler_f(a, b):
    return lambda: np.random.gamma(a, b)


def uniform_sampler_f(a, b):
    return lambda: np.random.uniform(a, b)


def uniform_int_sampler_f(a, b):
    return lambda: np.random.randint(a, b)


def zipf_sampler_f(a, b, c):
    return lambda: min(b + np.random.zipf(a), c)


def scaled_beta_sampler_f(a, b, scale, minimum):
    return lambda: minimum + round(beta_sampler_f(a, b)() * (scale - minimum + 1) - 0.5)


def normalize_data(data):
    """Standardise over the sequence axis (reference :73-78)."""
    return (data - data.mean(0)) / (data.std(0) + .000001)


def normalize_by_used_features_f(x, num_features_used, num_features):
    return x / (num_features_used / num_features)


class Binarize(nn.Module):
    """1 where x exceeds its median (reference :85-91)."""

    def __init__(self, p=0.5):
        super().__init__()
        self.p = p

    def forward(self, x):
        return (x > torch.median(x)).float()


def order_by_y(x, y):
    """Sort rows by (+/-) y and interleave the two halves (reference :94-100)."""
    order = torch.argsort(y if random.randint(0, 1) else -y, dim=0)[:, 0, 0]
    order = order.reshape(2, -1).transpose(0, 1).reshape(-1)
    return x[order], y[order]
