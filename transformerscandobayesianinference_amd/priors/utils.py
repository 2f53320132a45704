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
            if not fuse_x_y:
                x, y, target_y = get_batch_method_(*args, **kwargs)
                return (x, y), target_y
            # the fused input COPIES y (torch.cat below), so a deferred repair of a failed factorisation -- which rewrites
            # y in place after the draw returned -- would not reach it: check such draws at once
            with undeferred_draw_checks():
                x, y, target_y = get_batch_method_(*args, **kwargs)
            shifted_y = torch.cat([torch.zeros_like(y[:1]), y[:-1]], 0).unsqueeze(-1).float()
            return torch.cat([x, shifted_y], -1), target_y

        def __len__(self):
            return self.num_steps

        def __iter__(self):
            draw = lambda: self.gbm(**self.get_batch_kwargs, fuse_x_y=self.fuse_x_y)
            if getattr(self, 'prefetch', False) and torch.cuda.is_available():
                group = max(1, int(getattr(self, 'prefetch_group', 1)))
                # ... counted in DATASETS for small batches (round 5): the blocked Cholesky is a chain of ~100 dependent launches whatever the batch, so a group
                # of 10 steps x 4 datasets pays it for 40 datasets where 10 x 64 pay it for 640 (the notebooks train at batch_size 4)
                want = getattr(self, 'prefetch_group_datasets', None)
                if want and 'batch_size' in self.get_batch_kwargs and not getattr(self, '_prefetch_group_fixed', False):
                    group = max(group, -(-int(want) // max(1, self.get_batch_kwargs['batch_size'])))
                group = min(group, self.num_steps)
                budget = getattr(self, 'prefetch_bytes_per_dataset', None)
                if budget is not None and 'batch_size' in self.get_batch_kwargs:
                    # sampler workspace of one group (and the pending group holds as much again): keep it inside a fixed share
                    # of the device memory -- the reference's per-step draw must not run out of memory here either
                    per_step = budget(self.get_batch_kwargs) * self.get_batch_kwargs['batch_size']
                    free = torch.cuda.mem_get_info()[0]
                    group = max(1, min(group, int(free * self.prefetch_memory_share // max(per_step, 1))))
                if group > 1 and not self.fuse_x_y and 'batch_size' in self.get_batch_kwargs:
                    return prefetch_on_side_stream(self._draw_group, (self.num_steps + group - 1) // group, self.num_steps, group)
                return prefetch_on_side_stream(lambda n: [draw() for _ in range(n)], self.num_steps, self.num_steps, 1)
            return iter(draw() for _ in range(self.num_steps))

        def _draw_group(self, n):
            """n steps' worth of datasets from ONE call of the prior (datasets are i.i.d., so a draw of n * batch_size
            columns split into n batches is distributed exactly like n draws): the sampler's long chain of small
            dependent kernels is paid once per group instead of once per step."""
            kw = dict(self.get_batch_kwargs)
            B = kw['batch_size']
            kw['batch_size'] = B * n
            (x, y), target = self.gbm(**kw, fuse_x_y=False)
            return [((x[:, i * B:(i + 1) * B], y[:, i * B:(i + 1) * B]), target[:, i * B:(i + 1) * B]) for i in range(n)]

    return DL


# ---- deferred draw checks ---------------------------------------------------------------------------------------------
# A sampler whose failure flag lives on the device (the Cholesky `info` of priors.fast_gp) must not force a host sync
# inside a draw that runs ahead on a side stream.  While a `deferred_draw_checks()` block is open such samplers append
# a callable to the open list instead of checking at once; the prefetching iterator runs the callables after the
# group's event has completed and BEFORE the first batch of the group is handed out.  A callable returns True when it
# had to repair its draw (it then re-ran the sampler, in place, on the current stream).
_open_check_lists = []


class deferred_draw_checks:
    def __enter__(self):
        self.checks = []
        _open_check_lists.append(self.checks)
        return self.checks

    def __exit__(self, *exc):
        _open_check_lists[:] = [c for c in _open_check_lists if c is not self.checks]
        return False


class undeferred_draw_checks:
    """Inside this block `defer_draw_check` declines (returns False): the sampler checks its draw before returning."""

    def __enter__(self):
        _open_check_lists.append(None)

    def __exit__(self, *exc):
        _open_check_lists.pop()
        return False


def defer_draw_check(fn):
    """Queue `fn` if a deferred block is open (returns True), else leave it to the caller (returns False)."""
    if _open_check_lists and _open_check_lists[-1] is not None:
        _open_check_lists[-1].append(fn)
        return True
    return False


def _tensors(obj):
    if torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, (tuple, list)):
        for o in obj:
            yield from _tensors(o)


def prefetch_on_side_stream(draw_group, num_groups, num_steps, group):
    """Generator over `num_steps` batches.  `draw_group(n)` returns a list of n batches; group g + 1 is
    enqueued on a side HIP stream before the first batch of group g is handed out, so the prior sampler -- a
    long chain of small, latency-bound kernels (blocked Cholesky) -- overlaps the training steps of the
    previous group instead of serialising with them (the reference samples synchronously).
    Sharing the GPU with the training kernels stretches that chain (each of its ~100 dependent kernels waits
    for a free CU slot), which is why a group should span several steps: measured on MI355X, one step of
    look-ahead with one draw per step leaves every draw on the critical path.  The consumer's stream waits on
    the group's event; tensors are marked as used by the consumer stream so the caching allocator cannot
    recycle them early."""
    side = torch.cuda.Stream()
    origin = torch.cuda.current_stream()

    def enqueue(g):
        n = min(group, num_steps - g * group)
        side.wait_stream(origin)   # anything the draw reads (hyper-parameters ...) was produced on the caller's stream
        with torch.cuda.stream(side), deferred_draw_checks() as checks:
            batches = draw_group(n)
            done = torch.cuda.Event()
            done.record(side)
        return batches, done, checks

    pending = enqueue(0) if num_groups > 0 else None
    for g in range(num_groups):
        batches, done, checks = pending
        if checks:
            # the group was enqueued a whole group of training steps ago, so its event has normally completed long before
            # the host gets here (the main stream still holds queued steps: the GPU does not idle while the host waits)
            done.synchronize()
            with torch.cuda.stream(side):
                if any([check() for check in checks]):   # a failed factorisation was redrawn with jitter, in place
                    done = torch.cuda.Event()
                    done.record(side)
        pending = enqueue(g + 1) if g + 1 < num_groups else None
        for batch in batches:
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
