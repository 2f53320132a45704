"""Bar ("Riemann") output distribution of the PFN (reference bar_distribution.py).

`forward` (the training loss, reference :25-33 / :89-108) and `mean` (posterior-predictive mean,
:35-38 / :110-117) run in fused HIP kernels (csrc/bar.hip) through the C ABI; the evaluation-only
helpers (`quantile`, `mode`, `ei`) and the one-off border construction `get_bucket_limits` are
PyTorch plumbing exactly as in the reference.  The loss kernels have no CPU fallback.
"""
import torch
from torch import nn

from transformerscandobayesianinference_amd import _hip


class _BarNLL(torch.autograd.Function):
    """nll[r] = -log p(y[r] | logits[r,:]) with d nll / d logits = softmax(logits) - onehot(bucket)."""

    @staticmethod
    def forward(ctx, logits, y, borders, full_support):
        _hip.require_gpu_tensor(logits, 'logits')
        lib = _hip.lib()
        logits = logits.contiguous().float()
        y = y.contiguous().float().to(logits.device)
        borders = borders.contiguous().float().to(logits.device)   # a criterion left on the CPU must not hand a host pointer to the kernel
        R, nbars = logits.shape
        nll = torch.empty(R, device=logits.device, dtype=torch.float32)
        lse = torch.empty_like(nll)
        bucket = torch.empty(R, device=logits.device, dtype=torch.int32)
        _hip.check(lib.pfn_bar_nll_forward(logits.data_ptr(), nbars, y.data_ptr(), borders.data_ptr(), R, nbars,
                                           int(full_support), nll.data_ptr(), lse.data_ptr(), bucket.data_ptr(),
                                           _hip.stream_ptr(logits.device)), 'pfn_bar_nll_forward')
        ctx.save_for_backward(logits, lse, bucket)
        return nll

    @staticmethod
    def backward(ctx, gout):
        logits, lse, bucket = ctx.saved_tensors
        R, nbars = logits.shape
        gout = gout.contiguous().float()
        dlogits = torch.empty_like(logits)
        _hip.check(_hip.lib().pfn_bar_nll_backward(logits.data_ptr(), nbars, lse.data_ptr(), bucket.data_ptr(),
                                                   gout.data_ptr(), R, nbars, dlogits.data_ptr(),
                                                   _hip.stream_ptr(logits.device)), 'pfn_bar_nll_backward')
        return dlogits, None, None, None


def _bar_mean(logits, borders, full_support):
    _hip.require_gpu_tensor(logits, 'logits')
    shape = logits.shape[:-1]
    flat = logits.detach().reshape(-1, logits.shape[-1]).contiguous().float()
    out = torch.empty(flat.shape[0], device=flat.device, dtype=torch.float32)
    borders = borders.contiguous().float().to(flat.device)
    _hip.check(_hip.lib().pfn_bar_mean(flat.data_ptr(), flat.shape[1], borders.data_ptr(),
                                       flat.shape[0], flat.shape[1], int(full_support), out.data_ptr(),
                                       _hip.stream_ptr(flat.device)), 'pfn_bar_mean')
    return out.view(shape)


class BarDistribution(nn.Module):
    """Piecewise-constant density over sorted `borders` (min, ..., max); bucket k is (b_k, b_{k+1}].

    Buffers `borders` / `bucket_widths` become part of the model state-dict through
    `model.criterion` (reference train.py:45)."""
    _full_support = False

    def __init__(self, borders: torch.Tensor):
        super().__init__()
        assert len(borders.shape) == 1
        self.register_buffer('borders', borders)
        self.register_buffer('bucket_widths', self.borders[1:] - self.borders[:-1])
        span = self.borders[-1] - self.borders[0]
        assert (self.bucket_widths.sum() - span).abs() < 1e-4, f'diff: {self.bucket_widths.sum() - span}'
        assert (torch.argsort(borders) == torch.arange(len(borders), device=borders.device)).all(), "Please provide sorted borders!"
        self.num_bars = len(borders) - 1

    def map_to_bucket_idx(self, y):
        """searchsorted(borders, y) - 1 with both end points mapped inside (reference :19-23)."""
        idx = torch.searchsorted(self.borders, y) - 1
        idx[y == self.borders[0]] = 0
        idx[y == self.borders[-1]] = self.num_bars - 1
        return idx

    def forward(self, logits, y):
        """Negative log density; logits [..., num_bars], y [...] -> [...]. Out-of-support targets give
        NaN (the reference asserts, :27; a device kernel cannot)."""
        assert logits.shape[-1] == self.num_bars, f'{logits.shape[-1]} vs {self.num_bars}'
        nll = _BarNLL.apply(logits.reshape(-1, self.num_bars), y.reshape(-1), self.borders, self._full_support)
        return nll.view(y.shape)

    def bucket_means(self):
        return self.borders[:-1] + self.bucket_widths / 2

    def mean(self, logits):
        return _bar_mean(logits, self.borders, self._full_support)

    def quantile(self, logits, center_prob=.682):
        """Central interval [lower, upper] with mass `center_prob`, linear inside a bucket
        (reference :40-62; vectorised over rows instead of the reference's Python loop)."""
        shape = logits.shape
        probs = logits.reshape(-1, shape[-1]).softmax(-1)
        side = (1 - center_prob) / 2

        def lower(p, borders):
            cum = torch.cumsum(p, -1)
            idx = torch.searchsorted(cum, torch.full_like(cum[:, :1], side)).clamp(0, cum.shape[1] - 1)
            prev = torch.where(idx > 0, cum.gather(1, (idx - 1).clamp(min=0)), cum[:, -1:])  # cum[idx-1], idx=0 wraps as in the reference
            left, right = borders[idx], borders[idx + 1]
            return (left + (right - left) * (side - prev) / p.gather(1, idx)).squeeze(1)

        lo = lower(probs, self.borders)
        hi = lower(probs.flip(-1), self.borders.flip(0))
        return torch.stack([lo, hi], -1).reshape(*shape[:-1], 2).cpu()

    def mode(self, logits):
        """Centre of the most likely bucket (reference :64-67).  The plain centre also for the two half-normal tail
        buckets of the full-support variant -- the reference does not override `mode` there."""
        return BarDistribution.bucket_means(self)[logits.argmax(-1)]

    def ei(self, logits, best_f, maximize=True):
        """Expected improvement over `best_f` under the bar density (reference :69-80)."""
        lo, hi = self.borders[:-1], self.borders[1:]
        best = torch.as_tensor(best_f, dtype=lo.dtype, device=lo.device)
        if maximize:
            contrib = ((hi + torch.maximum(lo, best)) / 2 - best).clamp(min=0)
        else:
            contrib = -((torch.minimum(hi, best) + lo) / 2 - best).clamp(max=0)
        return torch.softmax(logits, -1) @ contrib.to(logits.dtype)


class FullSupportBarDistribution(BarDistribution):
    """Bar distribution whose two outer buckets are half-normal tails (reference :83-117)."""
    _full_support = True

    @staticmethod
    def halfnormal_with_p_weight_before(range_max, p=.5):
        scale = range_max / torch.distributions.HalfNormal(torch.tensor(1.)).icdf(torch.tensor(p))
        return torch.distributions.HalfNormal(scale)

    def forward(self, logits, y):
        assert self.num_bars > 1
        return super().forward(logits, y)

    def bucket_means(self):
        means = super().bucket_means().clone()
        tails = (self.halfnormal_with_p_weight_before(self.bucket_widths[0]),
                 self.halfnormal_with_p_weight_before(self.bucket_widths[-1]))
        means[0] = -tails[0].mean + self.borders[1]
        means[-1] = tails[1].mean + self.borders[-2]
        return means


def get_bucket_limits(num_outputs: int, full_range: tuple = None, ys: torch.Tensor = None):
    """Bucket borders: equal-count quantiles of `ys` (clipped to `full_range` if given) or a uniform
    grid over `full_range` (reference :121-143). One-off host-side setup, not on the hot path."""
    assert (ys is not None) or (full_range is not None)
    if ys is None:
        width = (full_range[1] - full_range[0]) / num_outputs
        limits = torch.cat([full_range[0] + torch.arange(num_outputs).float() * width, torch.tensor(full_range[1]).unsqueeze(0)], 0)
    else:
        ys = ys.flatten()
        extra = len(ys) % num_outputs
        if extra:
            ys = ys[:-extra]
        print(f'Using {len(ys)} y evals to estimate {num_outputs} buckets. Cut off the last {extra} ys.')
        per_bucket = len(ys) // num_outputs
        if full_range is None:
            full_range = (ys.min(), ys.max())
        else:
            assert full_range[0] <= ys.min() and full_range[1] >= ys.max()
            full_range = torch.tensor(full_range)
        ordered = ys.sort(0)[0]
        inner = (ordered[per_bucket - 1::per_bucket][:-1] + ordered[per_bucket::per_bucket]) / 2
        print(full_range)
        limits = torch.cat([full_range[0].unsqueeze(0), inner, full_range[1].unsqueeze(0)], 0)
    assert len(limits) - 1 == num_outputs and full_range[0] == limits[0] and full_range[-1] == limits[-1]
    return limits
