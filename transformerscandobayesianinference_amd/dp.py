"""Pure data parallelism over the synthetic-batch axis: one process per GPU, RCCL all-reduce of the
flat gradient buffer over xGMI.  The reference is single-process (train.py:29);
this is the only collective on the path and it runs once per optimizer step.

Every dataset (column b of x[T,B,F]) is independent, so rank r simply draws batch_size/world
datasets with its own data seed; all ranks share the `single_eval_pos` stream, which makes every
rank's loss a mean over the same number of terms -- the averaged gradient then equals the
single-process global-batch gradient and all ranks do identical work (no stragglers at the
all-reduce).
"""
import os
import random

import numpy as np
import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def world_size():
    return dist.get_world_size() if is_distributed() else 1


def rank():
    return dist.get_rank() if is_distributed() else 0


def init_from_env(backend=None):
    """Initialise torch.distributed from the torchrun environment (RANK / WORLD_SIZE / LOCAL_RANK /
    MASTER_*).  backend defaults to 'nccl' (= RCCL on ROCm) when a GPU is present, else 'gloo'.
    Returns (rank, world_size, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rk = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    # test hooks for a one-GPU box: PFN_DP_BACKEND=gloo with PFN_DP_SINGLE_DEVICE=1 runs every rank on device 0 (RCCL
    # refuses two ranks on one GPU), which exercises the whole multi-rank control path of bench.py / train()
    backend = backend or os.environ.get('PFN_DP_BACKEND')
    if os.environ.get('PFN_DP_SINGLE_DEVICE') == '1':
        local = 0
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rk, world_size=world)
    return rk, world, local


def seed_ranks(base_seed=None):
    """Rank-shared Python `random` stream (drives single_eval_pos), rank-distinct torch / numpy streams
    (drive the prior draws)."""
    if base_seed is None:
        t = torch.tensor([random.getrandbits(31)], dtype=torch.int64)
        if is_distributed():
            if dist.get_backend() == 'nccl':
                t = t.cuda()
            dist.broadcast(t, 0)
        base_seed = int(t.item())
    random.seed(base_seed)
    torch.manual_seed(base_seed + 1 + rank())
    np.random.seed((base_seed + 1 + rank()) % (2 ** 32))
    return base_seed


def local_batch_size(global_batch_size):
    w = world_size()
    assert global_batch_size % w == 0, f'batch_size {global_batch_size} must be divisible by the world size {w}'
    return global_batch_size // w


def all_reduce_gradients(flat_grad):
    """Sum the flat gradient buffer over ranks (one large collective: 56.7 MB f32 at the north-star
    config; reduce-scatter + all-gather inside RCCL drives all 7 xGMI links).  The 1/world factor is
    folded into the fused optimizer (`grad_multiplier`), so no extra pass over the buffer is needed."""
    if is_distributed():
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return flat_grad


class OverlappedGradientReducer:
    """The gradient all-reduce of one optimizer step as TWO collectives, the first overlapped with the backward.

    The flat gradient buffer is in state-dict order: [embedding | layer 0 ... layer L-1 | decoder].  The backward walks the layers
    top-down, and `pfn_stack_backward_split` launches the weight gradients of the top `first_group_layers` layers as soon as the
    data-gradient chain has left them (instead of with everything else at the very end), calling back on the host right behind that
    launch.  The callback records an event on the launching stream; `finish()` then issues
        tail = grad[offset of layer L - first_group_layers :]   on a side stream that waits for those events only, and
        head = grad[: that offset]                              on the caller's stream (after the whole backward),
    so the tail's all-reduce (half of the bytes at first_group_layers = L/2) runs over xGMI while the lower layers' data gradients and
    weight gradients are still being computed; only the head is exposed.  All collectives are issued from the caller's thread, in the
    same order on every rank.  The split is used only when every parameter's gradient comes out of the HIP stack (fused embedding,
    built-in decoder): a PyTorch-side encoder finishes its gradients after the stack and a custom decoder's parameters sit behind the
    stack's in the buffer, so those models reduce the whole buffer in one collective.

    Usage per optimizer step:   reducer.arm(n)  ->  n forward/backward passes (micro-batch streams)  ->  reducer.finish()."""

    def __init__(self, model=None, flat_grad=None, split=None, first_group_layers=None):
        self.model = model
        self.events, self._armed, self.expected = [], False, 0
        self.extra_events = []
        self.overlapped_last_step = False
        self.fallbacks = 0            # armed steps whose early collective could NOT be overlapped (the hook fired fewer times than armed for)
        self._callback_error = None   # an exception raised inside the ctypes host callback (ctypes swallows it): re-raised by finish()
        if model is not None:
            _, flat_grad = model.flat_parameters()
            L = model.nlayers
            ok = L >= 2 and model._fused_embedding() and not model._custom_decoder
            first_group_layers = (L // 2 if first_group_layers is None else first_group_layers) if ok else 0
            split = model.layer_offset(L - first_group_layers) if first_group_layers > 0 else None
            model._first_group_hook = self if first_group_layers > 0 else None
        self.grad = flat_grad
        self.split = split if split and 0 < split < flat_grad.numel() else None
        self.first_group_layers = first_group_layers or 0
        self.comm = torch.cuda.Stream(flat_grad.device) if flat_grad.is_cuda else None

    # ---- called by _StackFunction.backward ----
    def armed(self):
        return self._armed and self.split is not None

    def first_group_launched(self):
        """Host callback of pfn_stack_backward_split, on the thread and stream that run the backward."""
        try:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.grad.device))
            self.events.append(ev)
        except BaseException as e:      # (we are inside a ctypes callback: an exception would be printed and dropped)
            self._callback_error = e

    # ---- called by the training loop ----
    def arm(self, passes=1, wait_for=()):
        """The next `passes` backward passes complete this optimizer step's gradients (not armed: accumulation micro-steps).
        `wait_for`: events the early collective must ALSO wait for -- batches of the same optimizer step still running on other streams (train()'s
        alternating-stream schedule: only the last batch of a step is armed; the batches before it were enqueued earlier on their streams and
        their weight gradients land in the same buffer)."""
        self.events, self._armed, self.expected = [], True, passes
        self.extra_events = list(wait_for)

    def finish(self):
        """All-reduce (sum) the whole buffer; returns after the collectives are ordered before further work of the current stream."""
        was_armed, self._armed = self._armed, False
        # an exception raised inside the ctypes host callback is re-raised AFTER the collectives: a rank that raised before its all_reduce
        # would leave the other ranks waiting in theirs -- the job would hang instead of failing; with the error the step simply does not overlap
        err, self._callback_error = self._callback_error, None
        if not is_distributed():
            self.events = []
            if err is not None:
                raise RuntimeError('the first-group host callback of pfn_stack_backward_split failed') from err
            return self.grad
        works = []
        overlapped = err is None and self.split is not None and self.comm is not None and len(self.events) == self.expected and self.expected > 0
        if was_armed and self.split is not None and self.comm is not None and not overlapped:
            # same collectives, nothing hidden: say so instead of degrading silently (bench.py reports the count; every rank takes the same branch
            # because `expected` and the number of backward passes are rank-uniform)
            self.fallbacks += 1
            if self.fallbacks == 1:
                import warnings
                warnings.warn(f'OverlappedGradientReducer: armed for {self.expected} backward pass(es) but the early weight-gradient launch called back '
                              f'{len(self.events)} time(s); the gradient all-reduce of this step is not overlapped')
        if overlapped:
            with torch.cuda.stream(self.comm):
                for ev in self.events + self.extra_events:
                    self.comm.wait_event(ev)
                works.append(dist.all_reduce(self.grad[self.split:], op=dist.ReduceOp.SUM, async_op=True))
            works.append(dist.all_reduce(self.grad[:self.split], op=dist.ReduceOp.SUM, async_op=True))
        elif self.split is not None:      # same two collectives on every rank even when nothing could overlap (CPU tensors, hook not fired)
            works.append(dist.all_reduce(self.grad[self.split:], op=dist.ReduceOp.SUM, async_op=True))
            works.append(dist.all_reduce(self.grad[:self.split], op=dist.ReduceOp.SUM, async_op=True))
        else:
            works.append(dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, async_op=True))
        for wk in works:
            wk.wait()
        if overlapped:
            self.grad.record_stream(self.comm)
        self.events, self.extra_events = [], []
        self.overlapped_last_step = overlapped
        if err is not None:
            raise RuntimeError('the first-group host callback of pfn_stack_backward_split failed') from err
        return self.grad

    def detach(self):
        """Unhook from the model (end of train()): the model then holds no reference to this object, its stream or its events -- it can be
        pickled / deep-copied / moved, and a later re-flattening cannot leave `self.grad` pointing at a stale buffer."""
        if self.model is not None and getattr(self.model, '_first_group_hook', None) is self:
            self.model._first_group_hook = None
        self.events, self._armed = [], False

    def layout(self):
        n = self.grad.numel()
        return dict(total_bytes=4 * n, overlapped_bytes=4 * (n - self.split) if self.split else 0, exposed_bytes=4 * (self.split or n),
                    first_group_layers=self.first_group_layers)
