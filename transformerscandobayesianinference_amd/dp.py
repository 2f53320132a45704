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
