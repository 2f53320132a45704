"""Benchmark of the PFN training hot path on MI355X (contract: see the task statement / DESIGN.md).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one full training step of BASELINE.json configs[1] on every rank: draw `batch` synthetic
datasets from priors.fast_gp (HIP sampler), forward through the 6-layer PFN (HIP), bar-distribution
NLL (HIP), backward (HIP), [RCCL all-reduce of the flat gradient], clip + Adam (HIP).  The metric is
synthetic datasets / second over all ranks (weak scaling: per-GPU batch fixed).

The JSON line also carries
  roofline     : the dominant kernel of the step, timed live with HIP events on the launch stream,
                 algorithmic FLOPs (mask-aware) / duration vs the dense bf16 MFMA peak;
  step_roofline: the same accounting for the whole step (train(S,sep) = 3 * fwd(S,sep) per dataset);
  cpu_baseline : the CPU oracle (a port of the reference math, torch f32 on the host cores) timed on a
                 bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK = 2.5e15     # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
WORKLOAD = dict(prior='fast_gp', bptt=2000, num_features=18, emsize=512, nhead=4, nhid=1024, nlayers=6, num_bars=1000,
                hyperparameters=dict(noise=1e-4, outputscale=1.0, lengthscale=0.6))


def pairs(S, sep):
    return S * sep + (S - sep)


def fwd_flops(S, sep, nf, E, F, L, n_out):
    """Algorithmic forward FLOPs of one dataset: embeddings, L layers with mask-aware
    attention, decoder on the test rows only."""
    return (2 * S * nf * E + 2 * sep * E
            + L * (6 * S * E * E + 4 * E * pairs(S, sep) + 2 * S * E * E + 4 * S * E * F)
            + (S - sep) * (2 * E * F + 2 * F * n_out))


def train_flops(S, sep, nf, E, F, L, n_out):
    return 3 * fwd_flops(S, sep, nf, E, F, L, n_out)


def build_model(device, precision, w=WORKLOAD):
    from transformerscandobayesianinference_amd import bar_distribution, encoders
    from transformerscandobayesianinference_amd.priors import fast_gp
    from transformerscandobayesianinference_amd.transformer import TransformerModel
    torch.manual_seed(0)
    ys = fast_gp.get_batch(5000, 20, w['num_features'], device=device, hyperparameters=w['hyperparameters'])[1]
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):   # get_bucket_limits prints (reference behaviour); stdout carries the JSON line only
        borders = bar_distribution.get_bucket_limits(w['num_bars'], ys=ys.cpu())
    criterion = bar_distribution.FullSupportBarDistribution(borders)
    model = TransformerModel(encoders.Linear(w['num_features'], w['emsize']), w['num_bars'], w['emsize'], w['nhead'], w['nhid'],
                             w['nlayers'], 0.0, y_encoder=encoders.Linear(1, w['emsize']), precision=precision)
    model.criterion = criterion
    with torch.no_grad():  # random-init weights of the named architecture; un-zero the residual branches so all kernels see real data
        for layer in model.transformer_encoder.layers:
            layer.linear2.weight.normal_(0, 0.02)
            layer.self_attn.out_proj.weight.normal_(0, 0.02)
    return model.to(device)


def time_kernel(fn, iters=10, warm=3):
    """Average duration (s) of fn() with HIP events recorded on the stream the kernels run on."""
    for _ in range(warm):
        fn()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(iters):
        fn()
    stop.record()
    stop.synchronize()
    return start.elapsed_time(stop) / 1e3 / iters


def kernel_breakdown(batch, sep, w=WORKLOAD):
    """Isolated timings of the step's kernel classes at the workload shape (through the single-op C ABI)."""
    from transformerscandobayesianinference_amd import _hip
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import hipops
    dev = torch.device('cuda')
    S, E, F, H, L, O = w['bptt'], w['emsize'], w['nhid'], w['nhead'], w['nlayers'], w['num_bars']
    M, Mt = batch * S, batch * (S - sep)
    bf = torch.bfloat16
    r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(bf)
    out = []

    f32 = lambda *s: torch.randn(*s, device=dev)
    Hh = _hip

    def gemm(name, n, k, flags, count):
        """One encoder GEMM with its REAL epilogue (bias / GELU / residual / output streams), M = batch * bptt rows."""
        A, B_ = r(M, k), r(n, k)
        kw = {}
        if flags & Hh.EPI_BIAS: kw['bias'] = f32(n)
        if flags & Hh.EPI_RESID: kw['resid'] = f32(M, n)
        if flags & (Hh.EPI_GELU_BWD | Hh.EPI_RESID_T): kw['aux'] = r(M, n)
        if flags & Hh.EPI_OUT_F32: kw['out_f32'] = torch.empty(M, n, device=dev)
        if flags & Hh.EPI_OUT_T: kw['out_t'] = torch.empty(M, n, dtype=bf, device=dev)
        if flags & Hh.EPI_OUT2_T: kw['out2_t'] = torch.empty(M, n, dtype=bf, device=dev)
        t = time_kernel(lambda: hipops.gemm_nt(A, B_, flags, Hh.PREC_BF16, **kw))
        out.append(dict(kernel=f'gemm_nt[{name} {M}x{n}x{k}]', rocprof_name=f'gemm_nt_big_kernel<{flags}, 2, 64>', single=True,
                        launches_per_step=count, seconds=t, flops=2.0 * M * n * k))

    def wgrad_group():
        # every weight gradient of the stack in ONE grouped launch (pfn_stack_backward defers them)
        probs = []
        for _ in range(L):
            probs += [(r(M, E), r(M, F), torch.zeros(E, F, device=dev), None), (r(M, F), r(M, E), torch.zeros(F, E, device=dev), torch.zeros(F, device=dev)),
                      (r(M, E), r(M, E), torch.zeros(E, E, device=dev), None), (r(M, 3 * E), r(M, E), torch.zeros(3 * E, E, device=dev), torch.zeros(3 * E, device=dev))]
        t = time_kernel(lambda: hipops.gemm_tn_group(probs, 0), iters=5, warm=2)
        out.append(dict(kernel=f'gemm_tn_group[{4 * L} weight gradients, {M} tokens]', rocprof_name='gemm_tn_big_kernel', single=True,
                        launches_per_step=1, seconds=t, flops=2.0 * M * L * (2 * E * F + 4 * E * E)))

    gemm('qkv', 3 * E, E, Hh.EPI_BIAS | Hh.EPI_OUT_T, L)
    gemm('out_proj + residual', E, E, Hh.EPI_BIAS | Hh.EPI_RESID | Hh.EPI_OUT_F32, L)
    gemm('linear1 + GELU', F, E, Hh.EPI_BIAS | Hh.EPI_GELU | Hh.EPI_OUT_T | Hh.EPI_OUT2_T, L)
    gemm('linear2 + residual', E, F, Hh.EPI_BIAS | Hh.EPI_RESID | Hh.EPI_OUT_F32, L)
    gemm('d(hpre) = dy2.W2 * gelu\'', F, E, Hh.EPI_GELU_BWD | Hh.EPI_OUT_T, L)
    gemm('dx1 = dh.W1 + dy2', E, F, Hh.EPI_RESID_T | Hh.EPI_OUT_F32, L)
    gemm('d(ctx) = dy1.Wo', E, E, Hh.EPI_OUT_T, L)
    gemm('dx = dqkv.Win + dy1', E, 3 * E, Hh.EPI_RESID_T | Hh.EPI_OUT_F32, L)
    wgrad_group()
    qkv = r(batch, S, 3 * E)
    t = time_kernel(lambda: hipops.attention_fwd(qkv, H, sep, _hip.PREC_BF16))
    attn_fl = 4.0 * E * pairs(S, sep) * batch
    out.append(dict(kernel='attn_fwd', rocprof_name='attn_fwd_kernel', single=True, launches_per_step=L, seconds=t, flops=attn_fl))
    ctx, lse = hipops.attention_fwd(qkv, H, sep, _hip.PREC_BF16)
    dctx = r(batch, S, E)
    t = time_kernel(lambda: hipops.attention_bwd(qkv, ctx, lse, dctx, H, sep, _hip.PREC_BF16))
    out.append(dict(kernel='attn_bwd (delta + dq + dv + dk kernels)', rocprof_name='attn_bwd_*', single=False, launches_per_step=L, seconds=t, flops=2.0 * attn_fl))
    for k in out:
        k['tflops'] = k['flops'] / k['seconds'] / 1e12
        k['step_seconds'] = k['seconds'] * k['launches_per_step']
    return out


def usable_cores(cap=32):
    """Host cores this process may really use: affinity mask, cgroup CPU quota, capped (torch's intra-op
    pool degrades badly when it is given more threads than the cgroup grants)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def cpu_baseline(w=WORKLOAD, batch=2, steps=3, warm=1):
    """The CPU oracle (port of the reference math, f32) on the host cores: GP draw + forward + bar NLL +
    backward + clip + Adam at the workload shape, bounded to a few steps."""
    from oracle import pfn_oracle
    from transformerscandobayesianinference_amd import bar_distribution, encoders
    from transformerscandobayesianinference_amd.transformer import TransformerModel
    threads = usable_cores()
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    S, nf = w['bptt'], w['num_features']
    model = TransformerModel(encoders.Linear(nf, w['emsize']), w['num_bars'], w['emsize'], w['nhead'], w['nhid'], w['nlayers'], 0.0,
                             y_encoder=encoders.Linear(1, w['emsize']))
    borders = torch.sort(torch.randn(w['num_bars'] + 1))[0] * 2
    params = {k: v.detach().clone().requires_grad_(True) for k, v in model.named_parameters()}
    opt = torch.optim.Adam(list(params.values()), lr=1e-4)
    sep = 1755
    times = []
    for it in range(warm + steps):
        t0 = time.time()
        x, y, _ = pfn_oracle.get_batch_fast_gp(batch, S, nf, w['hyperparameters'], dtype=torch.float32)
        logits = pfn_oracle.forward(params, x, y, sep, w['nhead'], dtype=torch.float32)
        loss = pfn_oracle.bar_nll(logits.reshape(-1, w['num_bars']), y[sep:].reshape(-1), borders).mean()
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(list(params.values()), 1.0)
        opt.step()
        if it >= warm:
            times.append(time.time() - t0)
    per_step = sum(times) / len(times)
    return dict(value=batch / per_step, unit='datasets/s', cores=threads, kind='port',
                sample=f'{steps} full training steps (GP draw + fwd + bar NLL + bwd + clip + Adam), batch {batch}, bptt {S}, sep {sep}, torch f32 CPU oracle',
                seconds_per_step=per_step)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32, help='datasets per GPU per step')
    ap.add_argument('--streams', type=int, default=2, help='concurrent micro-batches per step (column groups of the batch on separate HIP streams)')
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-breakdown', action='store_true')
    ap.add_argument('--fixed-sep', type=int, default=None, help='use one eval position instead of the weighted sampler')
    args = ap.parse_args()

    from transformerscandobayesianinference_amd import dp
    from transformerscandobayesianinference_amd.optim import FusedClipAdam
    from transformerscandobayesianinference_amd.priors import fast_gp
    from transformerscandobayesianinference_amd.streams import MicroBatchStreams
    from transformerscandobayesianinference_amd.utils import get_weighted_single_eval_pos_sampler
    rank, world, local = dp.init_from_env()
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)
    w = WORKLOAD
    S, nf, E, F, L, O = w['bptt'], w['num_features'], w['emsize'], w['nhid'], w['nlayers'], w['num_bars']

    model = build_model(device, args.precision)
    if world > 1:
        torch.distributed.broadcast(model.flat_parameters()[0], 0)
    model.train()
    opt = FusedClipAdam(model, lr=1e-4, max_grad_norm=1.0)
    opt.grad_multiplier = 1.0 / world
    random.seed(1234)                      # rank-shared eval-position stream
    torch.manual_seed(1234 + rank)         # rank-distinct prior draws
    sampler = get_weighted_single_eval_pos_sampler(S)
    seps = []

    def loader(num_steps):
        # the reference's DataLoader protocol (priors/utils.py); draws run a group of steps ahead on a side stream
        return iter(fast_gp.DataLoader(num_steps=num_steps, batch_size=args.batch, seq_len=S, num_features=nf,
                                       hyperparameters=w['hyperparameters'], device=device))

    micro = MicroBatchStreams(args.streams)

    def step(batches):
        sep = args.fixed_sep if args.fixed_sep is not None else sampler()
        seps.append(sep)
        (x, y), target = next(batches)
        # forward + bar NLL + backward of the batch, as `--streams` concurrent column groups (streams.py)
        losses = micro.forward_backward(model, (x, y), target, sep,
                                        lambda out, tg: model.criterion(out.reshape(-1, O), tg[sep:].reshape(-1)).view(out.shape[0], -1))
        if world > 1:
            dp.all_reduce_gradients(model.flat_parameters()[1])
        opt.step(zero_grad=True)
        return losses.mean()

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    import contextlib, io
    # ONE loader across warm-up and timed steps: a continuous training run.  The sampler works a group of steps
    # ahead on a side stream (priors/utils.py), so the draws consumed by the first timed steps were produced during
    # the warm-up -- and an equal amount of draw work, for the steps after the window, happens inside it (the loader
    # is one look-ahead group longer than warm-up + steps): the timed region carries exactly `steps` steps' worth of
    # prior sampling, forward, loss, backward and optimizer work in steady state.
    group = getattr(fast_gp.DataLoader, 'prefetch_group', 1)
    with contextlib.redirect_stdout(io.StringIO()):   # DataLoader.__init__ prints its kwargs (reference behaviour)
        batches = loader(args.warmup + args.steps + group)
    for _ in range(args.warmup):
        step(batches)
    barrier()
    seps.clear()
    t0 = time.time()
    for _ in range(args.steps):
        loss = step(batches)
    barrier()
    elapsed = time.time() - t0
    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    elapsed = t.item()
    final_loss = loss.item()

    if rank != 0:
        return
    total = args.batch * world * args.steps
    step_flops = sum(train_flops(S, s, nf, E, F, L, O) for s in seps) * args.batch * world
    result = {
        'metric': 'synthetic datasets/sec (GP prior, bptt=2000)', 'value': total / elapsed, 'unit': 'datasets/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'bf16' if args.precision == 'bf16' else 'f32', 'data': 'synthetic',
        'config': {'workload': 'priors.fast_gp, bptt=2000, num_features=18, emsize=512, nhead=4, nhid=1024, nlayers=6, 1000 bars (BASELINE.json configs[1])',
                   'per_gpu_batch': args.batch, 'global_batch': args.batch * world, 'seq_len': S, 'parallelism': f'dp{world}', 'micro_batch_streams': args.streams,
                   'eval_pos': 'weighted sampler(2000)' if args.fixed_sep is None else args.fixed_sep, 'mean_sep': sum(seps) / len(seps),
                   'final_loss': final_loss},
        'step_roofline': {'bound': 'mfma', 'achieved': step_flops / elapsed / 1e12 / world, 'peak': MFMA_BF16_PEAK / 1e12, 'unit': 'TFLOP/s',
                          'frac': step_flops / elapsed / world / MFMA_BF16_PEAK, 'note': 'whole step per GPU, algorithmic mask-aware FLOPs 3*fwd(S,sep)'},
    }
    if world == 1 and not args.no_kernel_breakdown:
        # kernels are launched per micro-batch (column group of the batch, streams.py): time them at THAT shape
        groups = args.streams if (args.streams > 1 and args.batch % args.streams == 0 and args.batch >= 2 * args.streams) else 1
        ks = kernel_breakdown(args.batch // groups, int(round(sum(seps) / len(seps))))
        for k in ks:
            k['launches_per_step'] *= groups
            k['step_seconds'] *= groups
        dom = max((k for k in ks if k['single']), key=lambda k: k['step_seconds'])   # the dominant single kernel of the step
        traffic, traffic_src = None, None
        pmc_path = os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')
        pmc = json.load(open(pmc_path)) if os.path.exists(pmc_path) else {}
        if pmc.get('batch') == args.batch and pmc.get('streams', 1) == args.streams:   # PMC passes of this command (tools/profile_bench.sh)
            hit = [v for k, v in pmc.get('kernels', {}).items() if dom['rocprof_name'].split('<')[0] in k and (('<' not in dom['rocprof_name']) or dom['rocprof_name'].split('<')[1].split(',')[0] + ',' in k or dom['rocprof_name'].split('<')[1].split(',')[0] + '>' in k)]
            if hit:
                traffic = hit[0].get('read_bytes', 0) + hit[0].get('write_bytes', 0)
                traffic_src = f"profiles/r01_pmc_traffic.json ({pmc.get('note', '')})"
        result['roofline'] = {'bound': 'mfma', 'kernel': dom['kernel'], 'rocprof_kernel': dom['rocprof_name'], 'achieved': dom['tflops'],
                              'peak': MFMA_BF16_PEAK / 1e12, 'unit': 'TFLOP/s', 'frac': dom['tflops'] * 1e12 / MFMA_BF16_PEAK,
                              'traffic': traffic, 'traffic_source': traffic_src, 'algorithmic_flops_per_launch': dom['flops'],
                              'avg_launch_us': dom['seconds'] * 1e6, 'launches_per_step': dom['launches_per_step'],
                              'note': 'launch timed alone with HIP events on its stream; inside the step two micro-batch streams and the sampler stream share the GPU, so a '
                                      'rocprofv3 trace of the default command (profiles/r01_bench_kernel_stats.csv) shows this kernel stretched by its co-runners by a few percent -- '
                                      'profiles/r01_bench_streams1_kernel_stats.csv (--batch 16 --streams 1: the same launch shape on one stream) is the trace whose average duration matches'}
        result['kernels'] = [{k: (round(v, 6) if isinstance(v, float) else v) for k, v in kk.items()} for kk in ks]
    if world == 1 and not args.no_cpu_baseline:
        result['cpu_baseline'] = cpu_baseline()
    print(json.dumps(result))


if __name__ == '__main__':
    main()
