"""PFN training loop on MI355X: same call surface as the reference `train.py` (train() :22-135,
Losses :14-19, CLI :137-287), re-hosted on the HIP stack.

Per step (reference :66-108): draw a synthetic batch from the prior, sample `single_eval_pos`,
forward through TransformerModel (HIP), criterion (HIP for bar distributions), backward (HIP), and
every `aggregate_k_gradients`-th step clip-to-1.0 + Adam (fused HIP).  Under torch.distributed
(one process per GPU) each rank draws batch_size/world datasets and the flat gradient buffer is
all-reduced over RCCL before the optimizer step.

Kept reference behaviours: gradients of the k accumulated micro-batches are
summed, not averaged (Q6); the LR factor of epoch 1 is 0 and the scheduler steps per epoch (Q5);
the model is returned on the CPU (Q8).  Removed: the per-step host syncs `loss.item()` /
`.cpu()` (:101-102) -- the loss statistics are accumulated on the device and read once per epoch.
"""
import argparse
import time
import warnings

import torch
import yaml
from torch import nn

from transformerscandobayesianinference_amd import _hip, dp, encoders, positional_encodings, priors
from transformerscandobayesianinference_amd.bar_distribution import BarDistribution, FullSupportBarDistribution, get_bucket_limits
from transformerscandobayesianinference_amd.optim import FusedClipAdam
from transformerscandobayesianinference_amd.streams import MicroBatchStreams
from transformerscandobayesianinference_amd.transformer import TransformerModel
from transformerscandobayesianinference_amd.utils import (StoreDictKeyPair, get_cosine_schedule_with_warmup, get_openai_lr,
                                                          get_uniform_single_eval_pos_sampler, get_weighted_single_eval_pos_sampler)


class Losses():
    gaussian = nn.GaussianNLLLoss(full=True, reduction='none')
    mse = nn.MSELoss(reduction='none')
    ce = nn.CrossEntropyLoss(reduction='none')
    bce = nn.BCEWithLogitsLoss(reduction='none')
    get_BarDistribution = BarDistribution


def _accepts_kwarg(fn, name):
    """True if `fn` can be called with keyword `name` (named parameter or **kwargs)."""
    import inspect
    if fn is None:
        return False
    try:
        params = inspect.signature(fn).parameters.values()
    except (TypeError, ValueError):
        return False
    return any(p.name == name or p.kind is inspect.Parameter.VAR_KEYWORD for p in params)


def _is_bar(criterion):
    return isinstance(criterion, BarDistribution) or "BarDistribution" in criterion.__class__.__name__


def compute_losses(criterion, output, targets, n_out):
    """Per-(position, dataset) losses for every criterion the reference dispatches on (train.py:78-89)."""
    if isinstance(criterion, nn.GaussianNLLLoss):
        assert output.shape[-1] == 2, 'need to write a little bit of code to handle multiple regression targets at once'
        losses = criterion(output[..., 0].flatten(), targets.flatten(), var=output[..., 1].abs().flatten())
    elif isinstance(criterion, (nn.MSELoss, nn.BCEWithLogitsLoss)):
        losses = criterion(output.flatten(), targets.flatten())
    else:
        losses = criterion(output.reshape(-1, n_out), targets.flatten())
    return losses.view(*output.shape[0:2]).squeeze(-1)


def train(priordataloader_class, criterion, encoder_generator, emsize=200, nhid=200, nlayers=6, nhead=2, dropout=0.2,
          epochs=10, steps_per_epoch=100, batch_size=200, bptt=10, lr=None, warmup_epochs=10, input_normalization=False,
          y_encoder_generator=None, pos_encoder_generator=None, decoder=None, extra_prior_kwargs_dict={},
          scheduler=get_cosine_schedule_with_warmup, load_weights_from_this_state_dict=None, validation_period=10,
          single_eval_pos_gen=None, gpu_device='cuda:0', aggregate_k_gradients=1, verbose=True, precision='fp16', micro_streams=2, epoch_callback=None,
          aggregate_streams=None, deterministic=False, aggregate_stacked=None):
    device = gpu_device if torch.cuda.is_available() else 'cpu:0'
    print(f'Using {device} device')
    if not str(device).startswith('cuda') and getattr(TransformerModel, 'requires_gpu', False):
        raise _hip.HipExtensionError('train(): no GPU is visible; the PFN hot path runs on an MI355X through libpfn_hip.so only '
                                     '(there is no CPU fallback -- the CPU restatement lives in oracle/ and is test infrastructure)')
    world = dp.world_size()
    if world > 1:
        dp.seed_ranks()
    if str(device).startswith('cuda'):
        torch.cuda.set_device(torch.device(device))     # side streams and scratch allocations follow the current device
    prior_kwargs = dict(extra_prior_kwargs_dict)
    if 'device' not in prior_kwargs and _accepts_kwarg(getattr(priordataloader_class, 'get_batch_method', None), 'device'):
        prior_kwargs['device'] = device                   # draw where the model lives (the priors default to cuda:0)
    dl = priordataloader_class(num_steps=steps_per_epoch, batch_size=dp.local_batch_size(batch_size), seq_len=bptt,
                               **prior_kwargs)

    encoder = encoder_generator(dl.num_features + 1 if dl.fuse_x_y else dl.num_features, emsize)
    n_out = dl.num_outputs
    if isinstance(criterion, nn.GaussianNLLLoss):
        n_out *= 2
    elif _is_bar(criterion):
        assert n_out == 1
        n_out = criterion.num_bars
    model = TransformerModel(encoder, n_out, emsize, nhead, nhid, nlayers, dropout,
                             y_encoder=y_encoder_generator(1, emsize), input_normalization=input_normalization,
                             pos_encoder=(pos_encoder_generator or positional_encodings.NoPositionalEncoding)(emsize, bptt * 2),
                             decoder=decoder, precision=precision, deterministic=deterministic)
    model.criterion = criterion
    if load_weights_from_this_state_dict is not None:
        model.load_state_dict(load_weights_from_this_state_dict)
    model.to(device)
    if world > 1:  # identical initial weights on every rank
        flat, _ = model.flat_parameters()
        torch.distributed.broadcast(flat, 0)

    if lr is None:
        lr = get_openai_lr(model)
        print(f"Using OpenAI max lr of {lr}.")
    optimizer = FusedClipAdam(model, lr=lr, max_grad_norm=1.)
    optimizer.grad_multiplier = 1.0 / world
    scheduler = scheduler(optimizer, warmup_epochs, epochs)
    micro = MicroBatchStreams(micro_streams if str(device).startswith('cuda') else 1)   # concurrent half-batches (streams.py)
    # gradient accumulation over SMALL batches (the notebooks' batch_size 4 x aggregate_k_gradients 25): the batches of one optimizer step run whole, round-robin on
    # `aggregate_streams` HIP streams, instead of each being split into column groups (streams.py; measured in bench.py's batch_sweep).  None = automatic.
    explicit_streams = aggregate_streams is not None and aggregate_streams > 1
    plain_schedule_requested = aggregate_streams is not None and aggregate_streams <= 1      # aggregate_streams=0: the caller asked for the plain sequential schedule
    if aggregate_streams is None:
        # (one MI355X, configs[1]: batch 4 x 25 batches 956 datasets/s as column groups, 1680 / 1688 / 1927 on 2 / 4 / 8 alternating streams; batch 8 x 4:
        # 1637 -> 2066; batch 16 x 4: 2175 -> 2346 -- gpurun call 3 of round 4, profiles/r04_small_batch_streams.txt)
        aggregate_streams = min(8, aggregate_k_gradients) if (aggregate_k_gradients >= 2 and dp.local_batch_size(batch_size) * bptt <= 16 * 2048) else 0
    alt = MicroBatchStreams(aggregate_streams) if (aggregate_streams and aggregate_streams > 1 and aggregate_k_gradients > 1 and str(device).startswith('cuda')) else None
    # ... or, better (round 5): the batches of one optimizer step STACKED into one launch set per micro-batch stream, every dataset with its own eval position
    # (streams.py forward_backward_batches, TransformerModel.forward_batches).  aggregate_stacked: None = automatic (small batches under aggregate_k_gradients, a
    # model whose embedding and decoder run inside the HIP stack, and no explicit aggregate_streams request), True / False = forced.
    small_batches = aggregate_k_gradients >= 2 and dp.local_batch_size(batch_size) * bptt <= 16 * 2048
    if aggregate_stacked is None:
        aggregate_stacked = small_batches and not explicit_streams and not plain_schedule_requested
    stacked = bool(aggregate_stacked) and aggregate_k_gradients >= 2 and str(device).startswith('cuda') and micro.can_stack(model)
    # data-parallel runs: the flat gradient buffer is all-reduced as two collectives, the upper layers' half under the backward
    reducer = dp.OverlappedGradientReducer(model) if world > 1 and hasattr(model, 'flat_parameters') else None

    def train_epoch():
        model.train()
        total_loss = torch.zeros((), device=device)
        positional_sum = torch.zeros(bptt, device=device)
        positional_cnt = torch.zeros(bptt)
        before_get_batch = time.time()
        time_to_get_batch = forward_time = step_time = 0.
        assert len(dl) % aggregate_k_gradients == 0, 'Please set the number of steps per epoch s.t. `aggregate_k_gradients` divides it.'
        pending = []      # (eval position, losses) of batches still running on the alternating streams
        stack = []        # (data, targets, eval position) of the batches of the current optimizer step (stacked schedule)
        for batch, (data, targets) in enumerate(dl):
            time_to_get_batch = time.time() - before_get_batch
            before_forward = time.time()
            single_eval_pos = single_eval_pos_gen() if callable(single_eval_pos_gen) else single_eval_pos_gen
            data = tuple(e.to(device) for e in data) if isinstance(data, tuple) else data.to(device)
            last_micro_step = batch % aggregate_k_gradients == aggregate_k_gradients - 1
            if isinstance(data, tuple) and single_eval_pos is not None and stacked:
                stack.append((data, targets.to(device), single_eval_pos))
                loss = None
                if last_micro_step:
                    outs = micro.forward_backward_batches(model, stack, lambda out, tg, sep: compute_losses(criterion, out, tg[sep:], n_out),
                                                          before=(lambda n: reducer.arm(n)) if reducer is not None else None)
                    pending = [(sep_k, l) for (_, _, sep_k), l in zip(stack, outs)]
                    stack = []
                forward_time = time.time() - before_forward
            elif isinstance(data, tuple) and single_eval_pos is not None and alt is not None and alt.can_alternate(model):
                targets = targets.to(device)
                if reducer is not None and last_micro_step:
                    # data-parallel + alternating streams: the LAST batch of the optimizer step is armed, so the upper layers' half of the
                    # all-reduce starts behind its early weight-gradient launch; the collective also waits for the batches of this step that are still
                    # running on the other streams (their weight gradients land in the same buffer)
                    reducer.arm(1, wait_for=alt.fence_others(batch))
                losses = alt.forward_backward_on(batch, model, data, targets, single_eval_pos,
                                                 lambda out, tg, sep=single_eval_pos: compute_losses(criterion, out, tg[sep:], n_out))
                pending.append((single_eval_pos, losses))
                loss = None
                forward_time = time.time() - before_forward
            elif isinstance(data, tuple) and single_eval_pos is not None:
                targets = targets.to(device)
                if reducer is not None and last_micro_step:
                    reducer.arm(micro.groups(model, data[0].shape[1]))
                losses = micro.forward_backward(model, data, targets, single_eval_pos,
                                                lambda out, tg: compute_losses(criterion, out, tg[single_eval_pos:], n_out))
                loss = losses.mean()
                forward_time = time.time() - before_forward
            else:
                if reducer is not None and last_micro_step:
                    reducer.arm(1)
                output = model(data, single_eval_pos=single_eval_pos)
                forward_time = time.time() - before_forward
                if single_eval_pos is not None:
                    targets = targets[single_eval_pos:]
                losses = compute_losses(criterion, output, targets.to(device), n_out)
                loss = losses.mean()
                loss.backward()
            if last_micro_step:
                if alt is not None:
                    alt.join()       # every batch of this optimizer step has finished its backward (in stream order)
                if reducer is not None:
                    reducer.finish()
                elif world > 1:
                    dp.all_reduce_gradients(model.flat_parameters()[1])
                optimizer.step(zero_grad=True)
            step_time = time.time() - before_forward

            with torch.no_grad():
                if loss is None:     # alternating streams: the losses are folded in at the join (their streams have been waited for)
                    if last_micro_step and pending:
                        # one stack + one scatter-add for the whole optimizer step (25 batches at the notebooks' recipe: three tiny launches each added up)
                        lks = torch.stack([losses_k.mean() for _, losses_k in pending])
                        idx = torch.tensor([sep_k % bptt if sep_k < 0 else sep_k for sep_k, _ in pending], dtype=torch.long)      # (negative positions count from the end, as in model.forward; index_add_ takes none)
                        total_loss += lks.sum()
                        positional_sum.index_add_(0, idx.to(device, non_blocking=True), lks)
                        positional_cnt.index_add_(0, idx, torch.ones(len(pending)))
                        pending = []
                else:
                    total_loss += loss.detach()
                    if single_eval_pos is None:
                        positional_sum += losses.detach().mean(1)
                        positional_cnt += 1
                    else:
                        positional_sum[single_eval_pos] += loss.detach()
                        positional_cnt[single_eval_pos] += 1
            before_get_batch = time.time()
        if world > 1:
            torch.distributed.all_reduce(total_loss)
            total_loss /= world
        positional = (positional_sum.cpu() / positional_cnt).tolist()
        return total_loss.item() / steps_per_epoch, positional, time_to_get_batch, forward_time, step_time

    total_loss = float('inf')
    total_positional_losses = float('inf')
    skipped = 0
    for epoch in range(1, epochs + 1):
        epoch_start_time = time.time()
        total_loss, total_positional_losses, time_to_get_batch, forward_time, step_time = train_epoch()
        if hasattr(dl, 'validate') and epoch % validation_period == 0:
            with torch.no_grad():
                val_score = dl.validate(model)
        else:
            val_score = None

        if verbose and dp.rank() == 0:
            print('-' * 89)
            print(
                f'| end of epoch {epoch:3d} | time: {(time.time() - epoch_start_time):5.2f}s | mean loss {total_loss:5.2f} | '
                f"pos losses {','.join([f'{l:5.2f}' for l in total_positional_losses])}, lr {scheduler.get_last_lr()[0]}"
                f' data time {time_to_get_batch:5.2f} step time {step_time:5.2f}'
                f' forward time {forward_time:5.2f}' + (f'val score {val_score}' if val_score is not None else ''))
            print('-' * 89)
        skipped, skipped_before = optimizer.skipped_steps(), skipped      # (the epoch's loss has just been read back: no extra wait)
        if skipped > skipped_before:
            warnings.warn(f'epoch {epoch}: {skipped - skipped_before} optimizer step(s) skipped -- the gradient held an inf / NaN (csrc/optim.hip)', RuntimeWarning)
        if epoch_callback is not None:      # (model, epoch, mean loss, learning rate of the epoch, seconds): loss curves / checkpoints
            epoch_callback(model, epoch, total_loss, scheduler.get_last_lr()[0], time.time() - epoch_start_time)
        scheduler.step()
    if reducer is not None:
        reducer.detach()      # the returned model carries no reference to the reducer, its stream or the GPU gradient buffer
    model.optimizer_steps_skipped = skipped      # steps whose gradient held an inf / NaN (warned about per epoch above)
    return total_loss, total_positional_losses, model.to('cpu')


def _parse_args(config_parser, parser, argv=None):
    """Optional YAML overlay, then the normal parser (reference train.py:137-151)."""
    args_config, remaining = config_parser.parse_known_args(argv)
    if args_config.config:
        with open(args_config.config, 'r') as f:
            parser.set_defaults(**yaml.safe_load(f))
    args = parser.parse_args(remaining)
    return args, yaml.safe_dump(args.__dict__, default_flow_style=False)


def main(argv=None):
    config_parser = argparse.ArgumentParser(description='Only used as a first parser for the config file path.')
    config_parser.add_argument('--config')
    parser = argparse.ArgumentParser()
    parser.add_argument('prior', choices=['gp', 'mix_gp', 'ridge'])   # the reference's 'stroke' prior needs its image data (out of scope)
    parser.add_argument('--loss_function', default='barnll')
    parser.add_argument('--min_y', type=float, help='barnll can only model y in strict ranges, this is the minimum y can take.')
    parser.add_argument('--max_y', type=float, help='barnll can only model y in strict ranges, this is the maximum y can take.')
    parser.add_argument('--num_buckets', default=100, type=int)
    parser.add_argument("--extra_prior_kwargs_dict", default={'fuse_x_y': False}, dest="extra_prior_kwargs_dict",
                        action=StoreDictKeyPair, nargs="+", metavar="KEY=VAL", help='e.g. num_features=5 (required by the GP priors).')
    parser.add_argument('--encoder', default='linear', choices=['linear'])
    parser.add_argument('--y_encoder', default='linear', choices=['linear'])
    parser.add_argument('--pos_encoder', default='sinus', choices=['none', 'sinus', 'learned', 'paired_scrambled_learned'])   # reference default (:168)
    parser.add_argument('--bptt', default=10, type=int)
    parser.add_argument('--epochs', default=200, type=int)
    parser.add_argument('--warmup_epochs', default=50, type=int)
    parser.add_argument('--validation_period', default=10, type=int)
    parser.add_argument('--permutation_invariant_max_eval_pos', default=None, type=int)
    parser.add_argument('--permutation_invariant_sampling', default='weighted', choices=['weighted', 'uniform'])
    parser.add_argument('--emsize', default=512, type=int)
    parser.add_argument('--nlayers', default=6, type=int)
    parser.add_argument('--nhid', default=None, type=int)  # 2*emsize is the default
    parser.add_argument('--nhead', default=4, type=int)
    parser.add_argument('--dropout', default=.0, type=float)
    parser.add_argument('--steps_per_epoch', default=10, type=int)
    parser.add_argument('--batch_size', default=1000, type=int)
    parser.add_argument('--lr', '--learning_rate', default=.001, type=float)
    parser.add_argument('--precision', default='fp16', choices=['bf16', 'fp16', 'f32'])
    args, _ = _parse_args(config_parser, parser, argv)
    cfg = dict(args.__dict__)
    cfg.pop('config', None)
    if cfg['nhid'] is None:
        cfg['nhid'] = 2 * cfg['emsize']

    _, world, local = dp.init_from_env()
    if world > 1:
        cfg['gpu_device'] = f'cuda:{local}'               # one process per GPU (torchrun sets LOCAL_RANK)
        cfg['extra_prior_kwargs_dict'] = {**cfg['extra_prior_kwargs_dict'], 'device': cfg['gpu_device']}   # also for get_y_sample below
    prior = {'gp': priors.fast_gp.DataLoader, 'mix_gp': priors.fast_gp_mix.DataLoader, 'ridge': priors.ridge.DataLoader}[cfg.pop('prior')]
    loss_function, num_buckets = cfg.pop('loss_function'), cfg.pop('num_buckets')
    max_y, min_y = cfg.pop('max_y'), cfg.pop('min_y')

    def get_y_sample():
        dl = prior(num_steps=1, batch_size=cfg['batch_size'] * cfg['steps_per_epoch'], seq_len=cfg['bptt'], **cfg['extra_prior_kwargs_dict'])
        y_sample = next(iter(dl))[-1]
        print(f'Creating Bar distribution with borders from y sample of size {y_sample.numel()}')
        return y_sample.cpu()

    if loss_function == 'ce':
        criterion = nn.CrossEntropyLoss(reduction='none')
    elif loss_function == 'gaussnll':
        criterion = nn.GaussianNLLLoss(reduction='none', full=True)
    elif loss_function == 'mse':
        criterion = nn.MSELoss(reduction='none')
    elif loss_function == 'barnll':
        criterion = BarDistribution(borders=get_bucket_limits(num_buckets, full_range=(min_y, max_y)))
    elif loss_function == 'adaptivebarnll':
        criterion = BarDistribution(borders=get_bucket_limits(num_buckets, ys=get_y_sample(), full_range=(min_y, max_y)))
    elif loss_function == 'adaptivefullsupportbarnll':
        assert min_y is None and max_y is None, "Please do not specify `min_y` and `max_y` with `unboundedadaptivebarnll`."
        criterion = FullSupportBarDistribution(borders=get_bucket_limits(num_buckets, ys=get_y_sample()))
    else:
        raise NotImplementedError(f'loss_function == {loss_function}.')

    generators = {'linear': encoders.Linear}
    encoder_generator = generators[cfg.pop('encoder')]
    y_encoder_generator = generators[cfg.pop('y_encoder')]
    pos_encoder_generator = {'none': None, 'sinus': positional_encodings.PositionalEncoding,
                             'learned': positional_encodings.LearnedPositionalEncoding,
                             'paired_scrambled_learned': positional_encodings.PairedScrambledPositionalEncodings}[cfg.pop('pos_encoder')]
    max_eval_pos = cfg.pop('permutation_invariant_max_eval_pos')
    sampling = cfg.pop('permutation_invariant_sampling')
    if max_eval_pos is not None:
        get_sampler = get_weighted_single_eval_pos_sampler if sampling == 'weighted' else get_uniform_single_eval_pos_sampler
        cfg['single_eval_pos_gen'] = get_sampler(max_eval_pos)

    print("ARGS for `train`:", cfg)
    return train(prior, criterion, encoder_generator, y_encoder_generator=y_encoder_generator,
                 pos_encoder_generator=pos_encoder_generator, **cfg)


if __name__ == '__main__':
    main()
