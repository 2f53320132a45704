"""Replays the batches and the eval-position stream that oracle/make_golden.py::train_loop_case recorded through a `train()` function, the
way the reference's own `train.train` consumed them when the fixture was made (tests/golden/train_loop_small.pt): test infrastructure."""
import torch


class ReplayLoader:
    """PriorDataLoader protocol (reference priors/prior.py:4-12) over recorded batches; the cursor runs on across epochs."""
    num_outputs = 1
    fuse_x_y = False

    def __init__(self, num_steps, batch_size=None, seq_len=None, batches=None, **_):
        self.num_steps, self.batches, self.cursor = num_steps, batches, 0
        self.num_features = batches[0][0].shape[-1]

    def __len__(self):
        return self.num_steps

    def __iter__(self):
        for _ in range(self.num_steps):
            x, y = self.batches[self.cursor % len(self.batches)]
            self.cursor += 1
            yield (x, y), y.clone()


def replay(train_fn, rec, criterion_base, encoders, schedule_fn, **extra):
    """Runs train_fn (this repo's train.train, or a stand-in-patched one) on the recorded stream.  Returns (per-batch losses, learning rate of
    every batch, returned total loss, final state dict)."""
    cfg = rec['config']
    log = dict(loss=[], lr=[])
    state = {}

    class Recording(criterion_base):
        def forward(self, logits, y):
            losses = super().forward(logits, y)
            log['loss'].append(losses.detach().mean())
            log['lr'].append(state['opt'].param_groups[0]['lr'])
            return losses

    Recording.__name__ = 'RecordingFullSupportBarDistribution'
    it = iter(rec['seps'])

    def scheduler(optimizer, warmup, total):
        state['opt'] = optimizer
        return schedule_fn(optimizer, warmup, total)

    total, positional, model = train_fn(
        ReplayLoader, Recording(rec['borders'].clone()), encoders.Linear, emsize=cfg['E'], nhid=cfg['nhid'], nlayers=cfg['L'], nhead=cfg['H'], dropout=0.0,
        epochs=cfg['epochs'], steps_per_epoch=cfg['steps_per_epoch'], batch_size=cfg['B'], bptt=cfg['T'], lr=cfg['lr'], warmup_epochs=cfg['warmup_epochs'],
        y_encoder_generator=encoders.Linear, extra_prior_kwargs_dict=dict(batches=rec['batches']), scheduler=scheduler,
        load_weights_from_this_state_dict={k: v.clone() for k, v in rec['init_state_dict'].items()}, single_eval_pos_gen=lambda: next(it),
        aggregate_k_gradients=cfg['aggregate_k_gradients'], verbose=False, **extra)
    return [float(v) for v in log['loss']], log['lr'], total, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}


def update_error(final, rec):
    """relative L2 error of the whole parameter UPDATE (final - initial) against the reference's"""
    num = den = 0.0
    for k, v in rec['final_state_dict'].items():
        if k.startswith('criterion.'):
            continue
        d_ref = v.double() - rec['init_state_dict'][k].double()
        d_got = final[k].double() - rec['init_state_dict'][k].double()
        num += ((d_got - d_ref) ** 2).sum().item()
        den += (d_ref ** 2).sum().item()
    return (num / den) ** 0.5
