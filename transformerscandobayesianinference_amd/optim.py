"""Fused global-norm clip + Adam on the model's flat parameter buffer.

Replaces `clip_grad_norm_(model.parameters(), 1.)` + `torch.optim.Adam.step()` +
`optimizer.zero_grad()` of the reference loop (train.py:55,95-97) by ONE call into libpfn_hip.so
(`pfn_clip_adam_step`, csrc/optim.hip).  It is a torch.optim.Optimizer only so the reference's
LambdaLR schedulers (utils.py:10-51) can drive `param_groups[0]['lr']` unchanged.
"""
import torch

from transformerscandobayesianinference_amd import _hip


class FusedClipAdam(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=1.0):
        self.model = model
        super().__init__(list(model.parameters()), dict(lr=lr, betas=betas, eps=eps, max_grad_norm=max_grad_norm))
        self._step = 0
        self._m = self._v = self._scratch = None
        self.grad_multiplier = 1.0  # set to 1/world_size by the data-parallel wrapper (gradient averaging)

    def _buffers(self):
        flat, grad = self.model.flat_parameters()
        if self._m is None or self._m.numel() != flat.numel() or self._m.device != flat.device:
            self._m = torch.zeros_like(flat)
            self._v = torch.zeros_like(flat)
            self._scratch = torch.zeros(2048, dtype=torch.float32, device=flat.device)
        return flat, grad

    @torch.no_grad()
    def step(self, closure=None, zero_grad=False):
        """One optimizer step; with zero_grad=True the gradient buffer is cleared in the same pass."""
        assert closure is None
        flat, grad = self._buffers()
        g = self.param_groups[0]
        self._step += 1
        _hip.check(_hip.lib().pfn_clip_adam_step(flat.data_ptr(), grad.data_ptr(), self._m.data_ptr(), self._v.data_ptr(),
                                                 flat.numel(), float(g['lr']), g['betas'][0], g['betas'][1], g['eps'],
                                                 float(g['max_grad_norm'] or 0.0), float(self.grad_multiplier), self._step,
                                                 int(zero_grad), self._scratch.data_ptr(), _hip.stream_ptr(flat.device)),
                   'pfn_clip_adam_step')
        self.model.mark_params_updated()

    # ---- checkpointing: the moments live in flat buffers outside torch's per-parameter `state`, so they are exposed here;
    # `(model.state_dict(), optimizer.state_dict())` is the notebooks' checkpoint tuple (tabular.save_checkpoint) ----
    def state_dict(self):
        sd = super().state_dict()
        if self._m is not None:
            sd['flat'] = {'step': self._step, 'exp_avg': self._m.detach().cpu().clone(), 'exp_avg_sq': self._v.detach().cpu().clone()}
        else:
            sd['flat'] = {'step': self._step}
        # grad_multiplier (1 / world size) is run topology, not optimizer state: it is NOT persisted, so a checkpoint resumed
        # on a different number of GPUs keeps the value train() / the DP wrapper set for THIS run
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        flat_state = state_dict.pop('flat', None)
        state_dict.pop('grad_multiplier', None)     # written by round-2 checkpoints; ignored (see state_dict)
        super().load_state_dict(state_dict)
        if flat_state is not None:
            self._step = int(flat_state['step'])
            if 'exp_avg' in flat_state:
                flat, _ = self._buffers()
                if flat_state['exp_avg'].numel() != flat.numel():
                    raise ValueError(f"optimizer state holds {flat_state['exp_avg'].numel()} moments, the model has {flat.numel()} flat parameters")
                self._m.copy_(flat_state['exp_avg'])
                self._v.copy_(flat_state['exp_avg_sq'])

    def zero_grad(self, set_to_none=False):
        _, grad = self._buffers()
        grad.zero_()

    def skipped_steps(self):
        """Steps the kernel skipped because the gradient held an inf / NaN (csrc/optim.hip; parameters and moments untouched); forces a device sync."""
        return 0 if self._scratch is None else int(self._scratch[1025].item())

    def last_grad_norm(self):
        """Global gradient norm seen by the last step (before clipping); forces a device sync."""
        return float(self._scratch[0].item())
