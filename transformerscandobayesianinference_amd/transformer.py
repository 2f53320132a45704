"""TransformerModel: the PFN encoder-only transformer, running on hand-written gfx950 kernels.

API and state-dict parity with the reference `transformer.py` (constructor :14, forward :55-91,
init :43-53); the call `self.transformer_encoder(src, src_mask)` (:84) and everything around it
(embedding :66-74, decoder :85, test-row slice :91) is replaced by ONE autograd node that drives
libpfn_hip.so through the C ABI (`pfn_stack_forward` / `pfn_stack_backward`).

PyTorch's role here is plumbing only: it owns the parameter / gradient / activation memory and the
stream.  All parameters are views into one flat f32 buffer (and their .grad into one flat gradient
buffer) so the optimizer and the data-parallel all-reduce touch a single contiguous range.
There is no PyTorch fallback: CPU tensors or a missing library raise `HipExtensionError`.
"""
import ctypes

import torch
from torch import nn

from transformerscandobayesianinference_amd import _hip
from transformerscandobayesianinference_amd.positional_encodings import NoPositionalEncoding
from transformerscandobayesianinference_amd.utils import SeqBN

_ALIGN = 64  # elements; must match make_layout() in csrc/pfn_api.hip


class _SelfAttentionParams(nn.Module):
    """Parameter container with the key names of torch.nn.MultiheadAttention
    (`in_proj_weight`, `in_proj_bias`, `out_proj.{weight,bias}`) and its default initialisation."""

    def __init__(self, emsize):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * emsize, emsize))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * emsize))
        self.out_proj = nn.Linear(emsize, emsize)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)


class _EncoderLayerParams(nn.Module):
    """Parameter container with the key names of torch.nn.TransformerEncoderLayer."""

    def __init__(self, emsize, nhid):
        super().__init__()
        self.self_attn = _SelfAttentionParams(emsize)
        self.linear1 = nn.Linear(emsize, nhid)
        self.linear2 = nn.Linear(nhid, emsize)
        self.norm1 = nn.LayerNorm(emsize, eps=1e-5)
        self.norm2 = nn.LayerNorm(emsize, eps=1e-5)


class _EncoderParams(nn.Module):
    def __init__(self, emsize, nhid, nlayers):
        super().__init__()
        self.layers = nn.ModuleList([_EncoderLayerParams(emsize, nhid) for _ in range(nlayers)])


class _StackFunction(torch.autograd.Function):
    """logits = stack(x, y | src); one node for embedding + L layers + decoder."""

    @staticmethod
    def forward(ctx, flat_params, model, x, y, src, sep, inference, ragged=None):
        lib = _hip.lib()
        dev = flat_params.device
        stream = _hip.stream_ptr(dev)
        # `inference` is decided by the caller: grad mode is always off inside Function.forward, so it cannot be read here
        desc, shadow = model._operands(stream, inference)
        if src is not None:
            src = src.contiguous().float()
            S, B = src.shape[0], src.shape[1]
            xs = ys = None
            x_ptr = y_ptr = 0
            x_st = x_sb = y_st = y_sb = 0
        else:
            if x.dtype != torch.float32 or x.stride(-1) != 1:
                x = x.float().contiguous()
            if y.dtype != torch.float32:
                y = y.float()
            S, B = x.shape[0], x.shape[1]
            x_ptr, y_ptr = x.data_ptr(), y.data_ptr()
            x_st, x_sb = x.stride(0), x.stride(1)
            y_st, y_sb = y.stride(0), y.stride(1)
        ws_bytes = lib.pfn_workspace_bytes(ctypes.byref(desc), B, S)
        _hip.check(ws_bytes, 'pfn_workspace_bytes')
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        ctx.ragged = ragged
        if ragged is not None:
            # a ragged batch (forward_batches): per-dataset eval positions, compact test rows dataset-major -- one launch set for several micro-batches
            sep_of, row_off, sep_min, sep_max, test_rows = ragged
            logits = torch.empty((test_rows, desc.n_out or desc.emsize), dtype=torch.float32, device=dev)
            ctx.dropout_seed = model._next_dropout_seed() if (model.training and desc.dropout > 0) else None
            _hip.check(lib.pfn_stack_forward_ragged(ctypes.byref(desc), flat_params.data_ptr(), shadow.data_ptr(), x_ptr, x_st, x_sb, y_ptr, y_st, y_sb, B, S,
                                                    sep_of.data_ptr(), row_off.data_ptr(), sep_min, sep_max, test_rows, ws.data_ptr(), ws_bytes, logits.data_ptr(), stream,
                                                    int(ctx.dropout_seed is not None), ctx.dropout_seed or 0), 'pfn_stack_forward_ragged')
            ctx.model, ctx.ws, ctx.dims = model, ws, (B, S, sep_max)
            ctx.operands = (desc, shadow)
            ctx.inputs = (x, y, None)
            ctx.src_needs_grad = False
            return logits
        logits = torch.empty((S - sep, B, desc.n_out or desc.emsize), dtype=torch.float32, device=dev)   # n_out 0: the encoder's test rows
        # dropout (reference train.py:22 default 0.2; TransformerEncoderLayer's four sites): live in training mode only.  Every pass
        # takes its own 64-bit seed; the masks are counter-based functions of it, regenerated by the backward (include/pfn_hip.h)
        ctx.dropout_seed = None
        if model.training and desc.dropout > 0:
            ctx.dropout_seed = model._next_dropout_seed()
            _hip.check(lib.pfn_stack_forward_dropout(ctypes.byref(desc), flat_params.data_ptr(), shadow.data_ptr(),
                                                     x_ptr, x_st, x_sb, y_ptr, y_st, y_sb, _hip.ptr(src), B, S, sep,
                                                     ws.data_ptr(), ws_bytes, logits.data_ptr(), stream, ctx.dropout_seed), 'pfn_stack_forward_dropout')
        else:
            _hip.check(lib.pfn_stack_forward(ctypes.byref(desc), flat_params.data_ptr(), shadow.data_ptr(),
                                             x_ptr, x_st, x_sb, y_ptr, y_st, y_sb, _hip.ptr(src), B, S, sep,
                                             ws.data_ptr(), ws_bytes, logits.data_ptr(), stream), 'pfn_stack_forward')
        ctx.model, ctx.ws, ctx.dims = model, ws, (B, S, sep)
        ctx.operands = (desc, shadow)      # the backward reads the workspace in the layout THIS descriptor's precision wrote
        ctx.inputs = (x, y, src)
        ctx.src_needs_grad = src is not None and ctx.needs_input_grad[4]
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        lib = _hip.lib()
        model, ws = ctx.model, ctx.ws
        B, S, sep = ctx.dims
        x, y, src = ctx.inputs
        dev = ws.device
        stream = _hip.stream_ptr(dev)
        desc, shadow = ctx.operands
        if desc is not model._desc:
            raise RuntimeError('backward through an inference-precision forward: the forward of this graph ran the eval_precision kernels')
        model._attach_grads()
        dlogits = dlogits.contiguous().float()
        dsrc = None
        if src is not None:
            dsrc = torch.empty_like(src)
            args = (0, 0, 0, 0, 0, 0)
        else:
            args = (x.data_ptr(), x.stride(0), x.stride(1), y.data_ptr(), y.stride(0), y.stride(1))
        hook = model._first_group_hook
        split = hook is not None and hook.armed()
        if ctx.ragged is not None:
            sep_of, row_off, sep_min, sep_max, test_rows = ctx.ragged
            cb = _hip.HOST_CALLBACK(lambda user: hook.first_group_launched()) if split else None
            _hip.check(lib.pfn_stack_backward_ragged(ctypes.byref(desc), model._flat.data_ptr(), shadow.data_ptr(), *args, B, S,
                                                     sep_of.data_ptr(), row_off.data_ptr(), sep_min, sep_max, test_rows, ws.data_ptr(), ws.numel(), dlogits.data_ptr(),
                                                     model._flat_grad.data_ptr(), stream, hook.first_group_layers if split else 0, cb, None,
                                                     int(ctx.dropout_seed is not None), ctx.dropout_seed or 0), 'pfn_stack_backward_ragged')
            ctx.ws = None
            return None, None, None, None, None, None, None, None
        if split or ctx.dropout_seed is not None:
            # data-parallel runs (dp.OverlappedGradientReducer): the top layers' weight gradients are launched early and the hook
            # records an event right behind them, so their all-reduce can run under the rest of this backward.
            # dropout: the backward regenerates the forward's masks from its seed.
            cb = _hip.HOST_CALLBACK(lambda user: hook.first_group_launched()) if split else None
            _hip.check(lib.pfn_stack_backward_split(ctypes.byref(desc), model._flat.data_ptr(), shadow.data_ptr(), *args,
                                                    B, S, sep, ws.data_ptr(), ws.numel(), dlogits.data_ptr(),
                                                    model._flat_grad.data_ptr(), _hip.ptr(dsrc), stream,
                                                    hook.first_group_layers if split else 0, cb, None,
                                                    int(ctx.dropout_seed is not None), ctx.dropout_seed or 0),
                       'pfn_stack_backward_split')
        else:
            _hip.check(lib.pfn_stack_backward(ctypes.byref(desc), model._flat.data_ptr(), shadow.data_ptr(), *args,
                                              B, S, sep, ws.data_ptr(), ws.numel(), dlogits.data_ptr(),
                                              model._flat_grad.data_ptr(), _hip.ptr(dsrc), stream), 'pfn_stack_backward')
        ctx.ws = None
        return None, None, None, None, (dsrc if ctx.src_needs_grad else None), None, None, None


def ragged_layout(T, widths, single_eval_positions):
    """Host side of a ragged batch (pfn_stack_forward_ragged): batch k has widths[k] datasets at eval position single_eval_positions[k] (negative positions count
    from the end, as slicing does in the reference).  Returns (clamped positions per batch, eval position per dataset [B], first compact test row per dataset
    [B + 1]): dataset b's test rows are row_off[b] + (t - sep_of[b]) for t >= sep_of[b], batches and datasets in order."""
    seps = []
    for sep in single_eval_positions:
        sep = int(sep)
        seps.append(max(0, min(T, sep + T if sep < 0 else sep)))
    per_dataset = [s for s, w in zip(seps, widths) for _ in range(w)]
    offs = [0]
    for s in per_dataset:
        offs.append(offs[-1] + T - s)
    return seps, per_dataset, offs


class TransformerModel(nn.Module):
    requires_gpu = True   # train() checks this before building anything (the host-plumbing tests substitute a CPU stand-in)

    def __init__(self, encoder, n_out, ninp, nhead, nhid, nlayers, dropout=0.0, y_encoder=None, pos_encoder=None,
                 decoder=None, input_normalization=False, precision='fp16', eval_precision='f32', deterministic=False):
        super().__init__()
        self.model_type = 'Transformer'
        self.transformer_encoder = _EncoderParams(ninp, nhid, nlayers)
        self.ninp, self.nhead, self.nhid, self.nlayers, self.n_out = ninp, nhead, nhid, nlayers, n_out
        self.dropout = dropout
        self.encoder = encoder
        self.y_encoder = y_encoder
        self.pos_encoder = pos_encoder
        # reference :23: a decoder generator replaces the default MLP.  The default runs inside the HIP stack; a custom module
        # (decoders.ScaledDecoder ...) runs in PyTorch on the encoder's test rows, which the stack then returns instead of logits.
        self._custom_decoder = decoder is not None
        self.decoder = decoder(ninp, nhid, n_out) if decoder is not None else nn.Sequential(nn.Linear(ninp, nhid), nn.GELU(), nn.Linear(nhid, n_out))
        self.input_ln = SeqBN(ninp) if input_normalization else None
        # `precision`: operand type of the TRAINING path ('fp16', the default and the benchmarked mode since round 6: fp16 MFMA operands under a device-side loss
        # scale, saturating stores, keys centred, pre-LayerNorm sums in fp16; 'bf16' = rounds 1-5's path, same rate; 'f32' = exact-f32 parity mode).
        # `eval_precision`: operand type of INFERENCE passes -- model.eval() under torch.no_grad(), i.e. everything that produces
        # posterior-predictive outputs (validate / run_test / criterion.mean).  Default 'f32': the same kernels on the f32 matrix
        # instructions, so the outputs match the reference's CPU path to 1e-6 where 16-bit operands leave 2e-3 (fp16) to 5e-2 (bf16) on a TRAINED model
        # (profiles/r03_trained_*.json); a forward-only pass is not the throughput path.  None = as in training.
        self.precision = precision
        self.eval_precision = eval_precision
        self._flat = self._flat_grad = self._shadow = None
        self.schedule = None               # pfn_model_desc.schedule bits (_hip.SCHED_*); None = the library's defaults at the time the model is packed
        # deterministic=True: bit-reproducible gradients (PFN_SCHED_DETERMINISTIC: one writer per gradient element and launch, ordered partial sums;
        # the micro-batch / alternating streams of train() are switched off for such a model) -- the reference's CPU loop is deterministic (train.py:58-110)
        if deterministic:
            self.schedule = _hip.SCHED_DETERMINISTIC
        self._shadow_version = None
        self._desc = None
        self._eval_desc = self._eval_shadow = self._eval_shadow_version = None
        self._first_group_hook = None      # set by dp.OverlappedGradientReducer
        self._views = []
        self.init_weights()
        # a state dict loaded into a model that has already run keeps the flat views but changes their contents
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.mark_params_updated())

    # ---- reference helpers kept for API parity (host-side, unused by the kernels) ----
    @staticmethod
    def generate_square_subsequent_mask(sz):
        allowed = torch.tril(torch.ones(sz, sz, dtype=torch.bool))
        return torch.zeros(sz, sz).masked_fill(~allowed, float('-inf'))

    @staticmethod
    def generate_D_q_matrix(sz, query_size):
        """0/-inf mask: key j visible to query i iff j < sz - query_size or i == j (reference :34-41).
        The kernels never build it -- they take the integer `single_eval_pos`."""
        train_size = sz - query_size
        allowed = torch.zeros(sz, sz, dtype=torch.bool)
        allowed[:, :train_size] = True
        allowed |= torch.eye(sz, dtype=torch.bool)
        return torch.zeros(sz, sz).masked_fill(~allowed, float('-inf'))

    @classmethod
    def _check_src_mask(cls, src_mask, T, sep):
        """An explicit `src_mask` (reference :60-65 uses it in place of the one it would build) is accepted when it IS the mask the reference builds for this
        eval position -- generate_D_q_matrix(T, T - sep), as a 0/-inf float mask or as torch's boolean `True = masked` form: that is what the kernels
        implement from the integer.  Any other mask has no kernel here and raises (no caller on the path named by north_star passes one)."""
        want = cls.generate_D_q_matrix(T, T - sep).to(src_mask.device) == 0              # True = visible
        if tuple(src_mask.shape) != (T, T):
            raise ValueError(f'src_mask has shape {tuple(src_mask.shape)}, expected ({T}, {T})')
        visible = ~src_mask if src_mask.dtype == torch.bool else src_mask == 0
        finite_ok = src_mask.dtype == torch.bool or bool(((src_mask == 0) | (src_mask == float('-inf'))).all())
        if not finite_ok or not torch.equal(visible, want):
            raise NotImplementedError('src_mask differs from generate_D_q_matrix(len(x), len(x) - single_eval_pos): the kernels implement exactly that mask '
                                      '(keys below single_eval_pos + the query itself); arbitrary additive masks are not supported')

    @property
    def deterministic(self):
        return bool((self.__dict__.get('_schedule') or 0) & _hip.SCHED_DETERMINISTIC)

    @property
    def schedule(self):
        return self.__dict__.get('_schedule')

    @schedule.setter
    def schedule(self, bits):
        # the descriptor is cached when the parameters are packed: a change of the schedule after the first forward rebuilds it (it was silently
        # ignored).  The schedule does not touch the parameter layout, so the flat buffers (and any optimizer holding them) stay as they are.
        changed = bits != self.__dict__.get('_schedule')
        self.__dict__['_schedule'] = bits
        if changed and self.__dict__.get('_flat') is not None:
            self._desc = self._make_desc()
            if self._eval_desc is not None:
                self._eval_desc = self._make_desc(self.eval_precision)

    def init_weights(self):
        """Zero the two residual-branch output projections of every layer (reference :43-53)."""
        for layer in self.transformer_encoder.layers:
            nn.init.zeros_(layer.linear2.weight)
            nn.init.zeros_(layer.linear2.bias)
            nn.init.zeros_(layer.self_attn.out_proj.weight)
            nn.init.zeros_(layer.self_attn.out_proj.bias)

    # ---- flat parameter storage ----
    def _fused_embedding(self):
        return (type(self.encoder) is nn.Linear and type(self.y_encoder) is nn.Linear and self.input_ln is None
                and (self.pos_encoder is None or isinstance(self.pos_encoder, NoPositionalEncoding)))

    def _stack_parameters(self):
        """Parameters in the order of pfn_param_layout (state-dict order of the reference model)."""
        ps = []
        if self._fused_embedding():
            ps += [self.encoder.weight, self.encoder.bias, self.y_encoder.weight, self.y_encoder.bias]
        else:
            ps += [None, None, None, None]
        for l in self.transformer_encoder.layers:
            ps += [l.self_attn.in_proj_weight, l.self_attn.in_proj_bias, l.self_attn.out_proj.weight, l.self_attn.out_proj.bias,
                   l.linear1.weight, l.linear1.bias, l.linear2.weight, l.linear2.bias,
                   l.norm1.weight, l.norm1.bias, l.norm2.weight, l.norm2.bias]
        if not self._custom_decoder:
            ps += [self.decoder[0].weight, self.decoder[0].bias, self.decoder[2].weight, self.decoder[2].bias]
        return ps

    def _make_desc(self, precision=None):
        nf = self.encoder.in_features if self._fused_embedding() else 1
        prec = _hip.PRECISIONS[precision or self.precision]
        # schedule: the library's defaults (0) unless a test / profiling run changed them (pfn_set_tuning) or the model carries its own bits; fixed per
        # descriptor, so a forward and the backward that reads its workspace always agree about the buffer layout (ABI 6)
        sched = getattr(self, 'schedule', None)       # (a model pickled whole before ABI 6 has no such attribute: ADVICE r4)
        schedule = sched if sched is not None else _hip.lib().pfn_default_schedule()
        return _hip.ModelDesc(nf, self.ninp, self.nhead, self.nhid, self.nlayers, 0 if self._custom_decoder else self.n_out, prec, 1e-5, float(self.dropout or 0.0), schedule)

    def _is_flat(self):
        if self._flat is None:
            return False
        stack = [p for p in self._stack_parameters() if p is not None]
        first, last = stack[0], stack[-1]
        lo, hi = self._flat.data_ptr(), self._flat.data_ptr() + self._flat.numel() * 4
        return all(p.is_cuda and p.device == self._flat.device and lo <= p.data_ptr() < hi for p in (first, last))

    def _flatten(self, device):
        """(Re)pack every parameter into one flat f32 buffer; parameters become views of it."""
        lib = _hip.lib()
        desc = self._make_desc()
        n = _hip.check(lib.pfn_param_layout(ctypes.byref(desc), None, None, 0), 'pfn_param_layout')
        offs = (ctypes.c_int64 * n)()
        nums = (ctypes.c_int64 * n)()
        _hip.check(lib.pfn_param_layout(ctypes.byref(desc), offs, nums, n), 'pfn_param_layout')
        stack = self._stack_parameters()
        assert len(stack) == n
        total = lib.pfn_param_count(ctypes.byref(desc))
        placed = {id(p) for p in stack if p is not None}
        extras, cursor = [], total
        for p in self.parameters():
            if id(p) not in placed:
                extras.append((p, cursor))
                placed.add(id(p))
                cursor = (cursor + p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        flat = torch.zeros(cursor, dtype=torch.float32, device=device)
        grad = torch.zeros(cursor, dtype=torch.float32, device=device)
        with torch.no_grad():
            for p, off, num in list(zip(stack, offs, nums)) + [(p, o, p.numel()) for p, o in extras]:
                if p is None:
                    continue
                assert p.numel() == num, f'parameter of {p.numel()} elements does not fit its slot of {num}'
                old_grad = p.grad
                flat[off:off + num].copy_(p.detach().reshape(-1).to(device))
                p.data = flat[off:off + num].view(p.shape)
                if old_grad is not None:
                    grad[off:off + num].copy_(old_grad.reshape(-1).to(device))
                p.grad = grad[off:off + num].view(p.shape)
        self._flat, self._flat_grad, self._desc = flat, grad, desc
        self._views = [(p, off, num) for p, off, num in list(zip(stack, offs, nums)) + [(p, o, p.numel()) for p, o in extras] if p is not None]
        self._stack_numel = total
        self._shadow = torch.empty(lib.pfn_shadow_bytes(ctypes.byref(desc)), dtype=torch.uint8, device=device)
        self._shadow_version = None
        self._eval_desc = self._eval_shadow = self._eval_shadow_version = None
        if self.eval_precision not in (None, self.precision) and self._make_desc(self.eval_precision).precision != desc.precision:
            self._eval_desc = self._make_desc(self.eval_precision)        # its operand copies are allocated at the first inference pass
            if lib.pfn_shadow_bytes(ctypes.byref(self._eval_desc)) < 0:   # shape outside that precision's kernels:
                self._eval_desc = None                                    # inference then runs in the training precision

    def _attach_grads(self):
        """Point every .grad at its slice of the flat gradient buffer (after zero_grad(set_to_none=True)
        the buffer is cleared first, so stale values are not re-used)."""
        first = self._views[0][0]
        if first.grad is not None and first.grad.data_ptr() == self._flat_grad.data_ptr() + self._views[0][1] * 4:
            return
        if first.grad is None:
            self._flat_grad.zero_()
        for p, off, num in self._views:
            if p.grad is not None and p.grad.data_ptr() != self._flat_grad.data_ptr() + off * 4:
                self._flat_grad[off:off + num].add_(p.grad.reshape(-1))
            p.grad = self._flat_grad[off:off + num].view(p.shape)

    def mark_params_updated(self):
        """Called by whoever writes the parameters behind autograd's back: optimizers that update the flat buffer
        through raw pointers (FusedClipAdam), or code that assigns through `p.data` (which bumps no version counter)."""
        self._shadow_version = None
        self._eval_shadow_version = None

    def _param_version(self):
        """Every Parameter is a view of the flat buffer with its OWN version counter (`p.data = flat[...]` detaches
        it from the buffer's), so in-place updates through the parameters -- `load_state_dict`, `torch.optim.*`,
        `p.copy_()` -- are only visible there; updates of the flat buffer itself only on the buffer."""
        return (self._flat._version, sum(p._version for p, _, _ in self._views))

    def _refresh_shadow(self, stream):
        version = self._param_version()
        if self._shadow_version != version:
            _hip.check(_hip.lib().pfn_prepare_params(ctypes.byref(self._desc), self._flat.data_ptr(), self._shadow.data_ptr(), stream),
                       'pfn_prepare_params')
            self._shadow_version = version

    def _inference_pass(self):
        """True when this forward produces outputs only (eval mode, autograd off) and a separate inference precision is configured.
        Must be evaluated in `forward`, OUTSIDE the autograd Function (inside `Function.forward` grad mode is always off): an eval-mode
        forward that is differentiated afterwards (fine-tuning under eval(), input gradients) has to run the training-precision
        kernels, whose workspace layout the backward reads."""
        return self._eval_desc is not None and not self.training and not torch.is_grad_enabled()

    def _operands(self, stream, inference):
        """(model descriptor, operand copies of the weights) of a pass, refreshed if the parameters changed."""
        if not inference:
            self._refresh_shadow(stream)
            return self._desc, self._shadow
        lib = _hip.lib()
        if self._eval_shadow is None:
            self._eval_shadow = torch.empty(lib.pfn_shadow_bytes(ctypes.byref(self._eval_desc)), dtype=torch.uint8, device=self._flat.device)
        version = self._param_version()
        if self._eval_shadow_version != version:
            _hip.check(lib.pfn_prepare_params(ctypes.byref(self._eval_desc), self._flat.data_ptr(), self._eval_shadow.data_ptr(), stream), 'pfn_prepare_params')
            self._eval_shadow_version = version
        return self._eval_desc, self._eval_shadow

    def _next_dropout_seed(self):
        """A fresh 64-bit seed per training pass, derived from torch's global seed (rank-distinct under data parallelism) and a counter."""
        self._dropout_calls = getattr(self, '_dropout_calls', 0) + 1
        self._last_dropout_seed = (torch.initial_seed() * 0x9E3779B97F4A7C15 + self._dropout_calls * 0xD1B54A32D192ED03) & (2 ** 64 - 1)
        return self._last_dropout_seed

    def layer_offset(self, layer):
        """Offset (in elements) of encoder layer `layer`'s first parameter in the flat buffers: everything from there to the end of the
        buffer belongs to that layer, the layers above it and the decoder (state-dict order)."""
        self.flat_parameters()
        first = self.transformer_encoder.layers[layer].self_attn.in_proj_weight
        return next(off for p, off, _ in self._views if p is first)

    def flat_parameters(self):
        """(flat f32 parameter buffer, flat f32 gradient buffer); packs the model on first use."""
        if not self._is_flat():
            dev = next(self.parameters()).device
            _hip.require_gpu_tensor(next(self.parameters()), 'model parameters')
            self._flatten(dev)
        self._attach_grads()
        return self._flat, self._flat_grad

    # ---- several micro-batches in one launch set ----
    def can_forward_batches(self):
        return self._fused_embedding() and not self._custom_decoder

    def forward_batches(self, batches, single_eval_positions):
        """`model((x_k, y_k), single_eval_pos=sep_k)` for several batches at once: batches = [(x_k[T, b_k, F], y_k[T, b_k]), ...] with one T, one eval position
        per batch.  Returns the list of logits [T - sep_k, b_k, n_out] -- the tensors the separate calls would return -- from ONE launch set in which every
        dataset carries its own eval position (pfn_stack_forward_ragged).  The reference runs the k batches of an optimizer step one after the other
        (train.py:66-97); a batch of 4 datasets fills a fraction of the chip.  Losses formed per batch and summed give the reference's accumulated gradient."""
        assert self.can_forward_batches(), 'forward_batches needs the built-in linear encoders and decoder'
        assert len(batches) == len(single_eval_positions) and len(batches) > 0
        x = torch.cat([b[0] for b in batches], 1)
        y = torch.cat([b[1].to(x.device) for b in batches], 1)
        _hip.require_gpu_tensor(x, 'x')
        _hip.require_gpu_tensor(next(self.parameters()), 'model parameters')
        T = x.shape[0]
        for xb, _ in batches:
            assert xb.shape[0] == T, 'forward_batches: every batch needs the same sequence length'
        widths = [xb.shape[1] for xb, _ in batches]
        seps, per_dataset, offs = ragged_layout(T, widths, single_eval_positions)
        meta = torch.tensor(per_dataset + offs, dtype=torch.int64)          # one host-to-device copy for both arrays
        sep_of = meta[:len(per_dataset)].to(torch.int32).to(x.device, non_blocking=True)
        row_off = meta[len(per_dataset):].to(x.device, non_blocking=True)
        if not self._is_flat():
            self._flatten(x.device)
        flat = self._flat
        inference = self._inference_pass()
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            flat = flat.detach().requires_grad_(True)
        logits = _StackFunction.apply(flat, self, x, y, None, max(seps), inference, (sep_of, row_off, min(seps), max(seps), offs[-1]))
        # (torch.split: its backward is ONE concatenation of the per-batch gradients; slicing would zero-fill and add a full-size tensor per batch)
        parts = torch.split(logits, [w * (T - sep) for sep, w in zip(seps, widths)])
        return [p.view(w, T - sep, logits.shape[1]).transpose(0, 1) for p, sep, w in zip(parts, seps, widths)]      # dataset-major rows -> the reference's [T - sep, b, n_out]

    # ---- forward ----
    def forward(self, src, src_mask=None, single_eval_pos=None):
        assert single_eval_pos is not None, 'Single eval pos is required now.'
        assert isinstance(src, tuple), 'pass src as an (x, y) tuple together with single_eval_pos (the fuse_x_y path of the reference is dead code: transformer.py:56-59)'
        x_src, y_src = src
        _hip.require_gpu_tensor(x_src, 'x')
        _hip.require_gpu_tensor(next(self.parameters()), 'model parameters')
        T = x_src.shape[0]
        sep = int(single_eval_pos)
        if sep < 0:
            sep += T  # negative positions work through slicing in the reference
        sep = max(0, min(T, sep))
        if src_mask is not None:
            self._check_src_mask(src_mask, T, sep)
        if not self._is_flat():
            self._flatten(x_src.device)
        flat = self._flat
        inference = self._inference_pass()
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            flat = flat.detach().requires_grad_(True)  # connects the node to autograd; grads go to the flat buffer directly
        if self._fused_embedding():
            out = _StackFunction.apply(flat, self, x_src, y_src.to(x_src.device), None, sep, inference)
            return self.decoder(out) if self._custom_decoder else out
        # custom encoders / positional encodings / SeqBN: PyTorch computes the embedding, HIP runs the stack
        x_emb = self.encoder(x_src)
        y_emb = self.y_encoder(y_src.unsqueeze(-1))
        emb = torch.cat([x_emb[:sep] + y_emb[:sep], x_emb[sep:]], 0)
        if self.input_ln is not None:
            emb = self.input_ln(emb)
        if self.pos_encoder is not None:
            emb = self.pos_encoder(emb)
        out = _StackFunction.apply(flat, self, None, None, emb, sep, inference)
        return self.decoder(out) if self._custom_decoder else out
