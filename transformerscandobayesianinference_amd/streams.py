"""Concurrent micro-batches: one optimizer step's batch is split into `n` equal column groups whose forward +
loss + backward run on `n` HIP streams at the same time.

Every kernel of the encoder stack alternates a memory-bound phase (operand prologue, bias / GELU / residual /
LayerNorm epilogue, attention row loads and stores) with a matrix-core phase, and one launch occupies the whole
chip in lock-step; two independent launches in flight fill each other's phases and tails (measured on MI355X at the
north-star shape: 32 datasets as 2 x 16 on two streams 17.6 ms vs 18.9 ms on one).  Datasets are independent columns
(SURVEY.md 8(e)), every group uses the same `single_eval_pos`, and the loss of the step is the mean over equally
sized groups, so the summed gradients are those of the full batch; all gradient kernels accumulate with atomics
into the shared flat gradient buffer.  The reference has no counterpart (single stream, train.py:66-97)."""
import torch

from transformerscandobayesianinference_amd import _hip


class MicroBatchStreams:
    def __init__(self, n):
        self.n = max(1, int(n))
        self.streams = [torch.cuda.Stream() for _ in range(self.n)] if self.n > 1 and torch.cuda.is_available() else []

    def groups(self, model, batch):
        """Number of concurrent column groups (= backward passes) a batch of `batch` datasets runs as."""
        # only the all-HIP path is split: with a PyTorch-side embedding (custom encoders, SeqBN, positional encodings)
        # SeqBN would normalise per group, a scrambled encoding would draw per group, and autograd's gradient
        # accumulation into the shared flat buffer (non-atomic read-modify-write) would run on two streams at once
        # ... and so would a custom `decoder=` module: it runs in PyTorch on the stack's output, and AccumulateGrad's in-place
        # `+=` on its parameters (views of the flat buffer) is not atomic across the two streams
        # ... nor a deterministic model (PFN_SCHED_DETERMINISTIC): its gradient kernels write each element from one place, in stream order
        fused = getattr(model, '_fused_embedding', lambda: False)() and not getattr(model, '_custom_decoder', False) and not getattr(model, 'deterministic', False)
        # (the groups need not be equal: `batch % n` of them take one dataset more, each weighted by its share of the batch)
        return self.n if (self.streams and fused and batch >= 2 * self.n) else 1

    def forward_backward(self, model, data, targets, single_eval_pos, loss_fn):
        """data = (x[T,B,F], y[T,B]); targets [T,B] (already sliced to the test rows by the caller's loss_fn if needed).
        loss_fn(output, targets_group) -> per-(position, dataset) losses [T - sep, b].  Runs backward of the mean loss
        (over all groups) and returns the detached losses [T - sep, B]."""
        x, y = data
        n = self.groups(model, x.shape[1])
        if n == 1:
            output = model(data, single_eval_pos=single_eval_pos)
            losses = loss_fn(output, targets)
            losses.mean().backward()
            return losses.detach()
        main = torch.cuda.current_stream()
        model.flat_parameters()
        model._refresh_shadow(_hip.stream_ptr(x.device))      # operand copies of the weights: once, before the fork
        B = x.shape[1]
        h, rem = divmod(B, n)
        outs = []
        lo = 0
        for i, s in enumerate(self.streams[:n]):
            s.wait_stream(main)
            with torch.cuda.stream(s):
                sl = slice(lo, lo + h + (1 if i < rem else 0))
                lo = sl.stop
                output = model((x[:, sl], y[:, sl]), single_eval_pos=single_eval_pos)
                losses = loss_fn(output, targets[:, sl])
                (losses.mean() * ((sl.stop - sl.start) / B)).backward()      # the mean over the whole batch: every group by its share of the datasets
                outs.append(losses.detach())
            for t in (x, y, targets):
                t.record_stream(s)
        for s in self.streams[:n]:
            main.wait_stream(s)
        for o in outs:
            o.record_stream(main)
        return torch.cat(outs, 1)

    # ---- whole batches on alternating streams (gradient accumulation over small batches) -------------------------------------------
    # train(aggregate_k_gradients = k) sums the gradients of k consecutive batches before one optimizer step (reference train.py:92-97); the
    # notebooks train BASELINE configs[1] that way at batch_size 4 (SetupForGPFittingExperiments.ipynb:143-149).  A batch of 4 datasets is 8000
    # tokens -- a fraction of one round of tiles on 256 CUs -- and splitting it into column groups makes the launches smaller still.  The k
    # batches of one optimizer step are independent given the weights, so they run WHOLE, round-robin on the streams: up to `n` batches in flight,
    # their gradients accumulated atomically into the shared flat buffer, joined before the optimizer step.
    def can_alternate(self, model):
        fused = getattr(model, '_fused_embedding', lambda: False)() and not getattr(model, '_custom_decoder', False) and not getattr(model, 'deterministic', False)
        return bool(self.streams) and fused

    def forward_backward_on(self, slot, model, data, targets, single_eval_pos, loss_fn):
        """The whole batch on stream `slot % n`; returns the detached per-(position, dataset) losses, valid on the caller's stream after join()."""
        x, y = data
        main = torch.cuda.current_stream()
        model.flat_parameters()
        model._refresh_shadow(_hip.stream_ptr(x.device))      # operand copies of the weights: on the caller's stream, before any batch of the step reads them
        s = self.streams[slot % self.n]
        s.wait_stream(main)
        with torch.cuda.stream(s):
            output = model(data, single_eval_pos=single_eval_pos)
            losses = loss_fn(output, targets)
            losses.mean().backward()
            out = losses.detach()
        for t in (x, y, targets):
            t.record_stream(s)
        out.record_stream(main)       # allocated on `s`, read by the caller on `main` after join(): the allocator must not hand the block back to `s` early
        return out

    # ---- the batches of one optimizer step as ONE launch set per stream (ragged batch, round 5) ------------------------------------------------------
    # Alternating streams keep up to n small batches in flight, but every launch is still a small batch's (8000 tokens at batch_size 4: 128 attention
    # workgroups, <= 192 GEMM tiles on 256 CUs) and a step of 25 batches is ~7500 launches.  Here the datasets of several batches are stacked along B
    # and run as one launch set in which every dataset carries its own eval position (TransformerModel.forward_batches, pfn_stack_forward_ragged):
    # the launches are as large as a big batch's.  The batches of a step are split into `groups` contiguous chunks, one per stream.
    MAX_GROUP_DATASETS = 64      # datasets per launch set (the activation workspace grows with it: ~0.34 GB per dataset at the north-star shape)
    WORKSPACE_MEMORY_SHARE = 0.5  # ... and never more datasets than fit this share of the device's FREE memory over the streams' concurrent launch sets

    def max_group_datasets(self, model, seq_len, n_concurrent):
        """Datasets per launch set: MAX_GROUP_DATASETS, lowered until `n_concurrent` workspaces of that many datasets (pfn_workspace_bytes for this model's
        descriptor: the activations the backward re-reads) fit WORKSPACE_MEMORY_SHARE of the free device memory -- a larger model or a smaller GPU then runs
        smaller launch sets instead of running out of memory where the reference's one-batch-at-a-time loop fits."""
        import ctypes
        n = self.MAX_GROUP_DATASETS
        try:
            desc = model._operands(_hip.stream_ptr(model.flat_parameters()[0].device), False)[0]
            free, _ = torch.cuda.mem_get_info(model.flat_parameters()[0].device)
        except Exception:      # (no descriptor yet / not a cuda device: the constant stands)
            return n
        budget = self.WORKSPACE_MEMORY_SHARE * free / max(1, n_concurrent)
        while n > 1 and _hip.lib().pfn_workspace_bytes(ctypes.byref(desc), n, seq_len) > budget:
            n //= 2
        return n

    def can_stack(self, model):
        return getattr(model, 'can_forward_batches', lambda: False)()

    def forward_backward_batches(self, model, batches, loss_fn, before=None):
        """batches = [(data=(x, y), targets, single_eval_pos), ...]: the micro-batches of ONE optimizer step.  Runs backward of sum_k mean(loss_k) -- the
        reference's accumulated gradient (train.py:92-97) -- and returns the detached per-(position, dataset) losses of every batch, in order.
        loss_fn(output_k, targets_k, sep_k) -> losses [T - sep_k, b_k].  before(n_groups): called once ahead of the first launch set with the number of backward
        passes that follow (data-parallel runs arm their reducer for that many)."""
        total = sum(b[0][0].shape[1] for b in batches)
        concurrent = self.n if (self.streams and not getattr(model, 'deterministic', False)) else 1
        n_groups = max(1, -(-total // self.max_group_datasets(model, batches[0][0][0].shape[0], concurrent)))
        if self.streams and not getattr(model, 'deterministic', False):
            n_groups = max(n_groups, min(self.n, len(batches)))
        n_groups = min(n_groups, len(batches))
        bounds = [round(i * len(batches) / n_groups) for i in range(n_groups + 1)]
        groups = [batches[bounds[i]:bounds[i + 1]] for i in range(n_groups)]
        main = torch.cuda.current_stream()
        model.flat_parameters()
        dev = batches[0][0][0].device
        model._refresh_shadow(_hip.stream_ptr(dev))
        use_streams = bool(self.streams) and n_groups > 1 and not getattr(model, 'deterministic', False)
        outs = []
        if before is not None:
            before(n_groups)
        for gi, group in enumerate(groups):
            s = self.streams[gi % self.n] if use_streams else main
            if use_streams:
                s.wait_stream(main)
            with torch.cuda.stream(s):
                logits = model.forward_batches([d for d, _, _ in group], [sep for _, _, sep in group])
                losses = [loss_fn(out, tg, sep) for out, (_, tg, sep) in zip(logits, group)]
                sum(l.mean() for l in losses).backward()
                outs += [l.detach() for l in losses]
            if use_streams:
                for d, tg, _ in group:
                    for t in (d[0], d[1], tg):
                        t.record_stream(s)
        if use_streams:
            for s in self.streams:
                main.wait_stream(s)
            for o in outs:
                o.record_stream(main)
        return outs

    def fence_others(self, slot):
        """Events behind everything enqueued so far on the streams that will NOT run batch `slot` (earlier batches on the stream that will are ordered
        before it by the stream itself)."""
        evs = []
        for i, s in enumerate(self.streams):
            if i != slot % self.n:
                ev = torch.cuda.Event()
                ev.record(s)
                evs.append(ev)
        return evs

    def join(self):
        main = torch.cuda.current_stream()
        for s in self.streams:
            main.wait_stream(s)
