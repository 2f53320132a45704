"""GP prior sampler: x ~ U[0,1)^{B x T x F},  y_b ~ N(0, outputscale * RBF_lengthscale(x_b, x_b) + noise * I).

Replaces reference priors/fast_gp.py `get_batch` (:35-58), whose numerics live in gpytorch 1.5.0
(ExactGP in prior mode: ConstantMean(0) + ScaleKernel(RBFKernel) + GaussianLikelihood, :13-32,
:53-56).  Here the whole draw is one call into libpfn_hip.so (`pfn_gp_prior_sample`): uniform
features and base normals from a counter-based generator, Gram matrix, blocked f32 Cholesky and
the L.z product, batched over the B datasets.  The sampler always uses the exact Cholesky root --
what the reference notebook forces with `'fast_computations': (False, False, False)`
(SetupForGPFittingExperiments.ipynb:140); gpytorch's Lanczos shortcut for T > 800 is deliberately
not reproduced.  No CPU fallback.
"""
import torch

from transformerscandobayesianinference_amd import _hip
from transformerscandobayesianinference_amd.priors.utils import defer_draw_check, get_batch_to_dataloader
from transformerscandobayesianinference_amd.utils import default_device

KERNEL_RBF, KERNEL_MATERN52, KERNEL_MATERN32, KERNEL_MATERN12 = 0, 1, 2, 3
MATERN_KERNEL_OF_NU = {2.5: KERNEL_MATERN52, 1.5: KERNEL_MATERN32, 0.5: KERNEL_MATERN12}    # gpytorch MaternKernel(nu=...) accepts exactly these

_DEFAULT_HPS = {"noise": .1, "outputscale": .1, "lengthscale": .1}  # reference fast_gp.py:40
_call_counter = [0]
# gpytorch.utils.cholesky.psd_safe_cholesky (what ExactGP's prior draw and prediction go through, reference :53-56,
# :101-104): on a failed factorisation the diagonal is raised by 1e-6, 1e-5, 1e-4 in turn, with a warning, and
# NotPSDError is raised after the third failure (gpytorch 1.5.0 settings.cholesky_jitter = 1e-6 for float,
# cholesky_max_tries = 3; restated from memory of the pinned version -- the library is not installed here).
CHOLESKY_JITTERS = (1e-6, 1e-5, 1e-4)
repair_log = []   # (number of failed datasets, jitter that fixed them) of every repaired draw: tests and monitoring read it


class NotPSDError(RuntimeError):
    """The covariance matrix of a dataset is not positive definite in f32 even with the largest diagonal jitter."""


def _with_jitter(run, noise, info, what):
    """`run(noise_vector)` refills the outputs of the failed datasets; called with the info flags of a finished attempt.
    Re-runs with escalating diagonal jitter on the failed datasets until every factorisation succeeds; returns True if
    anything had to be repaired."""
    import warnings
    bad = info != 0
    if not bool(bad.any()):
        return False
    nbad = int(bad.sum())
    for jitter in CHOLESKY_JITTERS:
        warnings.warn(f'{what}: f32 Cholesky failed for {nbad} dataset(s); retrying those with {jitter:g} added to the diagonal '
                      f'(gpytorch psd_safe_cholesky behaviour)', RuntimeWarning)
        info2 = run(noise + jitter * bad.to(noise.dtype), bad)
        still = (info2 != 0) & bad
        if not bool(still.any()):
            repair_log.append((nbad, jitter))
            return True
        bad = still
    raise NotPSDError(f'{what}: covariance matrix of {int(bad.sum())} dataset(s) not positive definite after adding {CHOLESKY_JITTERS[-1]:g} to the diagonal')


def gp_sample(batch_size, seq_len, num_features, device, lengthscale, outputscale, noise, kernel=KERNEL_RBF,
              x=None, z=None, seed=None, check=True):
    """Batched draw through the C ABI.  lengthscale: float | [B] | [B,F]; outputscale, noise: float | [B].
    x ([B,T,F]) and z ([B,T] base normals) may be injected for parity tests; otherwise they come from
    the device generator, seeded from torch's global seed plus a per-call counter.
    check: the factorisation's failure flag is read (inside a prefetching loader: before the batch is handed out,
    priors/utils.py; otherwise here, which synchronises) and failed datasets are redrawn from the same (x, z) with
    gpytorch's jitter ladder; False leaves `info` to the caller.
    Returns (x[B,T,F], y[B,T], z[B,T], info[B])."""
    dev = torch.device(device)
    if dev.type != 'cuda':
        raise _hip.HipExtensionError(f'the GP prior sampler runs on the GPU only (got device {device}); no CPU fallback')
    lib = _hip.lib()
    B, T, F = batch_size, seq_len, num_features

    def per_dataset(v, shape):
        t = torch.as_tensor(v, dtype=torch.float32, device=dev)
        return t.expand(shape).contiguous() if t.dim() == 0 else t.reshape(B, -1).expand(shape).contiguous() if len(shape) == 2 else t.reshape(B).contiguous()

    ls = per_dataset(lengthscale, (B, F))
    osc = per_dataset(outputscale, (B,))
    nz = per_dataset(noise, (B,))
    gen_x, gen_z = x is None, z is None
    # the kernels want 16-byte aligned matrix rows (T % 4 == 0): draw the GP at up to 3 extra points and drop them --
    # the marginal of a GP on the first T points is unchanged, and with L lower triangular (L z)[:T] only sees z[:T]
    Tp = (T + 3) // 4 * 4
    if Tp != T:
        if not gen_x:
            x = torch.cat([x.to(dev).float(), torch.rand(B, Tp - T, F, device=dev)], 1)
        if not gen_z:
            z = torch.cat([z.to(dev).float(), torch.zeros(B, Tp - T, device=dev)], 1)
    x = torch.empty(B, Tp, F, dtype=torch.float32, device=dev) if gen_x else x.to(dev).float().contiguous()
    z = torch.empty(B, Tp, dtype=torch.float32, device=dev) if gen_z else z.to(dev).float().contiguous()
    y = torch.empty(B, Tp, dtype=torch.float32, device=dev)
    K = torch.empty(lib.pfn_gp_workspace_bytes(B, Tp), dtype=torch.uint8, device=dev)      # the [B, Tp, Tp] f32 matrix + the trailing update's plane scratch
    info = torch.zeros(B, dtype=torch.int32, device=dev)
    if seed is None:
        seed = torch.initial_seed()
    _call_counter[0] += 1
    offset = _call_counter[0]

    def run(xs, zs, lss, oscs, noise_vec, y_out, info_out, gx, gz, K_ws):
        _hip.check(lib.pfn_gp_prior_sample(xs.data_ptr(), zs.data_ptr(), y_out.data_ptr(), K_ws.data_ptr(), K_ws.numel() * K_ws.element_size(), lss.data_ptr(), oscs.data_ptr(),
                                           noise_vec.data_ptr(), xs.shape[0], Tp, F, kernel, int(gx), int(gz), seed & (2 ** 64 - 1),
                                           offset, info_out.data_ptr(), _hip.stream_ptr(dev)), 'pfn_gp_prior_sample')

    run(x, z, ls, osc, nz, y, info, gen_x, gen_z, K)
    del K
    if check:
        def retry(noise_vec, bad):
            # the FAILED datasets only, from the same (x, z) -- they are on the device now -- with a raised diagonal: the workspace of
            # the retry is 4 Tp^2 bytes per failed dataset, not another full batch's
            idx = bad.nonzero(as_tuple=True)[0]
            n = idx.numel()
            y2 = torch.empty(n, Tp, dtype=torch.float32, device=dev)
            info_sub = torch.zeros(n, dtype=torch.int32, device=dev)
            run(x[idx].contiguous(), z[idx].contiguous(), ls[idx].contiguous(), osc[idx].contiguous(), noise_vec[idx].contiguous(), y2, info_sub,
                False, False, torch.empty(lib.pfn_gp_workspace_bytes(n, Tp), dtype=torch.uint8, device=dev))
            y[idx] = y2
            info2 = torch.zeros_like(info)
            info2[idx] = info_sub
            return info2

        verify = lambda: _with_jitter(retry, nz, info, 'GP prior draw')
        if check == 'sync' or not defer_draw_check(verify):
            verify()
    if Tp != T:
        # y stays a view of the padded draw so that a deferred repair (which writes y in place) reaches the consumer
        x, y, z = x[:, :T], y[:, :T], z[:, :T]
    return x, y, z, info


@torch.no_grad()
def get_batch(batch_size, seq_len, num_features, device=default_device, hyperparameters=None, equidistant_x=False):
    """Same signature and return layout as the reference: (x[T,B,F], y[T,B], target_y[T,B])."""
    if isinstance(hyperparameters, (tuple, list)):
        hyperparameters = {"noise": hyperparameters[0], "outputscale": hyperparameters[1], "lengthscale": hyperparameters[2]}
    elif hyperparameters is None:
        hyperparameters = dict(_DEFAULT_HPS)
    x = None
    if equidistant_x:
        assert num_features == 1
        x = torch.linspace(0, 1., seq_len).unsqueeze(0).repeat(batch_size, 1).unsqueeze(-1)
    noise = max(float(hyperparameters["noise"]), 1e-9)  # GaussianLikelihood(noise_constraint=GreaterThan(1e-9)), reference :25
    x, y, _, _ = gp_sample(batch_size, seq_len, num_features, device, hyperparameters["lengthscale"],
                           hyperparameters["outputscale"], noise, KERNEL_RBF, x=x)
    sample = y.transpose(0, 1)
    return x.transpose(0, 1), sample, sample


DataLoader = get_batch_to_dataloader(get_batch)
DataLoader.num_outputs = 1


@torch.no_grad()
def gp_posterior(x, y, lengthscale, outputscale, noise, kernel=KERNEL_RBF, check=True):
    """Sequential exact-GP predictions through the C ABI (`pfn_gp_posterior`): for every dataset b and position t the
    posterior at x[b,t] given (x[b,:t], y[b,:t]).  x [B,T,F], y [B,T] on the GPU; hyper-parameters as in `gp_sample`.
    check: read the factorisation's failure flags (a host sync) and redo failed datasets with gpytorch's jitter ladder.
    Returns (mean[B,T], var[B,T] with observation noise, nll[B,T], info[B] of the first attempt)."""
    dev = x.device
    if dev.type != 'cuda':
        raise _hip.HipExtensionError(f'the GP posterior runs on the GPU only (got device {dev}); no CPU fallback')
    lib = _hip.lib()
    B, T, F = x.shape

    def per_dataset(v, cols):
        t = torch.as_tensor(v, dtype=torch.float32, device=dev)
        if t.dim() == 0:
            return t.expand(B, cols).contiguous()
        return t.reshape(B, -1).expand(B, cols).contiguous()

    ls, osc, nz = per_dataset(lengthscale, F), per_dataset(outputscale, 1), per_dataset(noise, 1)
    # 16-byte aligned matrix rows: append up to 3 points AFTER the real ones -- a prediction only sees earlier points
    Tp = (T + 3) // 4 * 4
    xp, yp = x.float(), y.float()
    if Tp != T:
        xp = torch.cat([xp, torch.rand(B, Tp - T, F, device=dev)], 1)
        yp = torch.cat([yp, torch.zeros(B, Tp - T, device=dev)], 1)
    xp, yp = xp.contiguous(), yp.contiguous()
    K = torch.empty(lib.pfn_gp_workspace_bytes(B, Tp), dtype=torch.uint8, device=dev)
    resid, w = torch.empty_like(yp), torch.empty_like(yp)

    def run(noise_vec):
        mean, var, nll = torch.empty_like(yp), torch.empty_like(yp), torch.empty_like(yp)
        info = torch.zeros(B, dtype=torch.int32, device=dev)
        _hip.check(lib.pfn_gp_posterior(xp.data_ptr(), yp.data_ptr(), K.data_ptr(), K.numel() * K.element_size(), resid.data_ptr(), w.data_ptr(), ls.data_ptr(),
                                        osc.data_ptr(), noise_vec.data_ptr(), B, Tp, F, kernel, nll.data_ptr(), mean.data_ptr(),
                                        var.data_ptr(), info.data_ptr(), _hip.stream_ptr(dev)), 'pfn_gp_posterior')
        return mean, var, nll, info

    mean, var, nll, info = run(nz)
    if check:
        def retry(noise_vec, bad):
            m2, v2, n2, info2 = run(noise_vec.contiguous())
            mean[bad], var[bad], nll[bad] = m2[bad], v2[bad], n2[bad]
            return info2

        _with_jitter(retry, nz.reshape(B), info, 'exact-GP posterior')
    return mean[:, :T], var[:, :T], nll[:, :T], info


def _hyperparameters_dict(hyperparameters):
    if isinstance(hyperparameters, (tuple, list)):
        return {"noise": hyperparameters[0], "outputscale": hyperparameters[1], "lengthscale": hyperparameters[2]}
    hps = dict(_DEFAULT_HPS)
    hps.update(hyperparameters or {})
    return hps


@torch.no_grad()
def evaluate(x, y, y_non_noisy, use_mse=False, hyperparameters={}, get_model_on_device=None, device=default_device,
             step_size=1, start_pos=0):
    """The exact-GP baseline of the reference (priors/fast_gp.py:88-120): for t in range(max(start_pos, 1), T, step_size)
    the loss of the GP posterior given the first t points at point t -- negative log density of y[t] under the noisy
    predictive, or squared error of its mean.  Same arguments (x [T,B,F], y [T,B]) and the same return value
    (losses [n_t, B] on the CPU, their means per t with the reference's leading 0. when start_pos == 0, seconds).
    The reference refits a gpytorch model per t; here every t comes from one batched factorisation on the GPU
    (`gp_posterior`), with the exact Cholesky the notebook selects (`fast_computations` off).  `get_model_on_device`
    exists for signature compatibility; a fitted-hyper-parameter model has no meaning here and is rejected."""
    import time
    if get_model_on_device is not None:
        raise NotImplementedError('evaluate() runs the fixed-hyper-parameter GP of priors.fast_gp; pass hyperparameters instead')
    start_time = time.time()
    hps = _hyperparameters_dict(hyperparameters)
    xb = x.to(device).transpose(0, 1).contiguous()
    yb = y.to(device).transpose(0, 1).contiguous()
    mean, var, nll, _ = gp_posterior(xb, yb, hps["lengthscale"], hps["outputscale"], max(float(hps["noise"]), 1e-9))
    ts = list(range(max(start_pos, 1), x.shape[0], step_size))
    idx = torch.as_tensor(ts, dtype=torch.long, device=mean.device)
    ls = ((mean - yb) ** 2 if use_mse else nll).index_select(1, idx).transpose(0, 1)     # [n_t, B]
    per_t = ls.mean(1)
    if start_pos == 0:
        per_t = torch.cat([per_t.new_zeros(1), per_t])
    if mean.is_cuda:
        torch.cuda.synchronize(mean.device)
    return ls.to('cpu'), per_t.to('cpu'), time.time() - start_time
DataLoader.prefetch = True        # draws run ahead of the training steps on a side stream (priors/utils.py)
DataLoader.prefetch_group = 10    # ... ten steps' worth of datasets per sampler call (MI355X, bptt 2000: 64 us per dataset at 4 x 32, 54 us at 10 x 32;
                                  # 16 MB of matrix + 3.7 MB of plane scratch per dataset at bptt 2000: 6.3 GB of the 288)
DataLoader.prefetch_group_datasets = 640    # ... and at least this many datasets per call when the batches are small (priors/utils.py)
DataLoader.prefetch_memory_share = 0.125    # ... but never more than an eighth of the free device memory per group (two groups are alive at a time)
def workspace_bytes_per_dataset(kw):
    """K_ws per dataset as the library sizes it (the [Tp, Tp] f32 matrix + the plane scratch -- two plane sets of two fp16 planes since round 5: +23 % at bptt 2000, +39 % at 1000, +50 % at 512; ADVICE r4 / r5: not a constant factor, always asked of pfn_gp_workspace_bytes)."""
    Tp = (kw.get('seq_len', 0) + 3) // 4 * 4
    return int(_hip.lib().pfn_gp_workspace_bytes(1, Tp)) if Tp > 0 else 0


DataLoader.prefetch_bytes_per_dataset = staticmethod(workspace_bytes_per_dataset)
