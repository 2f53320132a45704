"""Evaluation sweeps over the number of training points -- the two curves of the paper's
GP-fitting figure: the PFN's loss at evaluation position `eval_pos` and the exact-GP baseline at the same positions.

`run_test` mirrors the notebook function of that name (`SetupForGPFittingExperiments.ipynb` cell 6, :176-224): for
every evaluation position it draws fresh datasets of `eval_pos + 1` points from the GP prior, runs the forward-only
path of the HIP stack with the last point as the single test row, and reports the bar NLL, the squared error of the
posterior-predictive mean and of the mode.  Same arguments, same 5-tuple.  Every position has its own train set, so
there is nothing to cache across positions; within a pass the test row attends to the train keys once.

`gp_baseline` is what the notebook obtains from `priors.fast_gp.evaluate`: the exact-GP posterior on the same prior,
here from `priors.fast_gp.gp_posterior` -- all positions of a draw from one factorisation.
"""
import numpy as np
import scipy.stats as st
import torch
from torch import nn

from transformerscandobayesianinference_amd.priors import fast_gp

GP_FITTING_HPS = {'noise': 1e-4, 'outputscale': 1., 'lengthscale': .6}   # the notebook's `hps` (cell 6)


def compute_mean_and_conf_interval(values, confidence=.95):
    """Mean and half-width of the Student-t confidence interval (notebook cell 6)."""
    values = np.asarray(values, dtype=np.float64)
    n = len(values)
    half = st.sem(values) * st.t.ppf((1 + confidence) / 2., n - 1)
    return values.mean(), half


@torch.no_grad()
def run_test(model, device='cuda:0', step_size=100, start_pos=1, batch_size=1000, sub_batch_size=10, seq_len=2000,
             num_features=5, hyperparameters=None, get_batch=fast_gp.get_batch):
    """Returns (eval_positions, mean_mse [n], mode_mse [n], nll [n], nll_confidence [n]) as CPU tensors.
    `sub_batch_size` only bounds memory, as in the notebook; the HIP path is happy with hundreds of datasets per pass."""
    assert batch_size % sub_batch_size == 0
    hyperparameters = dict(GP_FITTING_HPS) if hyperparameters is None else hyperparameters
    model.to(device)
    was_training = model.training
    model.eval()
    criterion = model.criterion
    gaussian = isinstance(criterion, nn.GaussianNLLLoss)
    eval_positions, mses, mode_mses, nlls, confidences = [], [], [], [], []
    for eval_pos in range(start_pos, seq_len, step_size):
        eval_positions.append(eval_pos)
        nll, mean_mse, mode_mse = [], [], []
        for _ in range(batch_size // sub_batch_size):
            x, y, target_y = get_batch(batch_size=sub_batch_size, seq_len=eval_pos + 1, num_features=num_features,
                                       hyperparameters=hyperparameters, device=device)
            logits = model((x, y), single_eval_pos=eval_pos)           # [1, sub_batch, n_out]
            target = target_y[eval_pos]
            if gaussian:
                nll.append(criterion(logits[0][..., 0], target, var=logits[0][..., 1].abs()))
                mean_mse.append(torch.zeros((), device=logits.device))
                mode_mse.append(torch.zeros((), device=logits.device))
                continue
            nll.append(criterion(logits[0], target))
            means = criterion.mean(logits)
            top = logits.argmax(-1)
            modes = (criterion.borders[top] + criterion.borders[top + 1]) / 2      # bucket centre, as in the notebook
            mean_mse.append(((means[0] - target) ** 2).mean())
            mode_mse.append(((modes[0] - target) ** 2).mean())
        nll = torch.cat(nll)
        nlls.append(nll.mean())
        mses.append(torch.stack(mean_mse).mean())
        mode_mses.append(torch.stack(mode_mse).mean())
        confidences.append(compute_mean_and_conf_interval(nll.cpu().numpy())[1])
    model.train(was_training)
    return (eval_positions, torch.stack(mses).cpu(), torch.stack(mode_mses).cpu(), torch.stack(nlls).cpu(),
            torch.tensor(confidences))


@torch.no_grad()
def gp_baseline(device='cuda:0', step_size=100, start_pos=1, batch_size=1000, sub_batch_size=100, seq_len=2000,
                num_features=5, hyperparameters=None):
    """The exact-GP curve on the same prior: (eval_positions, mean_mse [n], nll [n], nll_confidence [n]).  One draw of
    `seq_len` points per dataset serves every evaluation position (the posterior at point t given points 0..t-1)."""
    assert batch_size % sub_batch_size == 0
    hps = dict(GP_FITTING_HPS) if hyperparameters is None else fast_gp._hyperparameters_dict(hyperparameters)
    positions = list(range(start_pos, seq_len, step_size))
    idx = torch.as_tensor(positions, dtype=torch.long, device=device)
    nll, sq = [], []
    for _ in range(batch_size // sub_batch_size):
        x, y, _ = fast_gp.get_batch(sub_batch_size, seq_len, num_features, device=device, hyperparameters=hps)
        xb, yb = x.transpose(0, 1).contiguous(), y.transpose(0, 1).contiguous()
        mean, _, nl, _ = fast_gp.gp_posterior(xb, yb, hps['lengthscale'], hps['outputscale'], max(float(hps['noise']), 1e-9))
        nll.append(nl.index_select(1, idx))
        sq.append(((mean - yb) ** 2).index_select(1, idx))
    nll, sq = torch.cat(nll), torch.cat(sq)                                   # [batch, n_positions]
    conf = [compute_mean_and_conf_interval(nll[:, j].cpu().numpy())[1] for j in range(len(positions))]
    return positions, sq.mean(0).cpu(), nll.mean(0).cpu(), torch.tensor(conf)
