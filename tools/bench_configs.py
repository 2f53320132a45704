"""Training-step throughput of the other BASELINE.json configurations (4: BNN prior, 5: GP mixture, long sequence)."""
import contextlib, io, os, random, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transformerscandobayesianinference_amd import bar_distribution, encoders, utils as u
from transformerscandobayesianinference_amd.optim import FusedClipAdam
from transformerscandobayesianinference_amd.priors import fast_gp_mix, mlp
from transformerscandobayesianinference_amd.priors.utils import gamma_sampler_f, scaled_beta_sampler_f
from transformerscandobayesianinference_amd.streams import MicroBatchStreams
from transformerscandobayesianinference_amd.transformer import TransformerModel

dev = 'cuda'
torch.manual_seed(0); random.seed(0); np.random.seed(0)


def run(name, dl_cls, dl_kw, criterion, n_out, E, H, F, L, S, B, sampler, loss_of, steps=6, warm=2):
    with contextlib.redirect_stdout(io.StringIO()):
        dl = iter(dl_cls(num_steps=steps + warm + 4, batch_size=B, seq_len=S, device=dev, **dl_kw))
    model = TransformerModel(encoders.Linear(dl_kw['num_features'], E), n_out, E, H, F, L, 0.0, y_encoder=encoders.Linear(1, E), precision='bf16')
    model.criterion = criterion
    with torch.no_grad():
        for layer in model.transformer_encoder.layers:
            layer.linear2.weight.normal_(0, 0.02); layer.self_attn.out_proj.weight.normal_(0, 0.02)
    model.to(dev).train()
    opt = FusedClipAdam(model, lr=1e-4, max_grad_norm=1.0)
    micro = MicroBatchStreams(2)
    def step():
        sep = sampler()
        (x, y), target = next(dl)
        losses = micro.forward_backward(model, (x, y), target, sep, lambda out, tg: loss_of(out, tg[sep:]))
        opt.step(zero_grad=True)
        return losses
    for _ in range(warm): step()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(steps): l = step()
    torch.cuda.synchronize(); dt = (time.time() - t0) / steps
    print(f'{name}: {dt * 1e3:.2f} ms/step, {B / dt:.1f} datasets/s, loss {l.mean().item():.4f}, peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB')


hps = (lambda: 3, scaled_beta_sampler_f(2., 4., 150, 2), torch.nn.Tanh, gamma_sampler_f(3.6187797729244253, 0.06773738681062867),
       gamma_sampler_f(1.8663049257557085, 0.05275478076173361), lambda: 0.0, True, scaled_beta_sampler_f(1., 1.6, 60, 2),
       None, False, None, None, None, True, False, lambda n: ([], []), 0.0)
bce = torch.nn.BCEWithLogitsLoss(reduction='none')
run('config 4 (priors.mlp, bptt 1000, 60 features, E 512, L 6, BCE, batch 64)', mlp.DataLoader,
    dict(num_features=60, hyperparameters=hps, batch_size_per_gp_sample=8), bce, 1, 512, 4, 1024, 6, 1000, 64,
    u.get_uniform_single_eval_pos_sampler(1000), lambda out, tg: bce(out.squeeze(-1), tg))
mix_hp = {}
ys = fast_gp_mix.get_batch(256, 20, 18, device=dev, hyperparameters=mix_hp)[1].flatten().cpu()
crit = bar_distribution.FullSupportBarDistribution(bar_distribution.get_bucket_limits(1000, ys=ys))
run('config 5 (priors.fast_gp_mix, bptt 4000, E 1024, H 4 (d_h 256), L 12, batch 8)', fast_gp_mix.DataLoader,
    dict(num_features=18, hyperparameters=mix_hp), crit, 1000, 1024, 4, 2048, 12, 4000, 8,
    u.get_weighted_single_eval_pos_sampler(4000), lambda out, tg: crit(out.reshape(-1, 1000), tg.reshape(-1)).view(out.shape[0], -1))
crit2 = bar_distribution.FullSupportBarDistribution(bar_distribution.get_bucket_limits(1000, ys=ys))
run('config 5 with nhead 16 (d_h 64)', fast_gp_mix.DataLoader,
    dict(num_features=18, hyperparameters=mix_hp), crit2, 1000, 1024, 16, 2048, 12, 4000, 8,
    u.get_weighted_single_eval_pos_sampler(4000), lambda out, tg: crit2(out.reshape(-1, 1000), tg.reshape(-1)).view(out.shape[0], -1))
