"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- the reference's model as the reference BUILDS it, for the CPU baseline.

The reference's `TransformerModel` (transformer.py:14-25) is not its own arithmetic: it instantiates torch's
`nn.TransformerEncoder(nn.TransformerEncoderLayer(ninp, nhead, nhid, dropout, activation='gelu'), nlayers)` (:17-18) and calls it
with a dense additive [S, S] float mask built on the host in every forward (:34-41, :65), embeds x and y on ALL rows (:68-69) and
decodes ALL rows before slicing the test rows (:85, :91).  The explicit-math port in `pfn_oracle.py` is the parity checker; timing
it understates the reference, because torch's own modules take fused CPU paths (scaled_dot_product_attention, fused LayerNorm /
GELU kernels) that the written-out softmax does not.  This module assembles exactly the torch modules the reference constructs, so
`bench.py`'s `cpu_baseline` can time what a user of the reference actually runs on the host.  `/root/reference` does not exist on
the GPU box, so the class is restated here (same sub-module names => same state-dict keys, loadable from a reference checkpoint).

Checked against the port in tests/test_oracle.py::test_torch_modules_model_matches_the_port.
"""
import torch
from torch import nn


class ReferenceModelOnTorchModules(nn.Module):
    def __init__(self, num_features, n_out, ninp, nhead, nhid, nlayers):
        super().__init__()
        layer = nn.TransformerEncoderLayer(ninp, nhead, nhid, 0.0, activation='gelu')       # transformer.py:17 (post-norm, seq-first)
        self.transformer_encoder = nn.TransformerEncoder(layer, nlayers, enable_nested_tensor=False)   # :18
        self.encoder = nn.Linear(num_features, ninp)                                        # encoders.Linear
        self.y_encoder = nn.Linear(1, ninp)
        self.decoder = nn.Sequential(nn.Linear(ninp, nhid), nn.GELU(), nn.Linear(nhid, n_out))   # :23

    @staticmethod
    def d_q_mask(sz, sep):
        """generate_D_q_matrix(sz, sz - sep) (:34-41): every query sees the keys [0, sep) and itself; 0 / -inf floats."""
        allowed = torch.zeros(sz, sz, dtype=torch.bool)
        allowed[:, :sep] = True
        allowed |= torch.eye(sz, dtype=torch.bool)
        return torch.zeros(sz, sz).masked_fill(~allowed, float('-inf'))

    def forward(self, src, single_eval_pos):
        x, y = src
        mask = self.d_q_mask(len(x), single_eval_pos).to(x.device)                          # :65 -- rebuilt every forward, as there
        x_src = self.encoder(x)                                                             # :68
        y_src = self.y_encoder(y.unsqueeze(-1))                                             # :69 (all rows)
        h = torch.cat([x_src[:single_eval_pos] + y_src[:single_eval_pos], x_src[single_eval_pos:]], 0)   # :73-74
        out = self.decoder(self.transformer_encoder(h, mask))                               # :84-85 (decoder on all rows)
        return out[single_eval_pos:]                                                        # :91


def from_state_dict(sd, nhead):
    """Module with the weights of a reference-format state dict (criterion.* buffers are ignored)."""
    sd = {k: v.detach().float() for k, v in sd.items() if not k.startswith('criterion.')}
    ninp = sd['encoder.weight'].shape[0]
    nlayers = 1 + max(int(k.split('.')[2]) for k in sd if k.startswith('transformer_encoder.layers.'))
    m = ReferenceModelOnTorchModules(sd['encoder.weight'].shape[1], sd['decoder.2.weight'].shape[0], ninp, nhead,
                                     sd['transformer_encoder.layers.0.linear1.weight'].shape[0], nlayers)
    m.load_state_dict(sd)
    return m
