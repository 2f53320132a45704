import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from transformerscandobayesianinference_amd import _hip, bar_distribution, encoders
from transformerscandobayesianinference_amd.transformer import TransformerModel
DEV='cuda:0'
cfg = dict(T=300, F=5, E=128, H=2, nhid=256, L=3, nbars=40)
def build(schedule=None):
    torch.manual_seed(21)
    borders = torch.sort(torch.randn(cfg['nbars'] + 1) * 1.5)[0]
    m = TransformerModel(encoders.Linear(cfg['F'], cfg['E']), cfg['nbars'], cfg['E'], cfg['H'], cfg['nhid'], cfg['L'], 0.0, y_encoder=encoders.Linear(1, cfg['E']), precision='f32', eval_precision='f32')
    with torch.no_grad():
        for layer in m.transformer_encoder.layers:
            layer.linear2.weight.normal_(0, 0.05); layer.self_attn.out_proj.weight.normal_(0, 0.05)
    m.schedule = schedule
    return m.to(DEV).train()
g = torch.Generator().manual_seed(4)
widths, seps = [4, 1, 3, 4, 2], [257, 290, 80, 131, 299]
batches = [(torch.rand(cfg['T'], w, cfg['F'], generator=g).to(DEV), torch.randn(cfg['T'], w, generator=g).to(DEV)) for w in widths]
m_top, m_all = build(None), build(_hip.SCHED_TOP_LAYER_ALL_ROWS)
with torch.no_grad():
    for name, m in (('top', m_top), ('all', m_all)):
        outs = m.forward_batches(batches, seps)
        for k, ((x, y), sep) in enumerate(zip(batches, seps)):
            for name2, m2 in (('top', m_top), ('all', m_all)):
                ref = m2((x, y), single_eval_pos=sep)
                d = (outs[k] - ref).abs()
                print(f'ragged[{name}] vs uniform[{name2}] sep {sep}: equal {torch.equal(outs[k], ref)} max {d.max().item():.3e} rows differing {(d.amax(-1) > 0).sum().item()} of {d.shape[0] * d.shape[1]}', 'first rows t:', (d.amax(-1).amax(1) > 0).nonzero().flatten()[:8].tolist())
    # uniform top vs uniform all
    for (x, y), sep in zip(batches, seps):
        a, b = m_top((x, y), single_eval_pos=sep), m_all((x, y), single_eval_pos=sep)
        print('uniform top vs all sep', sep, torch.equal(a, b), (a - b).abs().max().item())
print('---- isolation ----')
with torch.no_grad():
    ref = {sep: m_all((x, y), single_eval_pos=sep) for (x, y), sep in zip(batches, seps)}
    def run(idx, tag):
        outs = m_all.forward_batches([batches[i] for i in idx], [seps[i] for i in idx])
        print(tag, [(seps[i], bool(torch.equal(o, ref[seps[i]])), f'{(o - ref[seps[i]]).abs().max().item():.2e}') for i, o in zip(idx, outs)])
    run([2], 'only the sep-80 batch')
    run([2, 3], 'sep 80 first, then 131')
    run([3, 2], 'sep 131 first, then 80')
    run([0, 2], '257 then 80')
    run([2, 2], '80 twice')
    # a uniform call on a wider batch containing the same datasets at another batch index
    x80, y80 = batches[2]
    xb = torch.cat([batches[3][0][:, :2], x80], 1); yb = torch.cat([batches[3][1][:, :2], y80], 1)
    o = m_all((xb, yb), single_eval_pos=80)[:, 2:]
    print('uniform call, the sep-80 datasets at batch index 2..4:', torch.equal(o, ref[80]), (o - ref[80]).abs().max().item())
m_bf = build(None); 
