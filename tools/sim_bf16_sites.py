"""Per-operand-class sensitivity of the bf16 training forward (analysis tool; imports the oracle).  See tools/sim_bf16_rounding.py and profiles/r05_bf16_rounding_simulation.json."""
import sys, math, json, time, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from oracle import pfn_oracle as O
import sim_bf16_rounding as S
r16=S.r16
def forward(sd,x,y,sep,nhead,sites):
    dt=torch.float64
    p={k:v.detach().to(dt) for k,v in sd.items()}
    x,y=x.to(dt),y.to(dt); T,B,_=x.shape
    W=lambda w: r16(w) if 'W' in sites else w
    X=lambda a: r16(a) if 'X' in sites else a
    h=torch.cat([O._linear(x,p['encoder.weight'],p['encoder.bias'])[:sep]+O._linear(y.unsqueeze(-1),p['y_encoder.weight'],p['y_encoder.bias'])[:sep],O._linear(x,p['encoder.weight'],p['encoder.bias'])[sep:]],0)
    E=h.shape[-1]; D=E//nhead; mask=O.d_q_mask(T,sep,dt,h.device)
    L=1+max(int(k.split('.')[2]) for k in p if k.startswith('transformer_encoder.layers.'))
    for l in range(L):
        pre=f'transformer_encoder.layers.{l}.'
        qkv=O._linear(X(h),W(p[pre+'self_attn.in_proj_weight']),p[pre+'self_attn.in_proj_bias'])
        if 'QKV' in sites: qkv=r16(qkv)
        q,k,v=[t.reshape(T,B,nhead,D).permute(1,2,0,3) for t in qkv.split(E,-1)]
        s=q@k.transpose(-1,-2)/math.sqrt(D)+mask
        mx=s.amax(-1,keepdim=True); pu=torch.exp(s-mx)
        ctx=((r16(pu) if 'P' in sites else pu)@v)/pu.sum(-1,keepdim=True)
        ctx=ctx.permute(2,0,1,3).reshape(T,B,E)
        if 'CTX' in sites: ctx=r16(ctx)
        h=O._layer_norm(h+O._linear(ctx,W(p[pre+'self_attn.out_proj.weight']),p[pre+'self_attn.out_proj.bias']),p[pre+'norm1.weight'],p[pre+'norm1.bias'])
        act=O._gelu(O._linear(X(h),W(p[pre+'linear1.weight']),p[pre+'linear1.bias']))
        if 'ACT' in sites: act=r16(act)
        h=O._layer_norm(h+O._linear(act,W(p[pre+'linear2.weight']),p[pre+'linear2.bias']),p[pre+'norm2.weight'],p[pre+'norm2.bias'])
    d=O._gelu(O._linear(X(h[sep:]),W(p['decoder.0.weight']),p['decoder.0.bias']))
    if 'ACT' in sites: d=r16(d)
    return O._linear(d,W(p['decoder.2.weight']),p['decoder.2.bias'])
import bench
from transformerscandobayesianinference_amd import bar_distribution, encoders
from transformerscandobayesianinference_amd.transformer import TransformerModel
w=dict(bench.CONFIGS[2],parity_batch=2); sep=1755
torch.manual_seed(0)
m=TransformerModel(encoders.Linear(w['num_features'],w['emsize']),w['num_bars'],w['emsize'],w['nhead'],w['nhid'],w['nlayers'],0.0,y_encoder=encoders.Linear(1,w['emsize']))
with torch.no_grad():
    for layer in m.transformer_encoder.layers:
        layer.linear2.weight.normal_(0,0.02); layer.self_attn.out_proj.weight.normal_(0,0.02)
sd={k:v.detach().clone() for k,v in m.state_dict().items()}
x,y=bench.parity_inputs(w,torch.device('cpu'))
g=torch.Generator().manual_seed(7)
ys=O.get_batch_fast_gp(2000,20,w['num_features'],w['hyperparameters'],g,dtype=torch.float64)[1]
borders=bar_distribution.get_bucket_limits(w['num_bars'],ys=ys.float()).double()
rel=lambda a,b:((a-b).norm()/b.norm()).item()
exact=forward(sd,x,y,sep,w['nhead'],set())
mo=O.bar_mean(exact,borders)
ALL={'W','X','QKV','P','CTX','ACT'}
for name,sites in [('all',ALL)]+[('only '+s,{s}) for s in sorted(ALL)]+[('all but '+s,ALL-{s}) for s in sorted(ALL)]:
    lg=forward(sd,x,y,sep,w['nhead'],sites)
    print(f'{name:12s} logits {rel(lg,exact):.2e} means {rel(O.bar_mean(lg,borders),mo):.2e}',flush=True)
