"""Debug aid: per-call comparison of product (GPU) vs oracle (CPU) layer outputs for one QCNet forward."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _bootstrap; _bootstrap.ensure_package()
import torch
from oracle import qcnet_oracle as O
from tempenslearnagg_qcnet_amd.predictors import TempEnsLearnAgg
from tempenslearnagg_qcnet_amd.layers import AttentionLayer, FourierEmbedding
from tempenslearnagg_qcnet_amd.synthetic import config_scene, make_batch
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import randomize_, cpu_state_dict, rel_err

cfg = config_scene('tiny')
torch.manual_seed(0)
model = randomize_(TempEnsLearnAgg(**cfg['hparams']), seed=1).eval()
randomize_(model.temporal_aggregate_layer, seed=2)
sd = cpu_state_dict(model)
batch = make_batch([0, 1, 2], **cfg['scene'])
ts = int(sys.argv[1]) if len(sys.argv) > 1 else 10

log_o = []
_al, _fe = O.attention_layer, O.fourier_embedding
def al(sd_, p, x, r, ei, *a, **k):
    out = _al(sd_, p, x, r, ei, *a, **k); log_o.append((p, out)); return out
def fe(sd_, p, c, cats=None):
    out = _fe(sd_, p, c, cats); log_o.append((p, out)); return out
O.attention_layer, O.fourier_embedding = al, fe
with torch.no_grad():
    want_pred, want_enc = O.qcnet_forward(sd, batch, ts, cfg['hparams'], p='base_model.')

log_p = []
names = {m: n for n, m in model.named_modules()}
def hook(m, inp, out):
    log_p.append((names[m] + '.', out.detach().cpu()))
for m in model.modules():
    if isinstance(m, (AttentionLayer, FourierEmbedding)):
        m.register_forward_hook(hook)
model = model.to('cuda:0')
with torch.no_grad():
    got_pred, got_enc = model(batch.to('cuda:0'), ts)
print(len(log_o), len(log_p))
for (po, o), (pp, p) in zip(log_o, log_p):
    e = rel_err(p, o) if o.shape == p.shape else float('nan')
    # row-wise worst
    worst = ''
    if o.shape == p.shape and e > 1e-4:
        d = (p.double() - o.double()).abs().max(dim=-1).values
        idx = torch.nonzero(d > 1e-3 * float(o.abs().max())).flatten()
        worst = ' bad rows %d/%d first %s' % (idx.numel(), d.numel(), idx[:12].tolist())
    print('%-70s %-12s %.3e%s' % (po, tuple(o.shape), e, worst))
