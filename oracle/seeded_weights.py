"""CPU ORACLE tooling — test infrastructure, never the product.

Deterministic weights keyed by parameter NAME, so that a golden fixture can name its weights by a seed instead of
carrying 12 M floats: ``oracle/make_semi_golden.py`` (build container, reference classes) and the tests (oracle / HIP
path) both call ``seeded_state_dict`` on the key -> shape table of their own module and get the same numbers.
numpy's PCG64 stream per key (seeded by crc32 of the name), float64 draws rounded once to float32; the fixture stores
``weights_digest`` of the values it was made with and the tests assert it before comparing anything.

Value rules (off the trivial values of utils/weight_init.py:17-82 on purpose — zero biases and identity LayerNorms
would hide bias / affine bugs):
  2-D ``weight`` of a Linear / GRU matrix : uniform(-a, a), a = sqrt(6 / (fan_in + fan_out)) (GRU: per 128-row gate chunk)
  ``freqs.weight``                        : N(0, 0.05)      ``*_emb.weight`` (nn.Embedding) : N(0, 0.2)
  1-D ``weight`` (LayerNorm)              : 1 + N(0, 0.1)   1-D ``bias`` / GRU biases       : N(0, 0.1)
  ``traj_emb_h0``                         : N(0, 0.1)
"""
import math
import zlib
from typing import Dict, Mapping

import numpy as np
import torch


def seeded_tensor(key: str, shape, seed: int) -> torch.Tensor:
    rng = np.random.default_rng([zlib.crc32(key.encode()), int(seed)])
    shape = tuple(int(s) for s in shape)
    leaf = key.rsplit('.', 1)[-1]
    if key.endswith('freqs.weight'):
        v = rng.standard_normal(shape) * 0.05
    elif key.endswith('_emb.weight') and len(shape) == 2:
        v = rng.standard_normal(shape) * 0.2
    elif 'traj_emb_h0' in key:
        v = rng.standard_normal(shape) * 0.1
    elif len(shape) == 2:
        fan_out, fan_in = shape
        if leaf.startswith('weight_ih') or leaf.startswith('weight_hh'):
            fan_out //= 3
        a = math.sqrt(6.0 / (fan_in + fan_out))
        v = rng.uniform(-a, a, size=shape)
    elif leaf == 'weight':
        v = 1.0 + rng.standard_normal(shape) * 0.1
    else:
        v = rng.standard_normal(shape) * 0.1
    return torch.from_numpy(np.asarray(v, dtype=np.float64).astype(np.float32))


def seeded_state_dict(template: Mapping[str, torch.Tensor], seed: int) -> Dict[str, torch.Tensor]:
    """A state_dict with the template's keys and shapes.  Keys that alias one parameter (``attn_prenorm_x_dst`` of a
    non-bipartite AttentionLayer, layers/attention_layer.py:60-61: same storage under two names) get the values of the
    ``attn_prenorm_x_src`` name, whatever order a loader visits them in."""
    by_ptr = {}
    for k, v in template.items():
        by_ptr.setdefault((v.data_ptr(), tuple(v.shape)), []).append(k)
    out = {}
    for k, v in template.items():
        canon = sorted(by_ptr[(v.data_ptr(), tuple(v.shape))])[-1] if v.numel() else k   # '..x_src..' sorts after '..x_dst..'
        out[k] = seeded_tensor(canon, v.shape, seed).to(v.dtype)
    return out


def weights_digest(sd: Mapping[str, torch.Tensor]) -> np.ndarray:
    """[sum, sum of |.|, sum of key-weighted values] in float64 over the sorted keys: changes if any value, or the
    assignment of values to names, changes."""
    tot = np.zeros(3, dtype=np.float64)
    for i, k in enumerate(sorted(sd)):
        v = sd[k].detach().cpu().double().numpy().ravel()
        tot += [v.sum(), np.abs(v).sum(), ((zlib.crc32(k.encode()) % 997) / 997.0) * v.sum()]
    return tot
