"""MLPLayer — drop-in for layers/mlp_layer.py:20-36 (Linear, LayerNorm, ReLU, Linear; same ctor, forward, state_dict).

On the GPU with the AV2 shapes (input_dim = hidden_dim = 128, output_dim <= 128) the forward is one launch of the MFMA
chain kernel ``qcn_mlp_heads_f32``; ``MLPLayer.fused`` runs up to three heads that read the same rows in one launch
(the decoder's loc / scale / pi heads).  Like the other layers it is built for the AV2 shapes only and has no CPU path.
"""
from typing import List, Sequence

import torch
import torch.nn as nn

from .. import ops
from ..utils.weight_init import weight_init


class MLPLayer(nn.Module):

    def __init__(self, input_dim: int, hidden_dim: int, output_dim: int) -> None:
        super().__init__()
        self.mlp = nn.Sequential(
            nn.Linear(input_dim, hidden_dim),
            nn.LayerNorm(hidden_dim),
            nn.ReLU(inplace=True),
            nn.Linear(hidden_dim, output_dim),
        )
        self.apply(weight_init)
        if not (input_dim == ops.C and hidden_dim == ops.C and 1 <= output_dim <= ops.C):
            raise NotImplementedError('the gfx950 kernel is built for input_dim = hidden_dim = 128, output_dim <= 128 '
                                      '(got %d, %d, %d)' % (input_dim, hidden_dim, output_dim))
        self.output_dim = output_dim

    def _head(self):
        lin1, ln, _, lin2 = self.mlp
        return (lin1.weight, lin1.bias, ln.weight, ln.bias, lin2.weight, lin2.bias)

    @staticmethod
    def fused(layers: Sequence['MLPLayer'], x: torch.Tensor, cache: dict) -> List[torch.Tensor]:
        """``[layer(x) for layer in layers]`` in one launch; ``cache``: a dict owned by the caller that keeps the pack."""
        if not 1 <= len(layers) <= 3:
            raise ValueError('one to three heads per launch')
        prm = [p for l in layers for p in l.parameters()]
        mode = ops.matmul_mode()
        key = (mode,) + tuple((p.data_ptr(), p._version) for p in prm)
        if cache.get('key') != key:
            with torch.no_grad():
                cache['pack'] = ops.pack_mlp_heads([l._head() for l in layers], mode=mode)
            cache['key'] = key
        lead = x.shape[:-1]
        ys = ops.mlp_heads(x.reshape(-1, ops.C).contiguous().float(), cache['pack'], [l.output_dim for l in layers], mode=mode)
        return [y.view(*lead, y.size(-1)) for y in ys]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not hasattr(self, '_pack_cache'):
            self._pack_cache = {}
        return MLPLayer.fused([self], x, self._pack_cache)[0]
