"""FourierEmbedding — drop-in for layers/fourier_embedding.py:23-67 (same ctor, forward and state_dict keys).

The forward is ONE HIP kernel (``qcn_fourier_embed_f32``): Fourier features, the per-dimension MLPs with their
LayerNorms, the sum with the categorical embeddings and ``to_out`` are fused, with the contractions on the fp32 matrix
cores.  The nn.Linear / nn.LayerNorm sub-modules only hold the parameters (reference names); their weights are packed
into the kernel's chain layout once and re-packed when a parameter changes.
"""
from typing import List, Optional

import torch
import torch.nn as nn

from .. import ops
from ..utils.weight_init import weight_init


class FourierEmbedding(nn.Module):

    def __init__(self, input_dim: int, hidden_dim: int, num_freq_bands: int) -> None:
        super().__init__()
        if hidden_dim != ops.C or num_freq_bands != 64 or not 0 <= input_dim <= 4:
            raise NotImplementedError('the gfx950 kernel is built for hidden_dim=128, num_freq_bands=64, input_dim<=4 '
                                      '(got %d, %d, %d)' % (hidden_dim, num_freq_bands, input_dim))
        self.input_dim = input_dim
        self.hidden_dim = hidden_dim
        self.num_freq_bands = num_freq_bands

        self.freqs = nn.Embedding(input_dim, num_freq_bands) if input_dim != 0 else None
        self.mlps = nn.ModuleList([
            nn.Sequential(
                nn.Linear(num_freq_bands * 2 + 1, hidden_dim),
                nn.LayerNorm(hidden_dim),
                nn.ReLU(inplace=True),
                nn.Linear(hidden_dim, hidden_dim),
            ) for _ in range(input_dim)])
        self.to_out = nn.Sequential(
            nn.LayerNorm(hidden_dim),
            nn.ReLU(inplace=True),
            nn.Linear(hidden_dim, hidden_dim),
        )
        self.apply(weight_init)
        self._pack_key = None
        self._pack = None

    def packed_weights(self) -> torch.Tensor:
        """Parameters in the layout of include/qcnet_hip.h (qcn_fourier_embed_f32), cached per parameter version."""
        params = list(self.parameters())
        mode = ops.matmul_mode()
        key = (mode,) + tuple((p.data_ptr(), p._version) for p in params)
        if key != self._pack_key:
            with torch.no_grad():
                mlps = [(m[0].weight, m[0].bias, m[1].weight, m[1].bias, m[3].weight, m[3].bias) for m in self.mlps]
                ln, _, lin = self.to_out
                self._pack = ops.pack_fourier(self.freqs.weight if self.freqs is not None else None, mlps,
                                              (ln.weight, ln.bias, lin.weight, lin.bias), mode=mode)
            self._pack_key = key
        return self._pack

    def forward(self,
                continuous_inputs: Optional[torch.Tensor] = None,
                categorical_embs: Optional[List[torch.Tensor]] = None,
                categorical_repeat: int = 1,
                normalized_output: bool = False) -> torch.Tensor:
        """Reference signature plus two keyword extras used by the modules' fast paths:
        ``categorical_repeat`` = R means row n takes ``categorical_embs[...][n // R]`` (the reference materialises
        ``repeat_interleave(R)``, modules/qcnet_agent_encoder.py:109); ``normalized_output`` returns the result
        row-normalised, i.e. after the layer-independent half of the consumers' ``attn_prenorm_r`` (True: fp32 rows;
        'split' / 'bf16': the operand formats of the mode attention kernel, see ops.fourier_embed)."""
        if continuous_inputs is None and categorical_embs is None:
            raise ValueError('Both continuous_inputs and categorical_embs are None')
        if (continuous_inputs is None) != (self.input_dim == 0):
            raise ValueError('continuous_inputs does not match input_dim=%d' % self.input_dim)
        extra = None
        if categorical_embs is not None:
            extra = categorical_embs[0] if len(categorical_embs) == 1 else torch.stack(categorical_embs).sum(dim=0)
            extra = extra.contiguous()
        x = None
        if continuous_inputs is not None:
            x = continuous_inputs.reshape(-1, self.input_dim).contiguous().float()
        mode = ops.matmul_mode()
        pack = self.packed_weights()
        return ops.fourier_embed(x, self.input_dim, pack, extra, categorical_repeat, normalized_output, mode=mode)
