"""FourierEmbedding — drop-in for layers/fourier_embedding.py:23-67 (same ctor, forward and state_dict keys)."""
import math
from typing import List, Optional

import torch
import torch.nn as nn

from ..utils.weight_init import weight_init


class FourierEmbedding(nn.Module):

    def __init__(self, input_dim: int, hidden_dim: int, num_freq_bands: int) -> None:
        super().__init__()
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

    def forward(self,
                continuous_inputs: Optional[torch.Tensor] = None,
                categorical_embs: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
        if continuous_inputs is None:
            if categorical_embs is None:
                raise ValueError('Both continuous_inputs and categorical_embs are None')
            x = torch.stack(categorical_embs).sum(dim=0)
        else:
            # ((x * w) * 2) * pi, the reference's evaluation order (:58)
            ang = continuous_inputs.unsqueeze(-1) * self.freqs.weight * 2 * math.pi
            feat = torch.cat([ang.cos(), ang.sin(), continuous_inputs.unsqueeze(-1)], dim=-1)
            x = self.mlps[0](feat[:, 0])
            for i in range(1, self.input_dim):
                x = x + self.mlps[i](feat[:, i])
            if categorical_embs is not None:
                x = x + torch.stack(categorical_embs).sum(dim=0)
        return self.to_out(x)
