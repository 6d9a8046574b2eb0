"""Parameter initialisation with the reference's rules (utils/weight_init.py:17-82), restricted to the module
types that occur on the QCNet path: Linear, Embedding, LayerNorm, GRU / GRUCell."""
import torch.nn as nn


def _init_linear(m: nn.Linear) -> None:
    nn.init.xavier_uniform_(m.weight)           # :19
    if m.bias is not None:
        nn.init.zeros_(m.bias)                   # :20-21


def _init_embedding(m: nn.Embedding) -> None:
    nn.init.normal_(m.weight, mean=0.0, std=0.02)   # :30


def _init_layer_norm(m: nn.LayerNorm) -> None:
    nn.init.ones_(m.weight)                      # :35-36
    nn.init.zeros_(m.bias)


def _init_gru(m: nn.Module) -> None:
    for name, param in m.named_parameters():     # :72-82, one block per gate
        if 'weight_ih' in name:
            for gate in param.chunk(3, 0):
                nn.init.xavier_uniform_(gate)
        elif 'weight_hh' in name:
            for gate in param.chunk(3, 0):
                nn.init.orthogonal_(gate)
        elif 'bias' in name:
            nn.init.zeros_(param)


_RULES = ((nn.Linear, _init_linear), (nn.Embedding, _init_embedding), (nn.LayerNorm, _init_layer_norm),
          ((nn.GRU, nn.GRUCell), _init_gru))


def weight_init(m: nn.Module) -> None:
    for types, rule in _RULES:
        if isinstance(m, types):
            rule(m)
            return
