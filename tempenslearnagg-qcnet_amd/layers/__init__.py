from .attention_layer import AttentionLayer
from .fourier_embedding import FourierEmbedding
from .mlp_layer import MLPLayer

__all__ = ['AttentionLayer', 'FourierEmbedding', 'MLPLayer']
