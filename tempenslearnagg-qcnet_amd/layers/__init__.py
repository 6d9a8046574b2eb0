from .attention_layer import AttentionLayer, NormalizedEdgeFeatures
from .fourier_embedding import FourierEmbedding
from .mlp_layer import MLPLayer

__all__ = ['AttentionLayer', 'NormalizedEdgeFeatures', 'FourierEmbedding', 'MLPLayer']
