from .qcnet_agent_encoder import QCNetAgentEncoder
from .qcnet_decoder import QCNetDecoder
from .qcnet_encoder import QCNetEncoder
from .qcnet_map_encoder import QCNetMapEncoder
from .temporal_aggregate import TemporalAggregate

__all__ = ['QCNetAgentEncoder', 'QCNetDecoder', 'QCNetEncoder', 'QCNetMapEncoder', 'TemporalAggregate']
