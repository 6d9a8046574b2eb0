from .geometry import angle_between_2d_vectors, wrap_angle
from .graph import (CSRGraph, bipartite_dense_to_sparse, coalesce, dense_to_sparse, merge_edges, radius,
                    radius_graph, subgraph)
from .weight_init import weight_init

__all__ = ['angle_between_2d_vectors', 'wrap_angle', 'CSRGraph', 'bipartite_dense_to_sparse', 'coalesce',
           'dense_to_sparse', 'merge_edges', 'radius', 'radius_graph', 'subgraph', 'weight_init']
