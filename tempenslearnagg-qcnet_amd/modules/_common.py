"""Helpers shared by the QCNet modules."""
from typing import Optional

import torch

from .. import ops
from ..data import Batch
from ..utils.geometry import angle_between_2d_vectors, wrap_angle


def node_ptr(data, node_type: str, device) -> torch.Tensor:
    """Per-scenario node boundaries [B + 1] (int64): ``ptr`` of a Batch, [0, N] for a single scenario.
    Replaces the ``batch`` vectors + bucketize of the reference's radius calls
    (modules/qcnet_agent_encoder.py:148-157) without a host round trip."""
    store = data[node_type]
    if isinstance(data, Batch) and 'ptr' in store:
        return store['ptr'].to(device=device, dtype=torch.int64)
    return torch.tensor([0, int(store['num_nodes'])], dtype=torch.int64, device=device)


def rel_geometry(pos_src: torch.Tensor, pos_dst: torch.Tensor, head_src: torch.Tensor, head_dst: torch.Tensor,
                 edge_index: torch.Tensor, extra: Optional[torch.Tensor] = None) -> torch.Tensor:
    """r = [|dp|, angle(heading_dst, dp), wrap(dheading) (, extra)] with dp = pos_src[src] - pos_dst[dst]:
    the relative-geometry vector every relation feeds its FourierEmbedding
    (modules/qcnet_map_encoder.py:121-128, modules/qcnet_agent_encoder.py:133-139,161-166,171-176,
    modules/qcnet_decoder.py:161-167,183-188,201-206)."""
    return ops.rel_geometry(pos_src.contiguous(), pos_dst.contiguous(), head_src.contiguous(), head_dst.contiguous(),
                            edge_index, extra)


def rel_geometry_torch(pos_src: torch.Tensor, pos_dst: torch.Tensor, head_src: torch.Tensor, head_dst: torch.Tensor,
                       edge_index: torch.Tensor, extra: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The same vector from torch ops with the reference's helpers (the form the kernel is tested against)."""
    src, dst = edge_index[0], edge_index[1]
    rel_pos = pos_src[src] - pos_dst[dst]
    hd = head_dst[dst]
    ctr = torch.stack([hd.cos(), hd.sin()], dim=-1)
    cols = [torch.norm(rel_pos[:, :2], p=2, dim=-1),
            angle_between_2d_vectors(ctr, rel_pos[:, :2]),
            wrap_angle(head_src[src] - hd)]
    if extra is not None:
        cols.append(extra.to(rel_pos.dtype))
    return torch.stack(cols, dim=-1)
