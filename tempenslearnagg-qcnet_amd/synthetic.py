"""Synthetic AV2-shaped scenarios (SURVEY.md §8(d) generators).

The schema is the one ``datasets/argoverse_v2_dataset.py:273-284`` (agent store) and
``:479-512`` (map stores) write per scenario; there is no network for the real Argoverse 2
data, so benchmarks and parity tests draw scenes from here.  Seed = 2024 + scenario id,
``torch.Generator`` on CPU, fp32.
"""
import math
from typing import Dict, List, Sequence

import torch

from .data import Batch, HeteroData

NUM_STEPS = 110          # datasets/argoverse_v2_dataset.py:80-82 (50 history + 60 future)
NUM_HISTORICAL_STEPS = 50
MAX_TIME_SHIFT = 10      # predictors/temp_ens_learn_agg.py:197 loops time_shift = 1..10

# Model hyper-parameters: argparse defaults of predictors/qcnet_sliding.py:417-448 plus the
# upstream-QCNet AV2 values for the flags that have no default.
HPARAMS_AV2: Dict[str, object] = dict(
    dataset='argoverse_v2', input_dim=2, hidden_dim=128, output_dim=2, output_head=False,
    num_historical_steps=50, num_future_steps=60, num_modes=6, num_recurrent_steps=3,
    num_freq_bands=64, num_map_layers=1, num_agent_layers=2, num_dec_layers=2, num_heads=8,
    head_dim=16, dropout=0.1, pl2pl_radius=150.0, time_span=10, pl2a_radius=50.0, a2a_radius=50.0,
    num_t2m_steps=30, pl2m_radius=150.0, a2m_radius=150.0)

# BASELINE.json configs[3]: dense-scene stress.
HPARAMS_DENSE: Dict[str, object] = dict(HPARAMS_AV2, pl2a_radius=100.0, a2a_radius=100.0)

SCENE_AV2 = dict(num_agents=50, num_polygons=150, points_per_polygon=30, map_extent=100.0, agent_extent=60.0)
SCENE_DENSE = dict(num_agents=200, num_polygons=1000, points_per_polygon=30, map_extent=150.0,
                   agent_extent=100.0)
# A scene small enough for the CPU oracle to run the full TempEnsLearnAgg loop in seconds.
SCENE_TINY = dict(num_agents=6, num_polygons=12, points_per_polygon=6, map_extent=40.0, agent_extent=25.0)


def _uniform(gen: torch.Generator, shape, lo: float, hi: float) -> torch.Tensor:
    return torch.rand(shape, generator=gen, dtype=torch.float32) * (hi - lo) + lo


def make_scenario(scenario_id: int,
                  num_agents: int = 50,
                  num_polygons: int = 150,
                  points_per_polygon: int = 30,
                  map_extent: float = 100.0,
                  agent_extent: float = 60.0) -> HeteroData:
    gen = torch.Generator(device='cpu')
    gen.manual_seed(2024 + int(scenario_id))
    A, P = num_agents, num_polygons
    assert points_per_polygon % 3 == 0
    n_side = points_per_polygon // 3

    # ---------------------------------------------------------------- map
    pl_xy = _uniform(gen, (P, 2), -map_extent, map_extent)
    pl_orient = _uniform(gen, (P,), -math.pi, math.pi)
    pl_pos = torch.cat([pl_xy, torch.zeros(P, 1)], dim=-1)
    pl_type = torch.randint(0, 4, (P,), generator=gen).to(torch.uint8)
    pl_int = torch.randint(0, 3, (P,), generator=gen).to(torch.uint8)

    along = torch.stack([pl_orient.cos(), pl_orient.sin()], dim=-1)            # [P, 2]
    lateral = torch.stack([-pl_orient.sin(), pl_orient.cos()], dim=-1)
    steps = torch.arange(n_side, dtype=torch.float32) * 2.0                    # 2 m spacing
    offsets = torch.tensor([1.8, -1.8, 0.0])                                   # left, right, centre
    pt_xy = (pl_xy[:, None, None, :] + along[:, None, None, :] * steps[None, None, :, None] +
             lateral[:, None, None, :] * offsets[None, :, None, None])         # [P, 3, n_side, 2]
    pt_xy = pt_xy.reshape(P * points_per_polygon, 2)
    Npt = pt_xy.size(0)
    pt_pos = torch.cat([pt_xy, torch.zeros(Npt, 1)], dim=-1)
    pt_orient = pl_orient.repeat_interleave(points_per_polygon) + torch.randn(Npt, generator=gen) * 0.02
    pt_orient = (pt_orient + math.pi) % (2 * math.pi) - math.pi
    pt_mag = 2.0 + torch.randn(Npt, generator=gen).abs() * 0.05
    pt_type = torch.randint(0, 17, (P,), generator=gen).repeat_interleave(points_per_polygon).to(torch.uint8)
    pt_side = torch.tensor([0, 1, 2], dtype=torch.uint8).repeat_interleave(n_side).repeat(P)
    pt2pl = torch.stack([torch.arange(Npt, dtype=torch.long),
                         torch.arange(P, dtype=torch.long).repeat_interleave(points_per_polygon)], dim=0)
    # successor chain i -> i+1 (type 2 = SUCC) plus the reverse (type 1 = PRED)
    fwd = torch.stack([torch.arange(0, P - 1), torch.arange(1, P)], dim=0)
    bwd = fwd.flip(0)
    pl2pl = torch.cat([fwd, bwd], dim=1).long()
    pl2pl_type = torch.cat([torch.full((P - 1,), 2), torch.full((P - 1,), 1)]).to(torch.uint8)

    # ---------------------------------------------------------------- agents
    start = _uniform(gen, (A, 2), -agent_extent, agent_extent)
    heading0 = _uniform(gen, (A,), -math.pi, math.pi)
    speed = _uniform(gen, (A,), 0.0, 10.0)
    yaw_rate = torch.randn(A, generator=gen) * 0.02                            # rad / step, gentle turns
    t = torch.arange(NUM_STEPS, dtype=torch.float32)
    heading = heading0[:, None] + yaw_rate[:, None] * t[None, :] * 0.1
    vel_dir = torch.stack([heading.cos(), heading.sin()], dim=-1)              # [A, 110, 2]
    disp = vel_dir * (speed[:, None, None] * 0.1)
    xy = start[:, None, :] + torch.cumsum(disp, dim=1) - disp
    xy = xy + torch.randn(A, NUM_STEPS, 2, generator=gen) * 0.05
    vel = torch.zeros(A, NUM_STEPS, 2)
    vel[:, 1:] = (xy[:, 1:] - xy[:, :-1]) / 0.1
    vel[:, 0] = vel[:, 1]
    heading = (heading + math.pi) % (2 * math.pi) - math.pi

    observed = torch.ones(A, NUM_STEPS, dtype=torch.bool)
    kind = torch.rand(A, generator=gen)
    for a in range(1, A):                                                      # agent 0 (focal) is always fully observed
        if kind[a] < 0.7:
            continue
        if kind[a] < 0.9:   # contiguous sub-window covering the current steps 45..60
            lo = int(torch.randint(0, 45, (1,), generator=gen))
            hi = int(torch.randint(61, NUM_STEPS + 1, (1,), generator=gen))
        elif kind[a] < 0.95:  # track lost before the first "current" step
            lo = int(torch.randint(0, 20, (1,), generator=gen))
            hi = int(torch.randint(25, 49, (1,), generator=gen))
        else:               # track appears late
            lo = int(torch.randint(52, 70, (1,), generator=gen))
            hi = NUM_STEPS
        observed[a] = False
        observed[a, lo:hi] = True

    valid = observed.clone()
    current_valid = valid[:, NUM_HISTORICAL_STEPS - 1].clone()
    # vector representation: step t valid only if t-1 is valid (datasets/argoverse_v2_dataset.py:248-252)
    valid[:, 1:NUM_HISTORICAL_STEPS] = valid[:, :NUM_HISTORICAL_STEPS - 1] & valid[:, 1:NUM_HISTORICAL_STEPS]
    valid[:, 0] = False
    predict = observed.clone()
    predict[:, :NUM_HISTORICAL_STEPS] = False
    predict[~current_valid, NUM_HISTORICAL_STEPS:] = False

    position = torch.zeros(A, NUM_STEPS, 3)
    position[..., :2] = xy
    velocity = torch.zeros(A, NUM_STEPS, 3)
    velocity[..., :2] = vel
    position[~observed] = 0.0
    velocity[~observed] = 0.0
    heading = heading.masked_fill(~observed, 0.0)

    a_type = torch.randint(0, 10, (A,), generator=gen).to(torch.uint8)
    category = torch.randint(0, 3, (A,), generator=gen).to(torch.uint8)
    category[0] = 3                                                            # exactly one focal track

    return HeteroData({
        'scenario_id': 'synthetic-%06d' % int(scenario_id),
        'city': 'synthetic',
        'agent': {
            'num_nodes': A, 'av_index': 0, 'valid_mask': valid, 'predict_mask': predict,
            'id': ['a%d' % i for i in range(A)], 'type': a_type, 'category': category,
            'position': position, 'heading': heading, 'velocity': velocity,
        },
        'map_polygon': {
            'num_nodes': P, 'position': pl_pos, 'orientation': pl_orient, 'height': torch.zeros(P),
            'type': pl_type, 'is_intersection': pl_int,
        },
        'map_point': {
            'num_nodes': Npt, 'position': pt_pos, 'orientation': pt_orient, 'magnitude': pt_mag,
            'height': torch.zeros(Npt), 'type': pt_type, 'side': pt_side,
        },
        ('map_point', 'to', 'map_polygon'): {'edge_index': pt2pl},
        ('map_polygon', 'to', 'map_polygon'): {'edge_index': pl2pl, 'type': pl2pl_type},
    })


def make_batch(scenario_ids: Sequence[int], **scene) -> Batch:
    return Batch.from_data_list([make_scenario(i, **scene) for i in scenario_ids])


def config_scene(name: str) -> Dict[str, object]:
    """Scene + hyper-parameter preset for a BASELINE.json config name."""
    if name in ('cfg1', 'cfg2', 'cfg3', 'cfg5'):
        return dict(scene=SCENE_AV2, hparams=HPARAMS_AV2, batch={'cfg1': 1}.get(name, 32))
    if name == 'cfg4':
        return dict(scene=SCENE_DENSE, hparams=HPARAMS_DENSE, batch=4)
    if name == 'tiny':
        return dict(scene=SCENE_TINY, hparams=dict(HPARAMS_AV2, pl2pl_radius=30.0, pl2a_radius=20.0,
                                                   a2a_radius=20.0, pl2m_radius=30.0, a2m_radius=30.0), batch=2)
    raise ValueError('unknown config %r' % name)
