"""Scenario sharding across the GPUs of one node.

Scenarios are independent units (no relation crosses a scenario: every graph is built per example), so the path
shards with no collective inside the forward.  Each rank takes scenario ids ``rank, rank + W, ...``, runs them in
batches, and the only exchanges are one ``all_gather`` of fixed-size per-scenario prediction records and one
``all_reduce(sum)`` of metric ``(sum, count)`` pairs — the counterparts of Lightning's ``sync_dist`` logging and the
torchmetrics ``dist_reduce_fx='sum'`` states of the reference (predictors/temp_ens_learn_agg.py:243-247,
metrics/min_ade.py:29-30).  Backend: ``nccl`` (= RCCL over xGMI) on GPUs, ``gloo`` in the CPU tests.
"""
from typing import Callable, Dict, List, Sequence, Tuple

import torch
import torch.distributed as dist

RECORD_MODES = 6
RECORD_STEPS = 50
# scenario id, trajectories, probabilities of the AV2 configuration; records of another configuration have the width their
# predictions give (focal_records) and gather_records agrees on it across ranks
RECORD_WIDTH = 1 + RECORD_MODES * RECORD_STEPS * 2 + RECORD_MODES
MAX_SCENARIO_ID = 1 << 24    # ids travel as an fp32 column of the record: exact below 2^24


def shard_scenario_ids(num_scenarios: int, rank: int, world_size: int) -> List[int]:
    return list(range(rank, num_scenarios, world_size))


def batches(ids: Sequence[int], batch_size: int) -> List[List[int]]:
    return [list(ids[i:i + batch_size]) for i in range(0, len(ids), batch_size)]


def focal_index(scenario_ids: Sequence[int], data) -> Tuple[torch.Tensor, torch.Tensor]:
    """(rows of the focal tracks, scenario ids as an fp32 column) of a batch.  Both need a host round trip (``nonzero``,
    a host-to-device copy); they depend on the inputs alone, so they are made once per batch container and, on a GPU, on
    the preparation stream: the host then never waits for the prediction kernels of the main stream and can enqueue the
    next batch while this one runs."""
    from .utils.prep import run_prep

    if len(scenario_ids) and not 0 <= min(scenario_ids) <= max(scenario_ids) < MAX_SCENARIO_ID:
        raise ValueError('scenario ids must be dense integers in [0, 2^24): they are carried as an fp32 record column '
                         '(map dataset ids such as AV2 UUID strings to indices first)')

    def make():
        category = data['agent']['category']
        focal = (category == 3).nonzero(as_tuple=True)[0]
        ids = torch.tensor(list(scenario_ids), dtype=torch.float32).to(category.device).unsqueeze(1)
        return focal, ids

    def cached():
        return run_prep(data['agent']['category'].device, make)

    key = ('focal', tuple(scenario_ids))
    focal, ids = data.derived(key, cached) if hasattr(data, 'derived') else make()
    assert focal.numel() == len(scenario_ids), 'expected exactly one focal track per scenario'
    return focal, ids


def focal_records(scenario_ids: Sequence[int], data, pred: Dict[str, torch.Tensor]) -> torch.Tensor:
    """One fp32 record per scenario: [scenario id, traj_refine[focal, :, :50, :2] (agent frame), softmax(pi)]."""
    focal, ids = focal_index(scenario_ids, data)
    traj = pred['traj_refine'][focal][..., :2].reshape(len(scenario_ids), -1)
    prob = pred['pi_softmax'][focal]
    return torch.cat([ids, traj.float(), prob.float()], dim=1)


def endpoint_metric_sums(data, pred: Dict[str, torch.Tensor], time_shift: int = 10) -> torch.Tensor:
    """(sum, count) pairs of minADE / minFDE / miss-rate (2 m) of the focal tracks against the synthetic future,
    in the agent frame of step 49 + time_shift (ground truth as predictors/temp_ens_learn_agg.py:104-121)."""
    T = 50
    agent = data['agent']
    focal = (agent['category'] == 3).nonzero(as_tuple=True)[0]
    origin = agent['position'][focal, T - 1 + time_shift, :2]
    theta = agent['heading'][focal, T - 1 + time_shift]
    cos, sin = theta.cos(), theta.sin()
    rot = torch.stack([torch.stack([cos, -sin], dim=-1), torch.stack([sin, cos], dim=-1)], dim=-2)
    gt = torch.bmm(agent['position'][focal, T + time_shift:, :2] - origin.unsqueeze(1), rot)      # [F, 50, 2]
    valid = agent['predict_mask'][focal, T + time_shift:].float()
    traj = pred['traj_refine'][focal][..., :2]                                                     # [F, K, 50, 2]
    dist_ = torch.norm(traj - gt.unsqueeze(1), p=2, dim=-1) * valid.unsqueeze(1)
    ade = (dist_.sum(-1) / valid.sum(-1, keepdim=True).clamp(min=1)).min(dim=1).values
    fde = dist_[..., -1].min(dim=1).values
    n = torch.tensor(float(focal.numel()), device=traj.device)
    return torch.stack([ade.sum(), n, fde.sum(), n, (fde > 2.0).float().sum(), n])


VALIDATION_METRICS = ('Brier', 'minADE', 'minADE_best', 'minAHE', 'minFDE', 'minFHE', 'MR')
_VALIDATION_COLUMNS = ('brier_fde', 'ade_fde', 'ade_ade', 'ahe_fde', 'fde', 'fhe', 'miss_fde')


def validation_metric_sums(data, pred: Dict[str, torch.Tensor], num_historical_steps: int = 50, time_shift: int = 10,
                           output_dim: int = 2) -> torch.Tensor:
    """(sum, count) pairs of the reference's seven validation metrics (VALIDATION_METRICS order,
    predictors/temp_ens_learn_agg.py:264-275) for one batch: ground truth and all metric addends from the HIP kernels
    (qcn_to_agent_frame_f32 + qcn_motion_metrics_f32), no host round trip.  GPU only."""
    from . import ops
    T = num_historical_steps
    agent = data['agent']
    target = ops.to_agent_frame(agent['position'].contiguous(), agent['heading'].contiguous(), T - 1 + time_shift, T + time_shift)
    rec = ops.motion_metrics(pred['traj_refine'], target, prob=pred['pi_softmax'], valid_mask=agent['predict_mask'][:, T + time_shift:],
                             select=agent['category'] == 3, pos_dim=output_dim, pred_head=ops.HEAD_FROM_MOTION, target_head=-1)
    tot = rec.sum(dim=0)
    # (no index tensor built from a Python list here: that would be a host-to-device copy, i.e. a host stall, per batch)
    sums = torch.stack([tot[ops.METRIC_COLUMNS[c]] for c in _VALIDATION_COLUMNS])
    return torch.stack([sums, tot[ops.METRIC_COLUMNS['counted']].expand_as(sums)], dim=1).reshape(-1)


def gather_records(records: torch.Tensor, world_size: int) -> torch.Tensor:
    """all_gather of per-scenario records from every rank (ragged counts padded to the max), sorted by id.  A rank whose
    shard is empty passes a [0, anything] tensor: the record width is agreed on across ranks (the maximum; ranks with
    records must agree with each other)."""
    if world_size == 1 or not dist.is_initialized():
        out = records
    else:
        shape = torch.tensor([records.size(0), records.size(1) if records.size(0) else 0], dtype=torch.int64,
                             device=records.device)
        shapes = [torch.zeros_like(shape) for _ in range(world_size)]
        dist.all_gather(shapes, shape)
        shapes = [[int(v) for v in t.tolist()] for t in shapes]
        counts = [c for c, _ in shapes]
        width = max(w for _, w in shapes)
        if any(c and w != width for c, w in shapes):
            raise ValueError('ranks disagree on the record width: %s' % (shapes,))
        cap = max(counts)
        padded = records.new_full((cap, width), -1.0)
        if records.size(0):
            padded[:records.size(0)] = records
        gathered = records.new_empty((world_size * cap, width))
        if cap * width:
            dist.all_gather_into_tensor(gathered, padded)
        keep = torch.cat([torch.arange(cap, device=records.device) < c for c in counts])
        out = gathered[keep]
    return out[torch.argsort(out[:, 0])] if out.size(0) else out


def reduce_metric_sums(sums: torch.Tensor, world_size: int) -> torch.Tensor:
    if world_size > 1 and dist.is_initialized():
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    return sums


def start_loader(get_scenario: Callable[[int], object], num_scenarios: int, batch_size: int, rank: int, world_size: int,
                 num_workers: int):
    """The worker-process loader over this rank's shard (datasets/loader.py); its workers start building batches at once."""
    from .datasets.loader import scenario_loader
    return scenario_loader(shard_scenario_ids(num_scenarios, rank, world_size), get_scenario, batch_size, num_workers=num_workers)


_COPY_STREAMS: Dict[tuple, 'torch.cuda.Stream'] = {}


def to_device_ahead(cpu_batch, device):
    """The device copy of a (pinned) batch, made on a COPY stream of its own: the copies do not queue behind the kernels
    of the previous step, so the batch's ready mark (``data.mark_ready``, recorded on the copy stream) fires while that
    step still runs and the graph preparation of this batch (utils/prep.py) overlaps it — with the copies on the main
    stream the preparation could only start when the previous step had finished (cfg5: 685 -> scenarios/s, see DESIGN).
    The current stream waits for the mark; the tensors are marked as used on it and on the path's other streams
    (``prep.consumer_streams``), because their memory belongs to the copy stream's pool."""
    device = torch.device(device)
    if device.type != 'cuda':
        return cpu_batch.to(device)
    import threading
    from .utils import prep
    key = (threading.get_ident(), device.index if device.index is not None else torch.cuda.current_device())
    copy = _COPY_STREAMS.get(key)
    if copy is None:
        copy = _COPY_STREAMS[key] = torch.cuda.Stream(device=device)
    main = torch.cuda.current_stream(device)
    with torch.cuda.stream(copy):
        data = cpu_batch.to(device, non_blocking=True)          # records the ready mark on the copy stream
    for stream in [main] + prep.consumer_streams(device):
        prep.record_streams(data, stream)
    event = data.ready_event() if hasattr(data, 'ready_event') else None
    if event is not None:
        main.wait_event(event)
    else:
        main.wait_stream(copy)
    return data


endpoint_metric_sums.width = 6
validation_metric_sums.width = 2 * len(VALIDATION_METRICS)


def run_sharded(predict_fn: Callable, make_batch_fn: Callable[[List[int]], object], num_scenarios: int,
                batch_size: int, rank: int, world_size: int, device,
                metric_fn: Callable = endpoint_metric_sums, metric_width: int = None,
                get_scenario: Callable[[int], object] = None, num_workers: int = 0, loader=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Runs this rank's shard and returns (all records sorted by scenario id, reduced metric sums).
    Batches: ``make_batch_fn(ids)`` builds one synchronously inside the loop; with ``get_scenario`` (scenario id -> one
    scenario) instead, ``num_workers`` worker processes build and collate the batches AHEAD of the GPU and hand them over in
    pinned memory (datasets/loader.py — the reference's ``DataLoader(num_workers, pin_memory)``, evaluate.py:42-43), and the
    host-to-device copies are asynchronous: the loop never waits for a batch unless the workers are the slower side.
    ``loader``: an already started ``start_loader(...)`` iterator for this rank's shard (workers that outlive one run, the
    reference's ``persistent_workers=True``) instead of starting one here.
    ``metric_fn(data, pred)``
    returns the batch's (sum, count) pairs: ``validation_metric_sums`` on the GPU (the reference's seven metrics from the
    HIP kernel); the default is device-agnostic torch arithmetic so the collectives can be exercised over gloo.
    ``metric_width``: length of that vector (default: ``metric_fn.width``) — a rank whose shard is empty
    (num_scenarios < world_size) contributes zeros of the SAME length to the closing all_reduce."""
    if metric_width is None:
        metric_width = getattr(metric_fn, 'width', None)
    records = []
    sums = None
    my_ids = shard_scenario_ids(num_scenarios, rank, world_size)
    if loader is not None:
        source = loader
    elif get_scenario is not None:
        source = start_loader(get_scenario, num_scenarios, batch_size, rank, world_size, num_workers)
    else:
        source = ((ids, make_batch_fn(ids)) for ids in batches(my_ids, batch_size))
    for ids, cpu_batch in source:
        ids = [int(i) for i in ids]
        data = to_device_ahead(cpu_batch, device)
        pred = predict_fn(data)
        records.append(focal_records(ids, data, pred))
        part = metric_fn(data, pred).float()
        if metric_width is not None and part.numel() != metric_width:
            raise ValueError('metric_fn returned %d values, metric_width says %d' % (part.numel(), metric_width))
        sums = part if sums is None else sums + part
    if sums is None:
        if metric_width is None:
            raise ValueError('this rank has no scenarios and the width of metric_fn\'s result is unknown: pass metric_width '
                             '(every rank must all_reduce a vector of the same length)')
        sums = torch.zeros(metric_width, device=device)
    records = torch.cat(records) if records else torch.empty(0, 0, device=device)
    return gather_records(records, world_size), reduce_metric_sums(sums, world_size)
