"""CPU, world_size 2 over gloo: scenario sharding, the prediction all_gather and the metric all_reduce.
The forward is replaced by a cheap deterministic stand-in (the HIP path needs a GPU); what is under test is the
partitioning and the two collectives."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import ROOT

from tempenslearnagg_qcnet_amd import sharding
from tempenslearnagg_qcnet_amd.synthetic import SCENE_TINY, make_batch


def _fake_predict(data):
    focal = data['agent']['position'][:, 59, :2]
    A = focal.size(0)
    traj = focal.view(A, 1, 1, 2).expand(A, 6, 50, 2) + torch.arange(6).view(1, 6, 1, 1) * 0.5
    pi = focal[:, :1].expand(A, 6) * torch.arange(6)
    return {'traj_refine': torch.cat([traj, torch.ones_like(traj)], dim=-1), 'pi_softmax': torch.softmax(pi, -1)}


def _wide_metric(data, pred):
    """A 14-value (sum, count) vector like sharding.validation_metric_sums (which needs the GPU)."""
    base = sharding.endpoint_metric_sums(data, pred)
    return torch.cat([base, base, base[:2]])


_wide_metric.width = 14


def _worker(rank, world, port, num_scenarios, out_dir, wide=False):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    kw = {'metric_fn': _wide_metric} if wide else {}
    recs, sums = sharding.run_sharded(_fake_predict, lambda ids: make_batch(ids, **SCENE_TINY), num_scenarios, 3,
                                      rank, world, torch.device('cpu'), **kw)
    torch.save({'records': recs, 'sums': sums}, os.path.join(out_dir, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def test_sharded_run_equals_single_process(tmp_path):
    num_scenarios, world = 7, 2                                           # ragged: 4 + 3 scenarios
    mp.spawn(_worker, args=(world, _free_port(), num_scenarios, str(tmp_path)), nprocs=world, join=True)
    single_recs, single_sums = sharding.run_sharded(_fake_predict, lambda ids: make_batch(ids, **SCENE_TINY),
                                                    num_scenarios, 3, 0, 1, torch.device('cpu'))
    for r in range(world):
        got = torch.load(os.path.join(str(tmp_path), 'rank%d.pt' % r))
        assert got['records'].shape == (num_scenarios, sharding.RECORD_WIDTH)
        assert got['records'][:, 0].tolist() == list(range(num_scenarios))
        assert torch.equal(got['records'], single_recs)                   # bit-identical to the 1-process run
        assert torch.allclose(got['sums'], single_sums, rtol=1e-6)
        assert got['sums'][1] == num_scenarios


def test_empty_shard_with_a_wide_metric_vector(tmp_path):
    """num_scenarios < world_size: rank 1 has nothing to run and must still join both collectives with tensors of the
    other rank's widths (14 metric values, the record width the predictions give)."""
    num_scenarios, world = 1, 2
    mp.spawn(_worker, args=(world, _free_port(), num_scenarios, str(tmp_path), True), nprocs=world, join=True)
    single_recs, single_sums = sharding.run_sharded(_fake_predict, lambda ids: make_batch(ids, **SCENE_TINY),
                                                    num_scenarios, 3, 0, 1, torch.device('cpu'), metric_fn=_wide_metric)
    assert single_sums.numel() == 14
    for r in range(world):
        got = torch.load(os.path.join(str(tmp_path), 'rank%d.pt' % r))
        assert torch.equal(got['records'], single_recs)
        assert torch.allclose(got['sums'], single_sums, rtol=1e-6)


def test_empty_shard_needs_a_known_metric_width():
    import pytest
    with pytest.raises(ValueError, match='metric_width'):
        sharding.run_sharded(_fake_predict, lambda ids: make_batch(ids, **SCENE_TINY), 0, 3, 0, 1, torch.device('cpu'),
                             metric_fn=lambda d, p: torch.zeros(3))
    recs, sums = sharding.run_sharded(_fake_predict, lambda ids: make_batch(ids, **SCENE_TINY), 0, 3, 0, 1,
                                      torch.device('cpu'))
    assert recs.size(0) == 0 and sums.tolist() == [0.0] * 6


def test_scenario_ids_must_fit_the_fp32_column():
    import pytest
    data = make_batch([0], **SCENE_TINY)
    with pytest.raises(ValueError, match='2\\^24'):
        sharding.focal_index([1 << 24], data)


def test_partition_is_a_disjoint_cover():
    for world in (1, 2, 4, 8):
        ids = [i for r in range(world) for i in sharding.shard_scenario_ids(10000, r, world)]
        assert sorted(ids) == list(range(10000))
    assert sharding.batches(list(range(7)), 3) == [[0, 1, 2], [3, 4, 5], [6]]


def test_worker_loader_equals_synchronous_batches():
    """datasets/loader.py (worker processes collate ahead, the reference's DataLoader(num_workers), evaluate.py:42-43): the
    same (ids, batch) sequence as building every batch inside the loop — ragged last batch, an empty shard — and the same
    records through run_sharded."""
    import functools
    from tempenslearnagg_qcnet_amd.datasets import scenario_loader
    from tempenslearnagg_qcnet_amd.synthetic import make_scenario
    get = functools.partial(make_scenario, **SCENE_TINY)
    ids = [5, 0, 3, 9, 11, 2, 7]
    for workers in (0, 2):
        seen = list(scenario_loader(ids, get, 3, num_workers=workers, pin_memory=False))
        assert [[int(i) for i in b[0]] for b in seen] == [[5, 0, 3], [9, 11, 2], [7]]
        for got_ids, got in seen:
            want = make_batch([int(i) for i in got_ids], **SCENE_TINY)
            assert got.num_graphs == want.num_graphs
            for key in ('agent', 'map_polygon', 'map_point', ('map_point', 'to', 'map_polygon'), ('map_polygon', 'to', 'map_polygon')):
                for name, v in want[key].items():
                    if torch.is_tensor(v):
                        assert torch.equal(got[key][name], v), (key, name)
    assert list(scenario_loader([], get, 3, num_workers=2)) == []
    sync = sharding.run_sharded(_fake_predict, lambda i: make_batch(i, **SCENE_TINY), 8, 3, 0, 1, torch.device('cpu'))
    ahead = sharding.run_sharded(_fake_predict, None, 8, 3, 0, 1, torch.device('cpu'), get_scenario=get, num_workers=2)
    assert torch.equal(sync[0], ahead[0]) and torch.equal(sync[1], ahead[1])
