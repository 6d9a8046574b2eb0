"""CPU: scenario containers, collation and the synthetic generators (host logic)."""
import torch

from tempenslearnagg_qcnet_amd.data import Batch, HeteroData
from tempenslearnagg_qcnet_amd.synthetic import SCENE_AV2, SCENE_TINY, make_batch, make_scenario


def test_scenario_schema_matches_reference_dataset():
    d = make_scenario(0, **SCENE_AV2)
    a, pl, pt = d['agent'], d['map_polygon'], d['map_point']
    assert a['num_nodes'] == 50 and pl['num_nodes'] == 150 and pt['num_nodes'] == 4500
    assert a['valid_mask'].shape == (50, 110) and a['valid_mask'].dtype == torch.bool
    assert a['predict_mask'].shape == (50, 110) and not bool(a['predict_mask'][:, :50].any())
    assert a['position'].shape == (50, 110, 3) and a['velocity'].shape == (50, 110, 3) and a['heading'].shape == (50, 110)
    assert a['type'].dtype == torch.uint8 and int(a['type'].max()) < 10
    assert int((a['category'] == 3).sum()) == 1
    assert not bool(a['valid_mask'][:, 0].any())                      # vector representation: step 0 invalid
    assert float(a['heading'].min()) >= -3.1416 and float(a['heading'].max()) <= 3.1416
    assert pl['type'].dtype == torch.uint8 and int(pl['type'].max()) < 4 and int(pl['is_intersection'].max()) < 3
    assert int(pt['type'].max()) < 17 and int(pt['side'].max()) < 3
    ei = d['map_point', 'to', 'map_polygon']['edge_index']
    assert ei.shape == (2, 4500) and torch.equal(ei[0], torch.arange(4500)) and int(ei[1].max()) == 149
    e2 = d['map_polygon', 'to', 'map_polygon']
    assert e2['edge_index'].shape[1] == e2['type'].numel() and set(e2['type'].tolist()) <= {1, 2, 3, 4}
    # deterministic per scenario id
    assert torch.equal(make_scenario(0, **SCENE_AV2)['agent']['position'], a['position'])
    assert not torch.equal(make_scenario(1, **SCENE_AV2)['agent']['position'], a['position'])


def test_batch_collation_follows_pyg_rules():
    ds = [make_scenario(i, **SCENE_TINY) for i in range(3)]
    b = Batch.from_data_list(ds)
    A, P, Npt = 6, 12, 72
    assert isinstance(b, Batch) and isinstance(b, HeteroData) and b.num_graphs == 3
    assert b['agent']['num_nodes'] == 3 * A and b['agent']['position'].shape == (3 * A, 110, 3)
    assert b['agent']['batch'].tolist() == [0] * A + [1] * A + [2] * A
    assert b['agent']['ptr'].tolist() == [0, A, 2 * A, 3 * A]
    assert b['map_polygon']['ptr'].tolist() == [0, P, 2 * P, 3 * P]
    ei = b['map_point', 'to', 'map_polygon']['edge_index']
    assert torch.equal(ei[:, Npt:2 * Npt], ds[1]['map_point', 'to', 'map_polygon']['edge_index']
                       + torch.tensor([[Npt], [P]]))
    e2 = b['map_polygon', 'to', 'map_polygon']
    n_e = ds[0]['map_polygon', 'to', 'map_polygon']['edge_index'].size(1)
    assert torch.equal(e2['edge_index'][:, 2 * n_e:], ds[2]['map_polygon', 'to', 'map_polygon']['edge_index'] + 2 * P)
    assert e2['type'].numel() == 3 * n_e
    assert b['agent']['av_index'].tolist() == [0, 0, 0] and len(b['scenario_id']) == 3
    moved = b.to('cpu')
    assert moved.num_graphs == 3 and torch.equal(moved['agent']['heading'], b['agent']['heading'])
    assert not hasattr(ds[0], 'num_graphs')                            # the isinstance(data, Batch) switch


def test_make_batch_ids_are_disjoint_across_ranks():
    a = make_batch([0, 1], **SCENE_TINY)
    b = make_batch([2, 3], **SCENE_TINY)
    assert not torch.equal(a['agent']['position'], b['agent']['position'])


def test_pack_chain_matrix_is_a_permutation_that_keeps_the_linear_map():
    """ops.pack_chain_matrix: P[s, h, m, t] = W[32 t + m, kin(s, h)] — contracting the packed weights with a row vector
    held in the kernels' register order gives W x (include/qcnet_hip.h, 'Chain-packed 128x128 matrix')."""
    from tempenslearnagg_qcnet_amd import ops
    torch.manual_seed(3)
    w = torch.randn(128, 128)
    x = torch.randn(128)
    order = ops.chain_feature_order()
    assert sorted(order.reshape(-1).tolist()) == list(range(128))
    p, sc = ops.pack_chain_matrix(w, mode=ops.MATMUL_EXACT_F32)
    assert sc.tolist() == [1.0, 1.0]
    p = p.view(64, 2, 32, 4)
    y = torch.einsum('shmt,sh->mt', p.double(), x[order].double())          # [m, t] -> out feature 32 t + m
    ref = (w.double() @ x.double()).view(4, 32).t()
    assert torch.allclose(y, ref, atol=1e-12)
    # first layer of a Fourier MLP: cos | sin k order
    s_idx = torch.arange(64).unsqueeze(1)
    h_idx = torch.arange(2).unsqueeze(0)
    first = torch.where(s_idx < 32, 32 * h_idx + s_idx, 64 + 32 * h_idx + (s_idx - 32))
    assert sorted(first.reshape(-1).tolist()) == list(range(128))
    p1 = ops.pack_chain_matrix(w, first, mode=ops.MATMUL_EXACT_F32)[0].view(64, 2, 32, 4)
    y1 = torch.einsum('shmt,sh->mt', p1.double(), x[first].double())
    assert torch.allclose(y1, ref, atol=1e-12)
    # split mode: hi + lo planes of the scaled weight reproduce it to 2^-22, in the [k16-step, tile, plane, lane, i] order
    ps, sc = ops.pack_chain_matrix(w, mode=ops.MATMUL_SPLIT_F16)
    planes = ps.view(torch.float16).view(8, 4, 2, 2, 32, 8).double()                 # [j, t, p, h, m, i]
    wsum = (planes[:, :, 0] + planes[:, :, 1]) * float(sc[1])                        # [j, t, h, m, i]
    assert float(sc[0]) * float(sc[1]) == 1.0 and 2 ** 14 <= float(w.abs().max()) * float(sc[0]) < 2 ** 15
    xin = x[order].double().view(8, 8, 2).permute(0, 2, 1)                           # [j, h, i]
    ys = torch.einsum('jthmi,jhi->mt', wsum, xin)
    assert torch.allclose(ys, ref, rtol=0, atol=float(w.abs().max()) * float(x.abs().sum()) * 2.0 ** -21)


def test_ready_mark_and_derived_cache_follow_the_container():
    """data.mark_ready / data.derived (data.py): per-container state that copies do not inherit and that replacing a
    store drops; on the CPU there is no event to record."""
    from tempenslearnagg_qcnet_amd.data import Batch
    from tempenslearnagg_qcnet_amd import sharding
    from tempenslearnagg_qcnet_amd.synthetic import config_scene, make_batch
    cfg = config_scene('tiny')
    batch = make_batch([0, 1], **cfg['scene'])
    assert isinstance(batch, Batch)
    calls = []
    assert batch.derived('k', lambda: calls.append(1) or 7) == 7
    assert batch.derived('k', lambda: calls.append(1) or 8) == 7 and len(calls) == 1
    copy = batch.clone()
    assert '_derived' not in copy.__dict__ and getattr(copy, '_ready_event', None) is None
    batch.mark_ready()                                   # CPU tensors: nothing to record, derived values dropped
    assert getattr(batch, '_ready_event', None) is None and '_derived' not in batch.__dict__
    focal, ids = sharding.focal_index([0, 1], batch)
    assert focal.numel() == 2 and ids.shape == (2, 1) and ids[:, 0].tolist() == [0.0, 1.0]
    assert sharding.focal_index([0, 1], batch)[0] is focal          # once per container
    batch['agent'] = dict(batch['agent'])                # a replaced store is new input
    assert '_derived' not in batch.__dict__


def test_store_changes_drop_the_derived_cache_and_ready_mark():
    """`data['agent']['x'] = t` (the raw inner store), an in-place write or a delete must invalidate what was
    derived from the inputs (focal rows) and the ready mark of the preparation stream."""
    from tempenslearnagg_qcnet_amd import sharding
    from tempenslearnagg_qcnet_amd.synthetic import SCENE_TINY, make_batch
    data = make_batch([0, 1], **SCENE_TINY)
    calls = []
    fn = lambda: calls.append(1) or len(calls)
    assert data.derived('k', fn) == 1 and data.derived('k', fn) == 1
    data['agent']['category'] = data['agent']['category'].clone()          # replaced entry
    assert data.derived('k', fn) == 2
    data['agent']['category'].add_(0)                                      # in-place write (version counter moves)
    assert data.derived('k', fn) == 3
    assert data.derived('k', fn) == 3
    # a marker standing in for the CUDA event: dropped by every kind of change
    for change in (lambda: data['agent'].__setitem__('extra', torch.zeros(1)), lambda: data['agent'].pop('extra'),
                   lambda: data['agent']['heading'].mul_(1.0), lambda: data.__setitem__('foo', {'a': torch.zeros(1)})):
        data._ready_event, data._ready_versions = 'mark', data._tensor_versions()
        assert data.ready_event() == 'mark'
        change()
        assert data.ready_event() is None
    # the focal rows follow a rewritten category tensor (the cache used to be keyed by the scenario ids alone)
    rows0, _ = sharding.focal_index([0, 1], data)
    cat = data['agent']['category']
    new = cat.clone()
    new[rows0] = 0
    other = [i for i in range(cat.numel()) if i not in rows0.tolist()]
    b = data['agent']['batch']
    pick = [next(i for i in other if int(b[i]) == g) for g in (0, 1)]
    new[pick] = 3
    cat.copy_(new)
    rows1, _ = sharding.focal_index([0, 1], data)
    assert rows1.tolist() == pick


def test_frame_batch_with_a_field_selection_equals_the_full_one_on_those_fields():
    """data.frame_batch(fields=...) (what the ten-frame decoder pass uses: QCNetDecoder.DATA_FIELDS) replicates exactly the
    selected tensors, with the same values, ptr vectors and node counts as the full form, and nothing else."""
    from tempenslearnagg_qcnet_amd.data import frame_batch
    from tempenslearnagg_qcnet_amd.modules import QCNetDecoder
    data = make_batch([3, 4], **SCENE_TINY)
    full = frame_batch(data, 1, 4, 52)
    part = frame_batch(data, 1, 4, 52, fields=QCNetDecoder.DATA_FIELDS)
    assert part.num_graphs == full.num_graphs == 8
    assert set(part.node_types) == set(QCNetDecoder.DATA_FIELDS)
    for ntype, names in QCNetDecoder.DATA_FIELDS.items():
        assert part[ntype]['num_nodes'] == full[ntype]['num_nodes']
        assert torch.equal(part[ntype]['ptr'], full[ntype]['ptr'])
        assert 'batch' not in part[ntype] and 'batch' in full[ntype]
        for name in names:
            assert torch.equal(part[ntype][name], full[ntype][name]), (ntype, name)
        assert set(k for k, v in part[ntype].items() if torch.is_tensor(v)) == set(names) | {'ptr'}


def test_deferred_edge_features_outside_a_section_are_made_at_once():
    """utils.prep.defer / NormalizedEdgeFeatures.deferred: outside a preparation section (and always on the CPU) the thunk
    runs immediately; the object then is what the direct constructor gives."""
    from tempenslearnagg_qcnet_amd.layers import NormalizedEdgeFeatures
    from tempenslearnagg_qcnet_amd.utils import prep
    ran = []
    prep.defer(lambda: ran.append(1))
    assert ran == [1] and prep._state.deferred == []
    rows = torch.randn(5, 128)
    lazy = NormalizedEdgeFeatures.deferred(lambda: rows, (rows,))
    assert lazy.rhat is rows and lazy.num_rows == 5


def test_streaming_shape_plans_are_kept_per_shape_and_bounded():
    """QCNetAgentEncoder._stream_shape_plan: the shape-only index tensors of the streaming plan are built once per
    (agents, frames, steps) shape, reused for the next batch of that shape and dropped least-recently-used first."""
    from tempenslearnagg_qcnet_amd.modules import QCNetAgentEncoder
    from tempenslearnagg_qcnet_amd.synthetic import config_scene
    hp = config_scene('tiny')['hparams']
    enc = QCNetAgentEncoder(dataset='argoverse_v2', input_dim=2, hidden_dim=128, num_historical_steps=50, time_span=10,
                            pl2a_radius=50.0, a2a_radius=50.0, num_freq_bands=64, num_layers=2, num_heads=8, head_dim=16,
                            dropout=0.1)
    D = enc.streaming_plan(20)
    QCNetAgentEncoder._SHAPE_PLANS.clear()
    dev = torch.device('cpu')
    p6 = enc._stream_shape_plan(D, 6, 10, 59, 20, dev)
    assert enc._stream_shape_plan(D, 6, 10, 59, 20, dev) is p6                 # same shape: the same tensors
    p7 = enc._stream_shape_plan(D, 7, 10, 59, 20, dev)
    assert p7 is not p6 and p7['index'].shape == (10, 7, 30) and p6['index'].shape == (10, 6, 30)
    # frame 0 reads the long run's rows (u * A + a); later frames read one special row per agent at offset 20
    A, NA = 6, 6 * 59
    assert int(p6['index'][0, 2, 0]) == 20 * A + 2 and int(p6['index'][3, 2, 5]) == (3 + 25) * A + 2
    assert int(p6['index'][1, 0, 0]) >= NA and int(p6['index'][9, 5, 0]) == NA + 8 * A + 5
    for a in range(8, 8 + QCNetAgentEncoder._SHAPE_PLAN_LIMIT + 2):
        enc._stream_shape_plan(D, a, 10, 59, 20, dev)
    assert len(QCNetAgentEncoder._SHAPE_PLANS) == QCNetAgentEncoder._SHAPE_PLAN_LIMIT
    assert enc._stream_shape_plan(D, 6, 10, 59, 20, dev) is not p6             # evicted, rebuilt
    QCNetAgentEncoder._SHAPE_PLANS.clear()
