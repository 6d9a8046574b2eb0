"""GPU parity of the evaluation side: metric state updates, frame transforms, NMS baseline.

Metric classes: the oracle (oracle/metrics_oracle.py) follows the reference's ``update`` bodies — PARITY UNPINNED
(torchmetrics / torch_scatter absent, no reference fixtures).  NMS: PINNED against tests/golden/ref_nms.npz, produced by
the reference's own utils/nms.py (oracle/make_golden.py::main_nms).
"""
import math

import numpy as np
import pytest
import torch

from helpers import cpu_state_dict, load_golden, randomize_, rel_err

from oracle import metrics_oracle as mo

from tempenslearnagg_qcnet_amd import metrics, ops
from tempenslearnagg_qcnet_amd.predictors import QCNetSliding, TempEnsLearnAgg
from tempenslearnagg_qcnet_amd.synthetic import config_scene, make_batch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TOL = 1e-5     # fp32 sums of <= 110 terms against the fp32 oracle, relative to the metric's magnitude


def _case(n, k, t, seed, with_heading=False, ragged=True):
    g = torch.Generator().manual_seed(seed)
    target = torch.cumsum(torch.randn(n, t, 2, generator=g), dim=1)
    pred = target.unsqueeze(1) + torch.randn(n, k, t, 2, generator=g) * torch.linspace(0.2, 3.0, k).view(1, k, 1, 1)
    # make the best mode differ between criteria: one mode ends well but wanders, another is steady but ends off
    if k > 1:
        pred[:, 1, -1] = target[:, -1] + 0.01
    prob = torch.softmax(torch.randn(n, k, generator=g), dim=-1)
    valid = torch.ones(n, t, dtype=torch.bool)
    if ragged:
        for i in range(n):
            if i % 4 == 1:
                valid[i, int(torch.randint(1, t, (1,), generator=g)):] = False          # invalid tail (last valid < T - 1)
            elif i % 4 == 2:
                valid[i] = torch.rand(t, generator=g) > 0.4
            elif i % 7 == 3:
                valid[i] = False                                                        # dropped by valid_filter
    if with_heading:
        pred = torch.cat([pred, (torch.rand(n, k, t, 1, generator=g) - 0.5) * 4 * math.pi], dim=-1)
        target = torch.cat([target, (torch.rand(n, t, 1, generator=g) - 0.5) * 2 * math.pi], dim=-1)
    return pred, target, prob, valid


CLASS_CASES = [
    ('Brier', 'FDE'), ('Brier', 'ADE'), ('minADE', 'FDE'), ('minADE', 'ADE'), ('minAHE', 'FDE'), ('minAHE', 'ADE'),
    ('minFDE', 'FDE'), ('minFHE', 'FDE'), ('MR', 'FDE'), ('MR', 'MAXDE'), ('ProbMR', 'FDE'), ('ProbMR', 'MAXDE'),
]


@pytest.mark.parametrize('name,criterion', CLASS_CASES)
@pytest.mark.parametrize('n,k,t,max_guesses,use_prob,keep_final', [
    (37, 6, 50, 6, True, True), (16, 6, 110, 3, True, True), (9, 8, 60, 4, False, True), (21, 6, 50, 6, True, False),
    (5, 1, 7, 6, True, True)])
def test_metric_classes_vs_oracle(name, criterion, n, k, t, max_guesses, use_prob, keep_final):
    heading = name in ('minAHE', 'minFHE')
    pred, target, prob, valid = _case(n, k, t, seed=n * 100 + k, with_heading=heading)
    prob = prob if use_prob else None
    want_sum, want_count = mo.metric_update(name, pred, target, prob, valid, keep_final, criterion, max_guesses)
    m = getattr(metrics, name)(max_guesses=max_guesses)
    kw = {}
    if name in ('Brier', 'minADE', 'minAHE'):
        kw['min_criterion'] = criterion
    if name in ('MR', 'ProbMR'):
        kw['miss_criterion'] = criterion
    d = lambda x: None if x is None else x.to(DEV)
    # two updates: the states accumulate like torchmetrics states
    m.update(d(pred), d(target), d(prob), d(valid), keep_final, **kw)
    m.update(d(pred), d(target), d(prob), d(valid), keep_final, **kw)
    assert int(m.count) == 2 * want_count
    assert abs(float(m.sum) - 2 * float(want_sum)) <= TOL * max(1.0, 2 * abs(float(want_sum)))
    if want_count:
        assert abs(float(m.compute()) - float(want_sum) / want_count) <= TOL * max(1.0, abs(float(want_sum) / want_count))
    m.reset()
    assert float(m.sum) == 0.0 and int(m.count) == 0


def test_metric_without_valid_mask_and_bad_criterion():
    pred, target, prob, _ = _case(11, 6, 50, seed=5, ragged=False)
    want_sum, want_count = mo.metric_update('minADE', pred, target, prob, None)
    m = metrics.minADE()
    m.update(pred.to(DEV), target.to(DEV), prob.to(DEV))
    assert int(m.count) == want_count and abs(float(m.sum) - float(want_sum)) <= TOL * abs(float(want_sum))
    with pytest.raises(ValueError, match='not a valid criterion'):
        m.update(pred.to(DEV), target.to(DEV), prob.to(DEV), min_criterion='XDE')
    with pytest.raises(Exception):
        ops.motion_metrics(pred, target)                       # CPU tensors are refused
    rec = ops.motion_metrics(pred[:0].to(DEV), target[:0].to(DEV))
    assert rec.shape == (0, ops.METRIC_RECORD)


def test_average_meter():
    m = metrics.AverageMeter()
    v = torch.arange(10, dtype=torch.float32, device=DEV)
    m.update(v)
    m.update(v[:2])
    assert float(m.compute()) == pytest.approx((45 + 1) / 12)


def test_frame_transforms_vs_oracle():
    g = torch.Generator().manual_seed(3)
    a, total, k = 23, 110, 6
    for pos_dim in (3, 2):
        position = torch.cumsum(torch.randn(a, total, pos_dim, generator=g), dim=1) * 3
        heading = (torch.rand(a, total, generator=g) - 0.5) * 2 * math.pi
        want = mo.get_ground_truth(position, heading, 50, 60, time_shift=10)
        got = ops.to_agent_frame(position.to(DEV), heading.to(DEV), 59, 60)
        assert got.shape == want.shape == (a, 50, 4)
        assert rel_err(got[..., :3], want[..., :3]) < TOL
        assert float((got[..., 3].cpu() - want[..., 3]).abs().max()) < 1e-5
        traj = torch.randn(a, k, 60, 4, generator=g) * 20
        pi = torch.randn(a, k, generator=g)
        eval_mask = torch.rand(a, generator=g) < 0.3
        eval_mask[0] = True
        for ts in (10, 3):
            want_traj, want_pi = mo.postprocess(position, heading, eval_mask, traj, pi, ts)
            rows = eval_mask.nonzero(as_tuple=True)[0].to(DEV)
            got_traj = ops.to_world_frame(traj.to(DEV), position.to(DEV), heading.to(DEV), 49 + ts, rows)
            got_traj = got_traj[:, :, 10 - ts: 10 - ts + 50]
            assert got_traj.shape == want_traj.shape
            assert rel_err(got_traj, want_traj) < TOL
    # a round trip: world -> agent frame -> world
    back = ops.to_world_frame(ops.to_agent_frame(position.to(DEV), heading.to(DEV), 59, 60).unsqueeze(1), position.to(DEV),
                              heading.to(DEV), 59)
    assert rel_err(back[:, 0], position[:, 60:, :2]) < TOL


def test_nms_vs_reference_golden():
    G = load_golden('ref_nms.npz')
    trajs, conf, gt = (torch.from_numpy(G[k]) for k in ('trajs', 'conf', 'gt'))
    selected, score = ops.nms_select(trajs.to(DEV), conf.to(DEV), gt.to(DEV))
    assert selected.dtype == torch.int32 and selected.shape == (trajs.size(0), 10, 6)
    assert np.array_equal(selected.cpu().numpy().astype(np.int64), G['selected'])            # bit-exact vs utils/nms.py
    assert np.array_equal(conf.numpy()[np.arange(trajs.size(0))[:, None, None], G['selected']], G['selected_conf'])
    for a in range(trajs.size(0)):
        for w in range(10):
            fde, ade, miss = mo.calculate_metrics(G['trajs'][a][G['selected'][a, w]], G['gt'][a])
            got = score[a, w].cpu().numpy()
            assert abs(got[0] - fde) < 1e-5 * max(1, fde) and abs(got[1] - ade) < 1e-5 * max(1, ade) and int(got[2]) == miss


def test_nms_ties_and_small_windows():
    # equal confidences are taken higher index first; three far-apart end points and nine on top of the first
    n_c = 12
    ends = torch.zeros(n_c, 2)
    ends[1], ends[2] = torch.tensor([10.0, 0.0]), torch.tensor([0.0, 10.0])
    trajs = (ends.view(1, n_c, 1, 2) * torch.linspace(0, 1, 5).view(1, 1, 5, 1)).contiguous()
    conf = torch.full((1, n_c), 0.5)
    gt = trajs[:, 2].clone()
    sel, score = ops.nms_select(trajs.to(DEV), conf.to(DEV), gt.to(DEV), cand_per_window=6, n_select=6)
    want = [mo.nms(trajs[0, :6 * w].numpy(), np.arange(6 * w, dtype=np.float32) * 0 + 0.5 + np.arange(6 * w) * 1e-6)[0]
            for w in (1, 2)]
    for w in range(2):
        assert sel[0, w].cpu().tolist() == want[w].tolist()
    assert float(score[0, 0, 0]) == 0.0 and float(score[0, 0, 2]) == 0.0
    with pytest.raises(Exception):
        ops.nms_select(trajs.to(DEV), conf.to(DEV), gt.to(DEV), cand_per_window=4, n_select=6)


def _models(cfg, seed=0):
    torch.manual_seed(seed)
    model = randomize_(TempEnsLearnAgg(**cfg['hparams']), seed=seed + 1).eval()
    randomize_(model.temporal_aggregate_layer, seed=seed + 2)
    return model


def test_validation_step_metrics_vs_oracle():
    """TempEnsLearnAgg.validation_step: the oracle's metrics on the same predictions, both frame transforms included."""
    cfg = config_scene('tiny')
    model = _models(cfg).to(DEV)
    hp = cfg['hparams']
    for ids in ([0, 1, 2], [3, 4]):
        batch = make_batch(ids, **cfg['scene'])
        out = model.validation_step(batch.to(DEV))
        T = hp['num_historical_steps']
        target = mo.get_ground_truth(batch['agent']['position'], batch['agent']['heading'], T, hp['num_future_steps'])
        want = mo.validation_metrics(out['traj_refine'].cpu(), out['pi'].cpu(), target,
                                     batch['agent']['predict_mask'][:, T + 10:], batch['agent']['category'] == 3)
        rec = out['metric_records'].cpu()
        cols = {'Brier': 'brier_fde', 'minADE': 'ade_fde', 'minADE_best': 'ade_ade', 'minAHE': 'ahe_fde', 'minFDE': 'fde',
                'minFHE': 'fhe', 'MR': 'miss_fde'}
        for name, (s, c) in want.items():
            got = float(rec[:, ops.METRIC_COLUMNS[cols[name]]].double().sum())
            assert abs(got - float(s)) <= 1e-4 * max(1.0, abs(float(s))), name
            assert int(rec[:, ops.METRIC_COLUMNS['counted']].sum()) == c
    values = model.metrics.compute()
    assert set(values) == {'val_' + n for n in model.metrics.NAMES}
    assert all(torch.isfinite(v) for v in values.values())
    assert int(model.metrics.minFDE.count) == 5                       # one focal agent per scenario, five scenarios


def test_sliding_nms_baseline_vs_oracle():
    """QCNetSliding.validation_step (the NMS ensembling baseline): selections and scores against the oracle's NMS run on
    the same world-frame candidates."""
    cfg = config_scene('tiny')
    torch.manual_seed(0)
    model = randomize_(QCNetSliding(**cfg['hparams']), seed=4).eval().to(DEV)
    batch = make_batch([0, 1, 2], **cfg['scene'])
    out = model.validation_step(batch.to(DEV))
    trajs = out['trajs'].cpu().numpy()
    clsfs = out['clsfs'].cpu().numpy()
    gt = batch['agent']['position'][batch['agent']['category'] == 3, 60:, :2].numpy()
    assert trajs.shape[:3] == (3, 10, 6) and trajs.shape[3] == 50
    for a in range(trajs.shape[0]):
        for w in range(1, 11):
            cand = trajs[a, :w].reshape(-1, 50, 2)
            sel, _ = mo.nms(cand, clsfs[a, :w].reshape(-1))
            assert out['selected'][a, w - 1].cpu().tolist() == sel.tolist()
            fde, ade, miss = mo.calculate_metrics(cand[sel], gt[a])
            got = out['score'][a, w - 1].cpu().numpy()
            assert abs(got[0] - fde) < 1e-4 * max(1, fde) and abs(got[1] - ade) < 1e-4 * max(1, ade) and int(got[2]) == miss


def test_target_builder_on_gpu_equals_host_transform():
    from tempenslearnagg_qcnet_amd.transforms import TargetBuilder
    cfg = config_scene('tiny')
    host = make_batch([0, 1], **cfg['scene'])
    dev = make_batch([0, 1], **cfg['scene']).to(DEV)
    tb = TargetBuilder(50, 60)
    want = tb(host)['agent']['target']
    got = tb(dev)['agent']['target']
    assert got.is_cuda and got.shape == want.shape
    assert rel_err(got[..., :3], want[..., :3]) < TOL and float((got[..., 3].cpu() - want[..., 3]).abs().max()) < 1e-5
