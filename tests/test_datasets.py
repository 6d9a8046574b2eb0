"""CPU: the processed-scenario reader — the reference's file format round-trips, nothing in a file is
executed, collation of loaded scenarios equals collation of the originals, TargetBuilder equals the oracle's ground truth."""
import os
import pickle

import pytest
import torch

from oracle import metrics_oracle as mo

from tempenslearnagg_qcnet_amd.data import Batch, HeteroData
from tempenslearnagg_qcnet_amd.datasets import (ProcessedScenarioDataset, UnsafeScenarioFile, collate_scenarios, load_scenario,
                                                save_scenario)
from tempenslearnagg_qcnet_amd.synthetic import SCENE_TINY, make_batch, make_scenario
from tempenslearnagg_qcnet_amd.transforms import TargetBuilder


def _same(a, b):
    assert set(a.keys()) == set(b.keys())
    for key in a.keys():
        sa, sb = a[key], b[key]
        if not isinstance(sa, dict):
            assert sa == sb
            continue
        assert set(sa) == set(sb), key
        for f in sa:
            if torch.is_tensor(sa[f]):
                assert sa[f].dtype == sb[f].dtype and torch.equal(sa[f], sb[f]), (key, f)
            else:
                assert sa[f] == sb[f], (key, f)


def _write_split(tmp_path, ids):
    d = tmp_path / 'val' / 'processed_qcnet'
    d.mkdir(parents=True)
    originals = []
    for i in ids:
        s = make_scenario(i, **SCENE_TINY)
        s['scenario_id'] = 'scenario-%04d' % i
        s['city'] = 'synthetic'
        save_scenario(s, str(d / ('%04d.pkl' % i)))
        originals.append(s)
    (d / 'notes.txt').write_text('not a scenario')
    return originals


def test_reference_format_round_trip_and_collation(tmp_path):
    originals = _write_split(tmp_path, [3, 0, 5])
    ds = ProcessedScenarioDataset(str(tmp_path), 'val')
    assert len(ds) == 3 and ds.processed_file_names == ['0000.pkl', '0003.pkl', '0005.pkl']
    by_id = {s['scenario_id']: s for s in originals}
    for i in range(len(ds)):
        got = ds[i]
        assert isinstance(got, HeteroData)
        _same(got, by_id[got['scenario_id']])
    # the file itself is what the reference writes: a plain pickle of the nested dict (datasets/argoverse_v2_dataset.py:208-209)
    with open(os.path.join(ds.processed_dir, '0000.pkl'), 'rb') as fh:
        raw = pickle.load(fh)                  # a file this test wrote itself
    assert isinstance(raw, dict) and ('map_point', 'to', 'map_polygon') in raw and raw['agent']['position'].dim() == 3
    batches = list(ds.batches(2))
    assert [b.num_graphs for b in batches] == [2, 1] and isinstance(batches[0], Batch)
    want = make_batch([0, 3], **SCENE_TINY)
    for key in ('agent', 'map_polygon', 'map_point', ('map_point', 'to', 'map_polygon'), ('map_polygon', 'to', 'map_polygon')):
        for f, v in want[key].items():
            if torch.is_tensor(v):
                assert torch.equal(batches[0][key][f], v), (key, f)
    assert batches[0]['scenario_id'] == ['scenario-0000', 'scenario-0003']
    with pytest.raises(ValueError, match='not a valid split'):
        ProcessedScenarioDataset(str(tmp_path), 'nope')
    assert len(ProcessedScenarioDataset(str(tmp_path), 'train')) == 0


class _Boom:
    def __reduce__(self):
        return (os.system, ('echo pwned > /dev/null',))


def test_reader_executes_nothing_from_the_file(tmp_path):
    p = tmp_path / 'evil.pkl'
    with open(p, 'wb') as fh:
        pickle.dump({'agent': {'position': _Boom()}}, fh)
    with pytest.raises(UnsafeScenarioFile, match='system'):
        load_scenario(str(p))
    # numpy payloads are not part of the format either (the reference stores torch tensors only)
    import numpy as np
    with open(p, 'wb') as fh:
        pickle.dump({'agent': {'position': np.zeros(3)}}, fh)
    with pytest.raises(UnsafeScenarioFile):
        load_scenario(str(p))


def test_reader_validates_the_schema(tmp_path):
    s = make_scenario(1, **SCENE_TINY)
    p = str(tmp_path / 'a.pkl')
    bad = s.clone()
    del bad['agent']['heading']
    save_scenario(bad, p)
    with pytest.raises(ValueError, match="no 'heading'"):
        load_scenario(p)
    bad = s.clone()
    bad['map_point', 'to', 'map_polygon']['edge_index'] = bad['map_point', 'to', 'map_polygon']['edge_index'] + 10 ** 6
    save_scenario(bad, p)
    with pytest.raises(ValueError, match='out of range'):
        load_scenario(p)
    bad = s.clone()
    bad['agent']['velocity'] = bad['agent']['velocity'][:-1]
    save_scenario(bad, p)
    with pytest.raises(ValueError, match='rows for'):
        load_scenario(p)
    assert isinstance(load_scenario(p, validate=False), HeteroData)


def test_target_builder_matches_oracle_ground_truth(tmp_path):
    originals = _write_split(tmp_path, [2])
    ds = ProcessedScenarioDataset(str(tmp_path), 'val', transform=TargetBuilder(50, 60))
    data = ds[0]
    pos, head = originals[0]['agent']['position'], originals[0]['agent']['heading']
    # transforms/target_builder.py:30-46 is get_ground_truth without the ten-step shift and with all 60 future steps
    want = mo.get_ground_truth(pos, head, 50, 70, time_shift=0)
    assert data['agent']['target'].shape == (pos.size(0), 60, 4)
    assert torch.allclose(data['agent']['target'], want, rtol=1e-5, atol=1e-5)
    assert collate_scenarios([data, data])['agent']['target'].shape == (2 * pos.size(0), 60, 4)
