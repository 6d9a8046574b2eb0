"""Reader for the reference's processed scenarios and the collate that feeds the path.

The reference preprocesses every AV2 scenario into one file ``<processed_dir>/<scenario>.pkl`` holding a nested ``dict``
(``datasets/argoverse_v2_dataset.py:203-209``: ``scenario_id``, ``city``, the ``agent`` store of ``get_agent_features``
:273-284 and the map stores of ``get_map_features`` :479-512) written with ``pickle.dump``; ``get`` reads it back with
``pickle.load`` into a ``HeteroData`` (:519-521) and the PyG ``DataLoader`` collates a batch.

Here the file is read WITHOUT executing anything it names: the unpickler resolves only the three globals a plain pickle
of tensors needs (``torch._utils._rebuild_tensor_v2``, ``collections.OrderedDict`` and the storage loader, which is
replaced by ``torch.load(..., weights_only=True)`` on the embedded bytes) and refuses every other global, so a scenario
file cannot run code.  The result is validated against the schema the modules read and wrapped in this package's
``HeteroData``; ``collate_scenarios`` is PyG's collation rule (``data.Batch.from_data_list``).  Preprocessing raw AV2
parquet / map files (``av2`` API, pandas) stays outside the path.
"""
import collections
import io
import os
import pickle
from typing import Any, Callable, Dict, List, Optional, Sequence

import torch

from ..data import Batch, HeteroData


class UnsafeScenarioFile(pickle.UnpicklingError):
    """The file names a global outside the tensors-and-containers allow-list."""


def _load_storage(data: bytes):
    return torch.load(io.BytesIO(data), weights_only=True)


class _ScenarioUnpickler(pickle.Unpickler):
    _ALLOWED = {
        ('torch._utils', '_rebuild_tensor_v2'): torch._utils._rebuild_tensor_v2,
        ('torch.storage', '_load_from_bytes'): _load_storage,
        ('collections', 'OrderedDict'): collections.OrderedDict,
    }

    def find_class(self, module: str, name: str):
        try:
            return self._ALLOWED[(module, name)]
        except KeyError:
            raise UnsafeScenarioFile('scenario file names %s.%s, which a processed scenario never needs' % (module, name))

    def persistent_load(self, pid):
        raise UnsafeScenarioFile('persistent ids are not part of the processed-scenario format')


_NODE_KEYS = {
    'agent': ('num_nodes', 'valid_mask', 'predict_mask', 'type', 'category', 'position', 'heading', 'velocity'),
    'map_polygon': ('num_nodes', 'position', 'orientation', 'type', 'is_intersection'),
    'map_point': ('num_nodes', 'position', 'orientation', 'magnitude', 'type', 'side'),
}
_EDGE_KEYS = {
    ('map_point', 'to', 'map_polygon'): ('edge_index',),
    ('map_polygon', 'to', 'map_polygon'): ('edge_index', 'type'),
}


def _validate(stores: Dict[Any, Any], path: str) -> None:
    if not isinstance(stores, dict):
        raise ValueError('%s: a processed scenario is a dict of stores, got %s' % (path, type(stores).__name__))
    for key, fields in list(_NODE_KEYS.items()) + list(_EDGE_KEYS.items()):
        store = stores.get(key)
        if not isinstance(store, dict):
            raise ValueError('%s: store %r is missing' % (path, key))
        for f in fields:
            if f not in store:
                raise ValueError('%s: store %r has no %r' % (path, key, f))
    for key in _NODE_KEYS:
        store, n = stores[key], int(stores[key]['num_nodes'])
        for f, v in store.items():
            if torch.is_tensor(v) and v.dim() > 0 and v.numel() > 0 and v.size(0) != n:
                raise ValueError('%s: %s.%s has %d rows for %d nodes' % (path, key, f, v.size(0), n))
    for (src, _, dst), fields in _EDGE_KEYS.items():
        ei = stores[(src, _, dst)]['edge_index']
        if ei.dim() != 2 or ei.size(0) != 2 or ei.dtype != torch.int64:
            raise ValueError('%s: edge_index of %s->%s must be [2, E] int64' % (path, src, dst))
        if ei.numel() and (int(ei.min()) < 0 or int(ei[0].max()) >= int(stores[src]['num_nodes'])
                           or int(ei[1].max()) >= int(stores[dst]['num_nodes'])):
            raise ValueError('%s: edge_index of %s->%s is out of range' % (path, src, dst))
    a = stores['agent']
    steps = a['position'].size(1)
    if a['valid_mask'].shape != (int(a['num_nodes']), steps) or a['heading'].shape != (int(a['num_nodes']), steps):
        raise ValueError('%s: agent tensors disagree on [num_nodes, num_steps]' % path)


def load_scenario(path: str, validate: bool = True) -> HeteroData:
    """One processed scenario as ``HeteroData`` (the reference's ``ArgoverseV2Dataset.get``, :519-521)."""
    with open(path, 'rb') as handle:
        stores = _ScenarioUnpickler(handle).load()
    if validate:
        _validate(stores, path)
    return HeteroData(stores)


def save_scenario(data, path: str) -> None:
    """Writes a scenario in the reference's file format (:203-209): ``pickle.dump`` of the nested dict of stores."""
    stores = {k: (dict(data[k]) if isinstance(data[k], dict) else data[k]) for k in data.keys()}
    stores = {k: ({f: (v.detach().cpu() if torch.is_tensor(v) else v) for f, v in s.items()} if isinstance(s, dict) else s)
              for k, s in stores.items()}
    with open(path, 'wb') as handle:
        pickle.dump(stores, handle, protocol=pickle.HIGHEST_PROTOCOL)


def collate_scenarios(scenarios: Sequence[HeteroData]) -> Batch:
    """PyG ``Batch.from_data_list`` for the path's stores (what the reference's DataLoader does per batch)."""
    return Batch.from_data_list(list(scenarios))


class ProcessedScenarioDataset:
    """The processed half of ``ArgoverseV2Dataset`` (:84-160,513-521): ``root/split/processed_qcnet/*.pkl`` (or an explicit
    ``processed_dir``), ``len``, ``get`` / indexing with an optional ``transform``, and ``batches`` as the loader."""

    SPLITS = ('train', 'val', 'test', 'mini', 'densest', 'paper', 'video', 'thesis', 'slide-1', 'slide-2', 'ped', 'motor', 'tiny')

    def __init__(self, root: str, split: str, processed_dir: Optional[str] = None, transform: Optional[Callable] = None) -> None:
        if split not in self.SPLITS:
            raise ValueError(f'{split} is not a valid split')
        self.split = split
        self.transform = transform
        root = os.path.expanduser(os.path.normpath(root))
        self.processed_dir = (os.path.expanduser(os.path.normpath(processed_dir)) if processed_dir is not None
                              else os.path.join(root, split, 'processed_qcnet'))
        names = os.listdir(self.processed_dir) if os.path.isdir(self.processed_dir) else []
        self.processed_file_names = sorted(n for n in names if n.endswith(('pkl', 'pickle'))
                                           and os.path.isfile(os.path.join(self.processed_dir, n)))

    def __len__(self) -> int:
        return len(self.processed_file_names)

    len = __len__

    def get(self, idx: int) -> HeteroData:
        return load_scenario(os.path.join(self.processed_dir, self.processed_file_names[idx]))

    def __getitem__(self, idx: int) -> HeteroData:
        data = self.get(idx)
        return self.transform(data) if self.transform is not None else data

    def batches(self, batch_size: int, indices: Optional[Sequence[int]] = None):
        """Collated batches over ``indices`` (default: all scenarios in file order) — the evaluation loader
        (``evaluate.py:40-43``: no shuffling)."""
        order: List[int] = list(range(len(self))) if indices is None else list(indices)
        for i in range(0, len(order), batch_size):
            yield collate_scenarios([self[j] for j in order[i:i + batch_size]])
