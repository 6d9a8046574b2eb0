"""Scenario containers for the QCNet hot path.

The reference feeds its modules PyG ``HeteroData`` / ``Batch`` objects
(``datasets/argoverse_v2_dataset.py:519-521`` builds one per scenario, the PyG
``DataLoader`` collates them).  PyG is not part of this stack, so these two small
classes give the modules the exact access pattern they use:

* ``data['agent']['position']``, ``data['map_point', 'to', 'map_polygon']['edge_index']``
  (``modules/qcnet_map_encoder.py:96-144``),
* ``data.num_graphs`` and ``isinstance(data, Batch)`` switches
  (``modules/qcnet_agent_encoder.py:148-157``, ``modules/qcnet_decoder.py:174-199``).

Collation follows PyG's rule: node stores are concatenated, ``edge_index`` of a
``(src, rel, dst)`` store is offset by the cumulative node counts of ``src`` / ``dst``,
``num_nodes`` is summed, and every node store gains ``batch`` and ``ptr`` vectors.
"""
import weakref
from typing import Any, Dict, Iterable, List, Optional, Tuple, Union

import torch

Key = Union[str, Tuple[str, str, str]]

# Per-scenario scalar entries that collation turns into a per-graph tensor / list.
_PER_GRAPH_KEYS = ('av_index',)


class _Store(dict):
    """A node / edge store.  Any change of its entries (``data['agent']['position'] = t``, ``del``, ``update`` ...) is new
    input for the container that owns it: the owner's ready mark and derived-quantity cache are dropped (see
    ``HeteroData.mark_ready``).  Writes INTO a tensor are caught by the version check of ``HeteroData.ready_event``."""

    __slots__ = ('_owner',)

    def __init__(self, *args, owner=None, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self._owner = weakref.ref(owner) if owner is not None else None

    def _touch(self) -> None:
        owner = self._owner() if self._owner is not None else None
        if owner is not None:
            owner._invalidate()

    def __setitem__(self, key, value) -> None:
        super().__setitem__(key, value)
        self._touch()

    def __delitem__(self, key) -> None:
        super().__delitem__(key)
        self._touch()

    def update(self, *args, **kwargs) -> None:
        super().update(*args, **kwargs)
        self._touch()

    def pop(self, *args):
        out = super().pop(*args)
        self._touch()
        return out

    def popitem(self):
        out = super().popitem()
        self._touch()
        return out

    def clear(self) -> None:
        super().clear()
        self._touch()

    def setdefault(self, key, default=None):
        if key not in self:
            self[key] = default
        return self[key]

    def __reduce__(self):
        return (dict, (dict(self),))


class HeteroData:
    """One scenario: a dict of node stores (str keys) and edge stores (3-tuple keys)."""

    def __init__(self, stores: Dict[Key, Dict[str, Any]] = None) -> None:
        self._stores: Dict[Key, Dict[str, Any]] = {}
        for key, store in (stores or {}).items():
            if isinstance(store, dict):
                self._stores[key] = _Store(store, owner=self)
            else:  # global attribute such as scenario_id
                self._stores[key] = store

    def _invalidate(self) -> None:
        self.__dict__.pop('_ready_event', None)
        self.__dict__.pop('_ready_versions', None)
        self.__dict__.pop('_derived', None)

    def __getitem__(self, key) -> Dict[str, Any]:
        if isinstance(key, list):
            key = tuple(key)
        if key not in self._stores:
            self._stores[key] = _Store(owner=self)
        return self._stores[key]

    def __setitem__(self, key, value) -> None:
        self._stores[key] = _Store(value, owner=self) if isinstance(value, dict) else value
        self._invalidate()                           # a replaced store is new input: see mark_ready

    def __contains__(self, key) -> bool:
        return key in self._stores

    def keys(self) -> Iterable[Key]:
        return self._stores.keys()

    @property
    def node_types(self) -> List[str]:
        return [k for k, v in self._stores.items() if isinstance(k, str) and isinstance(v, dict)]

    @property
    def edge_types(self) -> List[Tuple[str, str, str]]:
        return [k for k in self._stores if isinstance(k, tuple)]

    def _map(self, fn):
        out = self.__class__.__new__(self.__class__)
        out.__dict__.update({k: v for k, v in self.__dict__.items()
                             if k not in ('_stores', '_ready_event', '_ready_versions', '_derived')})
        out._stores = {}
        for key, store in self._stores.items():
            if isinstance(store, dict):
                out._stores[key] = _Store({n: (fn(v) if torch.is_tensor(v) else v) for n, v in store.items()}, owner=out)
            else:
                out._stores[key] = fn(store) if torch.is_tensor(store) else store
        return out

    def to(self, device, non_blocking: bool = False):
        out = self._map(lambda t: t.to(device, non_blocking=non_blocking))
        if torch.device(device).type == 'cuda':
            out.mark_ready()
        return out

    def _tensor_versions(self) -> tuple:
        """(key, name, storage address, version counter) of every input tensor: changes when a tensor is replaced or
        written in place by any torch op (on any stream)."""
        out = []
        for key, store in self._stores.items():
            if isinstance(store, dict):
                for name, v in store.items():
                    if torch.is_tensor(v):
                        out.append((key, name, v.data_ptr(), v._version))
            elif torch.is_tensor(store):
                out.append((key, None, store.data_ptr(), store._version))
        return tuple(out)

    def mark_ready(self):
        """Record, on the current stream, the point from which the inputs of this container are final.  The predictors'
        preparation stream (utils/prep.py) then waits for THIS event instead of for everything enqueued on the main
        stream, so the graph preparation of one batch overlaps the attention / MLP kernels of the previous batch.
        ``to(device)`` records it (all tensors of the copy are new).  The mark is dropped — and the preparation stream
        falls back to waiting for the whole main stream — as soon as a store entry is replaced or an input tensor is
        written in place (its version counter moved): see ``ready_event``.  Only writes torch cannot see (a raw kernel
        on ``data_ptr()``) need an explicit ``mark_ready()`` afterwards."""
        dev = None
        for store in self._stores.values():
            for v in (store.values() if isinstance(store, dict) else (store,)):
                if torch.is_tensor(v) and v.is_cuda:
                    dev = v.device
                    break
            if dev is not None:
                break
        self._invalidate()
        if dev is not None:
            self._ready_event = torch.cuda.current_stream(dev).record_event()
            self._ready_versions = self._tensor_versions()
        return self

    def ready_event(self) -> Optional['torch.cuda.Event']:
        """The ready mark if it still describes the inputs (no store entry replaced, no tensor written since), else None."""
        event = self.__dict__.get('_ready_event')
        if event is None:
            return None
        if self.__dict__.get('_ready_versions') != self._tensor_versions():
            self._invalidate()
            return None
        return event

    def derived(self, name, fn):
        """Per-batch quantities computed from the inputs alone (e.g. the focal-track rows), computed once per container
        and recomputed when an input tensor was replaced or written in place since."""
        versions = self._tensor_versions()
        if self.__dict__.get('_derived_versions') != versions:
            self.__dict__['_derived'] = {}
            self.__dict__['_derived_versions'] = versions
        cache = self.__dict__.setdefault('_derived', {})
        if name not in cache:
            cache[name] = fn()
        return cache[name]

    def cpu(self):
        return self.to('cpu')

    def pin_memory(self):
        """Every tensor in page-locked host memory (torch's DataLoader calls this on custom batch objects when
        ``pin_memory=True``): ``to(device, non_blocking=True)`` is then asynchronous."""
        return self._map(lambda t: t.pin_memory() if not t.is_cuda else t)

    def clone(self):
        return self._map(lambda t: t.clone())


class Batch(HeteroData):
    """A collated list of scenarios (PyG ``Batch.from_data_list`` semantics)."""

    def __init__(self, stores=None, num_graphs: int = 1) -> None:
        super().__init__(stores)
        self.num_graphs = num_graphs

    @classmethod
    def from_data_list(cls, data_list: List[HeteroData]) -> 'Batch':
        assert len(data_list) > 0
        first = data_list[0]
        out = cls(num_graphs=len(data_list))
        node_counts: Dict[str, List[int]] = {}
        for ntype in first.node_types:
            counts = [int(d[ntype]['num_nodes']) for d in data_list]
            node_counts[ntype] = counts
            store: Dict[str, Any] = {}
            for name, value in first[ntype].items():
                if name == 'num_nodes':
                    store[name] = sum(counts)
                elif name in _PER_GRAPH_KEYS:
                    store[name] = torch.tensor([int(d[ntype][name]) for d in data_list], dtype=torch.long)
                elif torch.is_tensor(value):
                    store[name] = torch.cat([d[ntype][name] for d in data_list], dim=0)
                elif isinstance(value, list):
                    store[name] = [d[ntype][name] for d in data_list]
                else:
                    store[name] = [d[ntype][name] for d in data_list]
            store['batch'] = torch.repeat_interleave(torch.arange(len(data_list), dtype=torch.long),
                                                     torch.tensor(counts, dtype=torch.long))
            ptr = torch.zeros(len(data_list) + 1, dtype=torch.long)
            ptr[1:] = torch.tensor(counts, dtype=torch.long).cumsum(0)
            store['ptr'] = ptr
            out[ntype] = store
        for etype in first.edge_types:
            src, _, dst = etype
            src_off = [0]
            dst_off = [0]
            for c in node_counts[src][:-1]:
                src_off.append(src_off[-1] + c)
            for c in node_counts[dst][:-1]:
                dst_off.append(dst_off[-1] + c)
            store = {}
            for name, value in first[etype].items():
                if name == 'edge_index':
                    parts = []
                    for d, so, do in zip(data_list, src_off, dst_off):
                        ei = d[etype]['edge_index']
                        parts.append(ei + ei.new_tensor([[so], [do]]))
                    store[name] = torch.cat(parts, dim=1)
                elif torch.is_tensor(value):
                    store[name] = torch.cat([d[etype][name] for d in data_list], dim=0)
                else:
                    store[name] = [d[etype][name] for d in data_list]
            out[etype] = store
        for key in first.keys():
            if key not in out._stores:  # global attributes (scenario_id, city, ...)
                out[key] = [d[key] for d in data_list]
        return out


def frame_batch(data: HeteroData, first_shift: int, n_frames: int, num_steps: int, fields=None) -> Batch:
    """The sliding frames of one batch as ONE larger batch: scenario (f, b) is scenario b seen ``f`` steps later.

    ``fields``: {node type: names} — replicate only these tensors (and only these node types) and leave the ``batch``
    vectors out (their ``repeat_interleave`` is a host round trip); default: everything, like ``Batch.from_data_list``.

    Agent time series are shifted so that ``module(frame_batch(data, s0, F, n), ..., time_shift=s0)`` computes for rows
    [f * A, (f + 1) * A) what ``module(data, ..., time_shift=s0 + f)`` computes (``num_steps`` = steps kept after the
    shift; the decoder reads up to ``num_historical_steps + time_shift``).  ``predict_mask`` is only ever reduced with
    ``.any(-1)`` (modules/qcnet_decoder.py:156) and is replicated unshifted; map stores are replicated per frame."""
    F = int(n_frames)
    num_graphs = getattr(data, 'num_graphs', 1)
    out = Batch(num_graphs=num_graphs * F)
    agent = data['agent']
    A = int(agent['num_nodes'])
    store = {'num_nodes': A * F}
    for name, value in agent.items():
        if not torch.is_tensor(value) or name in ('batch', 'ptr'):
            continue
        if fields is not None and name not in fields.get('agent', ()):
            continue
        if name in ('position', 'heading', 'velocity', 'valid_mask'):
            store[name] = torch.cat([value[:, f:f + num_steps] for f in range(F)], dim=0)
        else:
            store[name] = value.repeat(F, *([1] * (value.dim() - 1)))
    out['agent'] = store
    node_types = [t for t in data.node_types if fields is None or t in fields]
    for ntype in node_types:
        if ntype != 'agent':
            n = int(data[ntype]['num_nodes'])
            out[ntype] = {'num_nodes': n * F}
            for name, value in data[ntype].items():
                if torch.is_tensor(value) and name not in ('batch', 'ptr') and (fields is None or name in fields[ntype]):
                    out[ntype][name] = value.repeat(F, *([1] * (value.dim() - 1)))
    for ntype in node_types:
        n = int(data[ntype]['num_nodes'])
        dev = data[ntype]['position'].device
        ptr = data[ntype].get('ptr')
        ptr = ptr.to(dev) if ptr is not None else torch.tensor([0, n], device=dev)
        offs = (torch.arange(F, device=dev).unsqueeze(1) * n + ptr[:-1].unsqueeze(0)).reshape(-1)
        out[ntype]['ptr'] = torch.cat([offs, ptr.new_tensor([n * F])])
        if fields is None:
            counts = out[ntype]['ptr'][1:] - out[ntype]['ptr'][:-1]
            out[ntype]['batch'] = torch.repeat_interleave(torch.arange(num_graphs * F, device=dev), counts)
    return out
