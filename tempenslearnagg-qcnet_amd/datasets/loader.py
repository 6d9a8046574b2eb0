"""The evaluation loader: batches built ahead of the GPU by worker processes, handed over in pinned memory.

Counterpart of ``DataLoader(dataset, batch_size, shuffle=False, num_workers, pin_memory, persistent_workers)`` in the
reference's ``evaluate.py:42-43`` (PyG's DataLoader = torch's with ``Batch.from_data_list`` as collate).  Scenarios come
from any indexable source — ``ProcessedScenarioDataset`` (the reference's processed files) or a function of the scenario
id (``synthetic.make_scenario``) — and are collated by ``data.Batch.from_data_list`` inside the workers; the main process
receives ``(ids, batch)`` with every tensor in pinned host memory, so ``batch.to(device, non_blocking=True)`` is a queue
of asynchronous copies and the host never waits for a batch that was built while the previous ones ran.

Workers are forked (the default start method): they touch CPU tensors only, never the GPU.
"""
from typing import Callable, Iterator, List, Sequence, Tuple

import torch
from torch.utils.data import DataLoader, Dataset

from ..data import Batch, HeteroData


class _ScenarioSource(Dataset):
    def __init__(self, ids: Sequence[int], get: Callable[[int], HeteroData]) -> None:
        self.ids = list(ids)
        self.get = get

    def __len__(self) -> int:
        return len(self.ids)

    def __getitem__(self, i: int):
        return self.ids[i], self.get(self.ids[i])


def _collate(items) -> Tuple[List[int], Batch]:
    return [i for i, _ in items], Batch.from_data_list([d for _, d in items])


def _worker_init(_worker_id: int) -> None:
    torch.set_num_threads(1)        # one core per worker: scenario assembly is many small tensor ops


def scenario_loader(ids: Sequence[int], get_scenario: Callable[[int], HeteroData], batch_size: int, num_workers: int = 4,
                    pin_memory: bool = True, prefetch_factor: int = 2) -> Iterator[Tuple[List[int], Batch]]:
    """Iterates ``(scenario ids, collated CPU batch)`` over ``ids`` in order, ``batch_size`` at a time (ragged last batch).
    ``num_workers = 0`` builds every batch in the calling thread (the synchronous form)."""
    source = _ScenarioSource(ids, get_scenario)
    if len(source) == 0:
        return iter(())
    kwargs = {}
    if num_workers > 0:
        kwargs = dict(persistent_workers=False, prefetch_factor=prefetch_factor, worker_init_fn=_worker_init)
    loader = DataLoader(source, batch_size=batch_size, shuffle=False, num_workers=num_workers, collate_fn=_collate,
                        pin_memory=pin_memory and torch.cuda.is_available(), drop_last=False, **kwargs)
    return iter(loader)
