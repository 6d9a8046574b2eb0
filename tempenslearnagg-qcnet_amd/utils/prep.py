"""Graph preparation on a second HIP stream.

Everything a module derives from ``data`` alone — masks, radius searches, CSR builds, relative geometry (and, until the
big ones were handed back to the main stream with ``defer``, the positional Fourier embeddings) — is a long chain of
small launches with a few host round trips (``nonzero``, edge counts).  None of
it depends on the learned node features, so ``run_prep`` executes such a section on a per-device side stream: its host
round trips then wait for the side stream only, the host keeps enqueueing, and the small launches overlap the attention
/ MLP kernels of the main stream (those of the previous module, or of the previous ``predict`` call).  The main stream
waits for the section's event before it consumes the results.

The reference has no counterpart (it runs one eager stream); with ``QCN_PREP_STREAM=0`` the sections run inline.
"""
import os
import threading
from typing import Callable, Dict

import torch

ENABLED = os.environ.get('QCN_PREP_STREAM', '1') != '0'

# Heavy launches without a host round trip (the positional Fourier embeddings of a relation: up to 2 M rows each) are NOT
# run on the side stream: they and the attention / chain kernels of the main stream each want a whole CU (130-140 KB of
# LDS), so beside each other they can only take turns.  A section hands such work back with ``defer``: it is enqueued on
# the main stream as soon as the section's results are visible there; the side stream keeps what it is for — the small
# launches and their host round trips.  Measured (cfg3, same box): step 39.4 -> 39.2 ms, chain kernels in the timed
# region 146 -> 157 TF/s, attention launches unchanged.  QCN_PREP_DEFER=0 keeps the embeddings on the side stream.
DEFER_HEAVY = os.environ.get('QCN_PREP_DEFER', '1') != '0'

# Threading: a section's nesting depth and its deferred work belong to the host thread that runs it (one thread per GPU in
# one process is a supported way to drive several devices): thread-local.  The side stream is per (thread, device) for
# the same reason — two threads predicting on one device must not interleave their sections on one stream.
_side: Dict[tuple, 'torch.cuda.Stream'] = {}
_side_lock = threading.Lock()


class _SectionState(threading.local):
    def __init__(self) -> None:
        self.depth = 0
        self.deferred = []


_state = _SectionState()


def side_stream(device: torch.device) -> 'torch.cuda.Stream':
    idx = (threading.get_ident(), device.index if device.index is not None else torch.cuda.current_device())
    stream = _side.get(idx)
    if stream is not None:
        return stream
    with _side_lock:
        if idx in _side:
            return _side[idx]
        # lowest priority the runtime offers: the preparation kernels fill the CUs the main stream leaves idle instead of
        # taking them from the attention / MLP kernels they overlap (QCN_PREP_PRIORITY overrides, 0 = same as main)
        prio = int(os.environ.get('QCN_PREP_PRIORITY', '0'))
        try:
            _side[idx] = torch.cuda.Stream(device=device, priority=prio)
        except (RuntimeError, ValueError):
            _side[idx] = torch.cuda.Stream(device=device)
        return _side[idx]


def _record(obj, stream, seen) -> None:
    """Tell the caching allocator that every tensor reachable from ``obj`` is also used on ``stream``."""
    if torch.is_tensor(obj):
        if obj.is_cuda and obj.data_ptr() not in seen:
            seen.add(obj.data_ptr())
            obj.record_stream(stream)
        return
    if obj is None or isinstance(obj, (int, float, str, bool)):
        return
    if id(obj) in seen:
        return
    seen.add(id(obj))
    if isinstance(obj, dict):
        for v in obj.values():
            _record(v, stream, seen)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _record(v, stream, seen)
    else:
        for name in getattr(type(obj), '__slots__', ()):
            _record(getattr(obj, name, None), stream, seen)
        if hasattr(obj, '__dict__'):
            for v in vars(obj).values():
                _record(v, stream, seen)


_extra: Dict[tuple, list] = {}


def register_stream(device: torch.device, stream: 'torch.cuda.Stream') -> None:
    """Another stream of this host thread that reads a batch's input tensors (the predictor's aggregate stream)."""
    idx = (threading.get_ident(), device.index if device.index is not None else torch.cuda.current_device())
    with _side_lock:
        if stream not in _extra.setdefault(idx, []):
            _extra[idx].append(stream)


def consumer_streams(device: torch.device) -> list:
    """Every stream of this host thread on which the path may read the input tensors of a batch besides the current one:
    the preparation stream and the registered ones.  A loader that allocates the device copy of a batch on a COPY stream
    marks the tensors as used on these (``record_streams``) so that the allocator does not hand their memory to the next
    copy while kernels of an earlier step still read it."""
    idx = (threading.get_ident(), device.index if device.index is not None else torch.cuda.current_device())
    return [side_stream(device)] + list(_extra.get(idx, []))


def record_streams(obj, stream) -> None:
    """Public form of the walk above: every GPU tensor reachable from ``obj`` (dicts, lists, objects) is marked as used on
    ``stream`` for the caching allocator — for results produced on one stream and consumed on another."""
    _record(obj, stream, set())


def _parameter_versions(module) -> tuple:
    """(storage address, version counter) of every parameter and buffer.  The walk over the module tree is done once and the
    tensor list kept on the module (walking 1500 parameters costs milliseconds per call, reading their counters does not):
    ``load_state_dict`` / ``module.to`` / in-place updates keep the Parameter objects, so the list stays valid; after
    REPLACING a sub-module or parameter object call ``forget_parameters(module)``."""
    tensors = module.__dict__.get('_prep_parameter_list')
    if tensors is None:
        tensors = list(module.parameters()) + list(module.buffers())
        module.__dict__['_prep_parameter_list'] = tensors
    return tuple([(t.data_ptr(), t._version) for t in tensors])


def forget_parameters(module) -> None:
    module.__dict__.pop('_prep_parameter_list', None)
    module.__dict__.pop('_prep_parameter_versions', None)


def inputs_ready(device: torch.device, data=None, module=None) -> None:
    """Order the side stream after the inputs (call once where fresh inputs enter, e.g. at the top of ``predict``):
    sections started later may read them.  The side stream waits for the container's ready mark ALONE
    (``data.mark_ready()``, recorded by ``data.to(device)``) — so that the preparation of this batch may overlap
    whatever the main stream still runs for the previous one — only when nothing it reads can still be in flight on the
    main stream: the mark still describes the inputs (``data.ready_event()``: no store entry replaced, no tensor
    written in place since) AND the parameters / buffers of ``module`` (the sections read Fourier weights and type
    embeddings on the side stream) are the ones the previous call saw, i.e. no ``load_state_dict`` / ``model.to`` /
    in-place update since.  In every other case the side stream waits for everything enqueued on the current stream."""
    if MEMO:
        _memo_key[0], _memo_key[1] = (id(data) if data is not None else None), 0
        if data is not None:
            _memo_batches[id(data)] = data
    if ENABLED and device.type == 'cuda':
        event = None
        if data is not None and os.environ.get('QCN_PREP_WAIT_MAIN', '0') == '0':
            event = data.ready_event() if hasattr(data, 'ready_event') else None
        if event is not None and module is not None:
            versions = _parameter_versions(module)
            if module.__dict__.get('_prep_parameter_versions') != versions:
                module.__dict__['_prep_parameter_versions'] = versions
                event = None        # weights changed (or first call): their writes may still be queued on the main stream
        elif module is not None:
            module.__dict__['_prep_parameter_versions'] = _parameter_versions(module)
        if event is not None:
            side_stream(device).wait_event(event)
        else:
            side_stream(device).wait_stream(torch.cuda.current_stream(device))


def defer(fn: Callable[[], None], inputs=()) -> None:
    """Inside a preparation section: run ``fn`` (launches only, no host round trip; reads ``inputs`` produced by the
    section) on the MAIN stream once the section is done.  Outside a section, or with deferral off, ``fn`` runs at once."""
    if ENABLED and DEFER_HEAVY and _state.depth > 0:
        _state.deferred.append((fn, tuple(inputs)))
    else:
        fn()


# Prepared batches (``memoize_sections(True)`` / QCN_PREP_MEMO=1): the results of a batch's sections are kept and handed out
# again when the SAME batch object is predicted again — the small launches and the host round trips of the graph
# construction happen once, every later call is pure launches on device-resident graphs (the deferred heavy launches still
# run).  That is what makes a call capturable into a HIP graph (tests/test_gpu_model.py::
# test_captured_step_replays_bit_identically) and what `bench.py`-style runs over a few resident batches use to price the
# preparation work.  Contract: the batch's tensors and the model's weights do not change while the results
# are kept; ``memoize_sections(False)`` drops them.  Not thread-safe across host threads that share batch objects.
MEMO = os.environ.get('QCN_PREP_MEMO', '0') != '0'
_memo: Dict[tuple, tuple] = {}
_memo_key = [None, 0]
_memo_batches: Dict[int, object] = {}      # the batch objects themselves: a kept batch's id() cannot be reused by another


def memoize_sections(on: bool) -> None:
    global MEMO
    MEMO = bool(on)
    _memo.clear()
    _memo_batches.clear()
    _memo_key[0], _memo_key[1] = None, 0


def run_prep(device: torch.device, fn: Callable[[], object]):
    """Run ``fn`` (which must depend on ``data`` and on results of earlier sections only) on the side stream and make
    the current stream wait for it; returns ``fn()``.  Nested sections run inline in the enclosing one.  Work the
    section handed back with ``defer`` is enqueued on the current stream before this returns."""
    if not ENABLED or _state.depth > 0 or device.type != 'cuda':
        return fn()
    main = torch.cuda.current_stream(device)
    if MEMO and _memo_key[0] is not None:
        key = (_memo_key[0], _memo_key[1])
        _memo_key[1] += 1
        hit = _memo.get(key)
        if hit is not None:
            for thunk, inputs in hit[1]:
                thunk()
            return hit[0]
    side = side_stream(device)
    _state.depth += 1
    try:
        with torch.cuda.stream(side):
            out = fn()
            done = side.record_event()
    except BaseException:
        _state.deferred.clear()  # a section that failed leaves no work behind for the next one
        raise
    finally:
        _state.depth -= 1
    seen: set = set()
    _record(out, main, seen)
    main.wait_event(done)
    pending, _state.deferred = _state.deferred, []
    for thunk, inputs in pending:        # on the main stream, in the order the section asked for them
        _record(inputs, main, seen)
        thunk()
    if MEMO and _memo_key[0] is not None:
        _memo[key] = (out, pending)
    return out
