"""Side HIP streams for the branches of a composable model that do not depend on each other: the fast and the slow pathway of
SlowFast between two lateral connections, the video and the 0D encoder of a fusion model.  Their kernels are small (SlowFast cfg5:
about 1100 launches of 9 us on average per step), so one stream leaves most of the 256 CUs idle; on separate streams the branches
overlap, and a captured step (``src/utils/graphed.py``) keeps that overlap as parallel graph branches.  Autograd runs every
backward node on the stream its forward ran on and orders the streams itself, so only the forward needs the fork/join here.
Default: only inside a stream capture (``MD_STREAMS=1`` always, ``MD_STREAMS=0`` never; same results either way: the kernels and
their order per tensor are unchanged).
"""
import os
from typing import Dict, Iterable, Tuple

import torch

# MD_STREAMS=1: always; MD_STREAMS=0: never; unset (None): only while a stream capture is open.  Eager, the composable models are
# host-bound (the GPU waits for Python), so overlapping branches buys nothing there and the extra event / record_stream calls cost
# host time (cfg4 eager 11.1 -> 12.0 ms with the streams on); inside a capture the same forks become parallel graph branches.
_ENABLED = {"1": True, "0": False}.get(os.environ.get("MD_STREAMS"), None)
# a unit's weight gradient on a helper stream beside its data gradient: bit-identical, measured SLOWER in the captured cfg5 step
# (8.9 ms against 7.4 ms with the pathway streams alone: every fork/join is a cross-stream edge the graph pays for), so off
_UNIT_HELPERS = os.environ.get("MD_STREAMS_WGRAD") == "1"
_SIDE: Dict[Tuple[int, int], "torch.cuda.Stream"] = {}


def prepare(device=None) -> None:
    """Create the side streams of ``device`` now (GraphedStep calls this before its warm-up: nothing is created inside a capture)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    for k in (0, 1):
        side_stream(dev, k)


def enabled(t: torch.Tensor) -> bool:
    if _ENABLED is False or not t.is_cuda:
        return False
    if _ENABLED:
        return True
    dev = t.device.index if t.device.index is not None else torch.cuda.current_device()
    return (dev, 0) in _SIDE and (dev, 1) in _SIDE and torch.cuda.is_current_stream_capturing()


def unit_helpers(t: torch.Tensor) -> bool:
    return _UNIT_HELPERS and enabled(t)


def side_stream(device: torch.device, k: int = 0) -> "torch.cuda.Stream":
    """The k-th side stream of ``device`` (created once per process)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), int(k))
    s = _SIDE.get(key)
    if s is None:
        s = _SIDE[key] = torch.cuda.Stream(device=key[0])
    return s


_HELPER: Dict[Tuple[int, int], "torch.cuda.Stream"] = {}
_POOL: Dict[int, list] = {}
_POOL_SIZE = 6


def helper_stream(device: torch.device) -> "torch.cuda.Stream":
    """A stream that belongs to the CURRENT stream of ``device`` (each stream that asks gets its own, from a pool created on first
    use -- i.e. in an eager warm-up step, never inside a capture): a backward node uses it to run its weight gradient beside its
    data gradient, whichever branch stream the node itself lives on."""
    dev = device.index if device.index is not None else torch.cuda.current_device()
    cur = torch.cuda.current_stream(dev)
    key = (dev, cur.cuda_stream)
    s = _HELPER.get(key)
    if s is None:
        pool = _POOL.get(dev)
        if pool is None:
            pool = _POOL[dev] = [torch.cuda.Stream(device=dev) for _ in range(_POOL_SIZE)]
        n = sum(1 for k in _HELPER if k[0] == dev)
        s = _HELPER[key] = pool[n % _POOL_SIZE]
    return s


def _tensors(xs: Iterable):
    for x in xs:
        t = getattr(x, "t", x)                     # CLAct carries its tensor in .t
        if isinstance(t, torch.Tensor) and t.is_cuda:
            yield t


def used_on(stream: "torch.cuda.Stream", *xs) -> None:
    """Tell the caching allocator that these tensors (allocated under another stream) are read by work queued on ``stream``: their
    memory is not handed out again before that work has run."""
    for t in _tensors(xs):
        t.record_stream(stream)


class fork:
    """``with fork(device, k, inputs) as f:`` runs the body on side stream k, after everything the current stream has queued so
    far.  ``f.mark()`` returns an event for what the side stream has queued up to that point; ``f.join(*outputs)`` (after the
    block) makes the current stream wait for the side stream.
    ``k=None``: the current stream's own helper stream (``helper_stream``) -- unless the current stream is itself one of the side
    streams, in which case the body simply runs in place.  A helper of a side stream would join that side stream only, and a stream
    capture on ROCm 7.2 wants every stream that takes part joined by the capturing stream itself (hipStreamEndCapture crashes
    otherwise), so only the stream the step was started on forks helpers."""

    def __init__(self, device: torch.device, k=0, inputs: Iterable = ()):
        self.main = torch.cuda.current_stream(device)
        if k is not None:
            self.side = side_stream(device, k)
        elif any(s == self.main for s in _SIDE.values()):
            self.side = None
        else:
            self.side = helper_stream(device)
        self.inputs = tuple(inputs)
        self.ctx = torch.cuda.stream(self.side) if self.side is not None else None

    def __enter__(self):
        if self.side is not None:
            self.side.wait_stream(self.main)
            used_on(self.side, *self.inputs)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        return self.ctx.__exit__(*exc) if self.ctx is not None else False

    def mark(self) -> "torch.cuda.Event":
        ev = torch.cuda.Event()
        ev.record(self.side)
        return ev

    def join(self, *outputs) -> None:
        if self.side is not None:
            self.main.wait_stream(self.side)
            used_on(self.main, *outputs)


def wait(ev: "torch.cuda.Event", *xs) -> None:
    """The current stream waits for ``ev``; ``xs`` are the tensors it is about to read from the stream that recorded it."""
    cur = torch.cuda.current_stream()
    cur.wait_event(ev)
    used_on(cur, *xs)
