"""Small host<->device transfers that must not stall the launch thread.

``torch.tensor(list).to(device)`` pays one pageable copy per tensor, and ``torch.tensor(list,
device=...)`` / ``.cpu()`` block the host until the stream has drained.  A decode step of the
eviction path needs half a dozen small integer arrays on the GPU (block tables, lengths, slots)
and, for score-based eviction, one small array back (the victims).  ``PinnedStager`` packs a
step's arrays into one pinned buffer of a small ring and uploads them with a single asynchronous
copy; ``AsyncReadback`` brings a device tensor back through pinned memory behind an event, so
the host waits for that copy only and not for everything queued after it.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

_ALIGN = 16


class PinnedStager:
    """Ring of pinned host buffers; one async H2D copy per ``upload`` call."""

    def __init__(self, depth: int = 8, min_bytes: int = 1 << 20) -> None:
        self.depth = depth
        self.min_bytes = min_bytes
        self._bufs: List[Optional[torch.Tensor]] = [None] * depth
        self._events: List[Optional[torch.cuda.Event]] = [None] * depth
        self._next = 0

    def upload(self, arrays: Sequence[np.ndarray], device: torch.device) -> List[torch.Tensor]:
        """Device copies of ``arrays`` (same dtypes and shapes), all views of one allocation."""
        device = torch.device(device)
        if device.type != "cuda":
            return [torch.from_numpy(np.array(a, copy=True)) for a in arrays]
        offs, total = [], 0
        for a in arrays:
            offs.append(total)
            total += (a.nbytes + _ALIGN - 1) // _ALIGN * _ALIGN
        total = max(total, _ALIGN)
        i = self._next
        self._next = (i + 1) % self.depth
        ev = self._events[i]
        if ev is not None:
            ev.synchronize()  # the copy that last used this buffer; long done in steady state
        buf = self._bufs[i]
        if buf is None or buf.numel() < total:
            # pinned allocations cost milliseconds: size generously (power of two) so that the
            # ring stops growing after the first few steps
            size = self.min_bytes
            while size < total:
                size *= 2
            buf = torch.empty(size, dtype=torch.uint8, pin_memory=True)
            self._bufs[i] = buf
        hn = buf.numpy()
        for a, off in zip(arrays, offs):
            if a.nbytes:
                hn[off:off + a.nbytes] = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
        dev = torch.empty(total, dtype=torch.uint8, device=device)
        dev.copy_(buf[:total], non_blocking=True)
        if ev is None:
            ev = self._events[i] = torch.cuda.Event()
        ev.record()
        out = []
        for a, off in zip(arrays, offs):
            t = dev[off:off + a.nbytes].view(_TORCH_DTYPE[a.dtype.type])
            out.append(t.view(a.shape) if a.ndim != 1 else t)
        return out


_TORCH_DTYPE = {np.int32: torch.int32, np.int64: torch.int64, np.float32: torch.float32,
                np.uint8: torch.uint8}

_stagers: Dict[int, PinnedStager] = {}


def stager(device) -> PinnedStager:
    """The process-wide stager of a device."""
    device = torch.device(device)
    key = -1 if device.type != "cuda" else (
        device.index if device.index is not None else torch.cuda.current_device())
    s = _stagers.get(key)
    if s is None:
        s = _stagers[key] = PinnedStager()
    return s


def upload(arrays: Sequence[np.ndarray], device) -> List[torch.Tensor]:
    return stager(device).upload(arrays, device)


def upload_i32(values, device) -> torch.Tensor:
    return upload([np.asarray(values, dtype=np.int32)], device)[0]


class AsyncReadback:
    """Device tensor -> host list; the copy is queued at construction, ``tolist`` waits for it."""

    def __init__(self, t: torch.Tensor) -> None:
        self._event = None
        if t.is_cuda:
            self._host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            self._host.copy_(t, non_blocking=True)
            self._event = torch.cuda.Event()
            self._event.record()
        else:
            self._host = t.clone()

    def tolist(self) -> list:
        if self._event is not None:
            self._event.synchronize()
            self._event = None
        return self._host.tolist()
