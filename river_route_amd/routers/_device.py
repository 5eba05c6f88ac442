"""Device-resident handling of one input file: buffers that are freed together, and the route-to-float32 sequence the
routers share (post-processing of river_route/routers/TransformMuskingum.py:128-142 on the GPU)."""
from __future__ import annotations

import numpy as np

from .._lib import RR_E_ALLOC, RR_E_UNSUPPORTED, RRError
from ..engine import DeviceBuffer, resample_cast_dev

__all__ = ['Arena', 'DeviceOutOfMemory', 'float32_rows']


class DeviceOutOfMemory(RuntimeError):
    """The file does not fit on the card next to the engine's own buffers: route it through the host-array path."""


class Arena:
    """Device buffers of one file; everything is released on exit, whatever happened in between."""

    def __init__(self, device: int):
        self.device, self._held = device, []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        for b in self._held:
            b.free()
        self._held.clear()

    def empty(self, nbytes: int) -> DeviceBuffer:
        try:
            buf = DeviceBuffer(max(int(nbytes), 8), self.device)
        except RRError as e:
            if e.code == RR_E_ALLOC:
                raise DeviceOutOfMemory(str(e)) from e
            raise
        self._held.append(buf)
        return buf

    def put(self, array: np.ndarray) -> DeviceBuffer:
        a = np.ascontiguousarray(array)
        return self.empty(a.nbytes).upload(a)

    def release(self, buf: DeviceBuffer) -> None:
        buf.free()
        self._held.remove(buf)


def float32_rows(arena: Arena, rows: int, n: int, factor: int, fused, plain) -> np.ndarray:
    """(rows / factor, n) float32 discharge of one routed file.

    `fused(d_f32)` enqueues the routing call that writes float32 rows itself (rr_*_route_f32_dev: the mean over `factor`
    rows and the cast happen in the pass that returns the records to params order); where the engine reports that form
    does not apply, `plain(d_f64)` routes into a float64 array and rr_resample_cast_dev reduces it."""
    d_f32 = arena.empty((rows // factor) * n * 4)
    try:
        fused(d_f32)
    except RRError as e:
        if e.code == RR_E_ALLOC:
            raise DeviceOutOfMemory(str(e)) from e
        if e.code != RR_E_UNSUPPORTED:
            raise
        d_f64 = arena.empty(rows * n * 8)
        try:
            plain(d_f64)
        except RRError as e2:
            if e2.code == RR_E_ALLOC:
                raise DeviceOutOfMemory(str(e2)) from e2
            raise
        resample_cast_dev(d_f64, rows, n, factor, d_f32, arena.device)
        out = d_f32.download(np.float32, (rows // factor, n))
        arena.release(d_f64)
        return out
    return d_f32.download(np.float32, (rows // factor, n))
