"""
Python handle on an rr_plan (include/rr_hip.h): network layout + device-resident coefficients + route calls.
Host arrays are numpy; `*_dev` methods take raw device addresses / torch tensors and a HIP stream handle.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import RRError, check, ptr

__all__ = ['Plan', 'RRError', 'uh_convolve', 'uh_convolve_dev', 'runoff_to_qlateral', 'DeviceBuffer', 'partition_forest', 'synchronize',
           'resample_cast_dev', 'copy_bandwidth', 'runoff_to_qlateral_dev', 'rows_upload', 'rows_download']


MODE_RAPID, MODE_MUSKINGUM, MODE_UNIT = 0, 1, 2      # include/rr_hip.h: RR_MODE_*


def _f64(a, name):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if not a.flags['WRITEABLE']:
        a = a.copy()
    return a


def _inplace_f64(a, name, shape):
    if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags['C_CONTIGUOUS'] and a.flags['WRITEABLE']):
        raise TypeError(f'{name} must be a writeable C-contiguous float64 numpy array (it is updated in place)')
    if a.shape != shape:
        raise ValueError(f'{name} has shape {a.shape}, expected {shape}')
    return a


class Plan:
    """Structure analysis of one river network, bound to one GPU.  Built from the CSC adjacency the reference's
    routers hold (river_route/routers/Muskingum.py:189-191)."""

    def __init__(self, csc_indptr, csc_indices, device: int = 0):
        indptr = np.ascontiguousarray(csc_indptr, dtype=np.int32)
        indices = np.ascontiguousarray(csc_indices, dtype=np.int32)
        if indptr.ndim != 1 or indptr.shape[0] < 1:
            raise ValueError('csc_indptr must be a 1-D array of length n + 1')
        self.n = int(indptr.shape[0] - 1)
        if indices.shape[0] != int(indptr[-1]):
            raise ValueError('csc_indices length does not match csc_indptr[-1]')
        self.device = int(device)
        self._h = C.c_void_p()
        check(_lib.lib().rr_plan_create(self.n, ptr(indptr), ptr(indices) if indices.size else None,
                                        self.device, C.byref(self._h)))
        self._indptr, self._indices = indptr, indices
        info = np.zeros(8, dtype=np.int64)
        check(_lib.lib().rr_plan_info(self._h, ptr(info)))
        self.n_edges, self.depth, self.widest_level, self.n_headwaters, self.n_outlets = (int(v) for v in info[1:6])
        self.identity_order = bool(info[6])
        self.n_inner = self.n - self.n_headwaters

    # -- lifetime --
    def close(self) -> None:
        if getattr(self, '_h', None) is not None and self._h.value:
            _lib.lib().rr_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- inspection --
    def layout(self):
        """(perm, lag, child_ptr) of the engine order, see rr_plan_layout."""
        perm = np.empty(self.n, dtype=np.int32)
        lag = np.empty(self.n, dtype=np.int32)
        child_ptr = np.empty(self.n + 1, dtype=np.int32)
        check(_lib.lib().rr_plan_layout(self._h, ptr(perm), ptr(lag), ptr(child_ptr)))
        return perm, lag, child_ptr

    def tile_info(self) -> dict:
        """Shape of the time-tiled kernel's layout, see rr_plan_tile_info."""
        info = np.zeros(8, dtype=np.int64)
        check(_lib.lib().rr_plan_tile_info(self._h, ptr(info)))
        return dict(ok=bool(info[0]), block=int(info[1]), positions=int(info[2]), ghosts=int(info[3]), tiles=int(info[4]),
                    levels=int(info[5]), threads=int(info[6]), batch_rows=int(info[7]))

    def tile_layout(self) -> dict:
        """Arrays of the subtree-tile layout, see rr_plan_tile_layout."""
        t = self.tile_info()
        nt, npos = t['tiles'], t['positions']
        out = dict(tile_ptr=np.empty(nt + 1, np.int32), tile_level=np.empty(nt, np.int32), perm=np.empty(npos, np.int32),
                   lag=np.empty(npos, np.int32), cfirst=np.empty(npos, np.int32), ccnt=np.empty(npos, np.uint32),
                   xpos=np.empty(npos, np.int32))
        check(_lib.lib().rr_plan_tile_layout(self._h, *(ptr(out[k]) for k in ('tile_ptr', 'tile_level', 'perm', 'lag', 'cfirst',
                                                                               'ccnt', 'xpos'))))
        return out

    def set_options(self, rows_per_chunk: int = 0, sample_every: int = -1) -> None:
        check(_lib.lib().rr_plan_set_options(self._h, int(rows_per_chunk), int(sample_every)))

    def set_row_format(self, in32_big_endian: bool = False, out32_big_endian: bool = False) -> None:
        """rr_plan_set_row_format: the float32 rows of the calls to come are big-endian (a NetCDF-3 file's bytes as they are)."""
        check(_lib.lib().rr_plan_set_row_format(self._h, int(bool(in32_big_endian)), int(bool(out32_big_endian))))

    def profile(self) -> dict:
        p = np.zeros(10, dtype=np.float64)
        check(_lib.lib().rr_plan_profile(self._h, ptr(p)))
        return dict(launches=int(p[0]), sampled=int(p[1]), sampled_ms=float(p[2]), min_ms=float(p[3]),
                    max_ms=float(p[4]), sampled_reaches=float(p[5]), region_ms=float(p[6]), reach_steps=float(p[7]),
                    brackets=int(p[8]), ticks_per_launch=int(p[9]))

    def profile_aux(self) -> dict:
        """rr_plan_profile_aux: the kernels around the routing kernel in the last call, {name: {launches, sampled, sampled_ms}}."""
        a = np.zeros(12, dtype=np.float64)
        check(_lib.lib().rr_plan_profile_aux(self._h, ptr(a)))
        names = ('k_rec_in', 'k_rec_out', 'k_tile (skeleton)', 'k_rec_out (holes)')
        return {nm: dict(launches=int(a[3 * k]), sampled=int(a[3 * k + 1]), sampled_ms=float(a[3 * k + 2])) for k, nm in enumerate(names) if a[3 * k] > 0}

    # -- coefficients --
    def set_coeffs(self, lhs_off_data, c2, c3, c4_dt=None) -> None:
        lhs = _f64(lhs_off_data, 'lhs_off_data')
        c2, c3 = _f64(c2, 'c2'), _f64(c3, 'c3')
        if lhs.shape != (self.n_edges,) or c2.shape != (self.n,) or c3.shape != (self.n,):
            raise ValueError('coefficient arrays do not match the plan (lhs_off_data per CSC entry, c2/c3 per reach)')
        c4 = None
        if c4_dt is not None:
            c4 = _f64(c4_dt, 'c4_dt')
            if c4.shape != (self.n,):
                raise ValueError('c4_dt must have one value per reach')
        check(_lib.lib().rr_plan_set_coeffs(self._h, ptr(lhs) if lhs.size else None, ptr(c2), ptr(c3), ptr(c4)))

    def set_unit_weights(self, c1=None, a_data=None) -> None:
        """rr_plan_set_unit_weights: general a_inner_data / a_hw_data (per CSC entry) and c1 per reach for unit_route; None, None
        goes back to the unit weights of the reference's callers."""
        if c1 is None and a_data is None:
            check(_lib.lib().rr_plan_set_unit_weights(self._h, None, None))
            return
        c1, a = _f64(c1, 'c1'), _f64(a_data, 'a_data')
        if c1.shape != (self.n,) or a.shape != (self.n_edges,):
            raise ValueError('c1 must have one value per reach and a_data one per CSC entry')
        check(_lib.lib().rr_plan_set_unit_weights(self._h, ptr(c1), ptr(a) if a.size else None))

    # -- host-array routing: the reference's kernel boundary --
    def rapid_route(self, q_t, qlateral, discharge_array, num_substeps: int) -> None:
        ql = np.ascontiguousarray(qlateral, dtype=np.float64)
        if ql.ndim != 2 or ql.shape[1] != self.n:
            raise ValueError(f'qlateral must have shape (T, {self.n})')
        T = ql.shape[0]
        _inplace_f64(q_t, 'q_t', (self.n,))
        _inplace_f64(discharge_array, 'discharge_array', (T, self.n))
        check(_lib.lib().rr_rapid_route(self._h, ptr(q_t), ptr(ql), ptr(discharge_array), T, int(num_substeps)))

    def muskingum_route(self, q_t, discharge_array, num_output_steps: int, num_routing_per_output: int) -> None:
        _inplace_f64(q_t, 'q_t', (self.n,))
        _inplace_f64(discharge_array, 'discharge_array', (int(num_output_steps), self.n))
        check(_lib.lib().rr_muskingum_route(self._h, ptr(q_t), ptr(discharge_array), int(num_output_steps),
                                            int(num_routing_per_output)))

    def unit_route(self, q_ch, q_full, convolved_lateral, discharge_array, num_substeps: int) -> None:
        conv = np.ascontiguousarray(convolved_lateral, dtype=np.float64)
        if conv.ndim != 2 or conv.shape[1] != self.n:
            raise ValueError(f'convolved_lateral must have shape (T, {self.n})')
        T = conv.shape[0]
        _inplace_f64(q_ch, 'q_ch', (self.n_inner,))
        _inplace_f64(q_full, 'q_full', (self.n_inner,))
        _inplace_f64(discharge_array, 'discharge_array', (T, self.n))
        check(_lib.lib().rr_unit_route(self._h, ptr(q_ch), ptr(q_full), ptr(conv), ptr(discharge_array), T,
                                       int(num_substeps)))

    # -- work memory --
    def reserve(self, mode: int, T: int, num_substeps: int = 1, host_rows: bool = False, plain_rows: bool = True, f32_out: bool = False, uh: bool = False) -> dict:
        """rr_plan_reserve: allocate what route calls of up to T rows x num_substeps sub-steps work in (the record ring, events,
        with host_rows the PCIe staging).  The *_dev entry points of the C ABI only enqueue and fail with RR_E_STATE when this
        has not been done; the methods below call it for the shape they are given (no-op once large enough).  plain_rows=False:
        the call hands over no lateral rows in a device array (fused convolution, gridded runoff), so the direct row path does not
        apply (RR_ROWS_NOT_PLAIN); f32_out: it writes float32 rows (RR_ROWS_F32_OUT); uh: it is unit_route_uh*_dev (RR_ROWS_UH: on the
        direct row path the convolved rows are work memory)."""
        info = np.zeros(8, dtype=np.int64)
        check(_lib.lib().rr_plan_reserve(self._h, int(mode), int(T), int(num_substeps), int(bool(host_rows)) | (0 if plain_rows else 2) | (4 if f32_out else 0) | (8 if uh else 0), ptr(info)))
        if int(info[0]) == 0 and int(T) * int(num_substeps) >= 32 and self.n > 0 and not getattr(self, '_warned_streaming', False):
            # the time-tiled kernel takes every call of 32 sub-steps or more -- unless the network does not tile, the coefficients give the
            # tributaries of a reach different weights, or its record ring (depth + tile levels x K tick-rows of every reach) does not fit
            # the card: the streaming kernel then routes at about a third of the rate.  Say so once, where the routers' log shows it.
            import logging
            self._warned_streaming = True
            logging.getLogger('river_route_amd').warning(
                'routing %d reaches (network depth %d) with the streaming kernel k_tick: the time-tiled kernel does not apply to this call '
                '(record ring too large for the device, per-edge weights, or a reach with more tributaries than a tile holds); '
                'expect about a third of its rate', self.n, self.depth)
        return dict(tiled=int(info[0]) == 1, direct=int(info[0]) == 2, ticks_per_launch=int(info[1]), ring_chunks=int(info[2]), work_bytes=int(info[3]),
                    staging_bytes=int(info[4]), pinned_bytes=int(info[5]), pipeline_ticks=int(info[6]), ring_bytes=int(info[7]))

    def direct_info(self) -> dict:
        """Whether the direct row path applies to this plan (the params order numbers small subtrees contiguously) and its shape,
        see rr_plan_direct_info."""
        info = np.zeros(8, dtype=np.int64)
        why = C.create_string_buffer(256)
        check(_lib.lib().rr_plan_direct_info(self._h, ptr(info), C.cast(why, C.c_void_p), 256))
        return dict(ok=bool(info[0]), tiles=int(info[1]), holes=int(info[2]), outlets=int(info[3]), skeleton_positions=int(info[4]),
                    skeleton_tiles=int(info[5]), skeleton_levels=int(info[6]), window_rows=int(info[7]), why=why.value.decode())

    def direct_layout(self) -> dict:
        """Arrays of the direct row path's layout, see rr_plan_direct_layout."""
        nt = self.direct_info()['tiles']
        out = dict(tile_c0=np.empty(nt, np.int32), tile_nc=np.empty(nt, np.int32), tile_lag_lo=np.empty(nt, np.int32), tile_span=np.empty(nt, np.int32),
                   delay=np.empty(self.n, np.int32), up3=np.empty(self.n, np.int32), xinfo=np.empty(self.n, np.int32))
        check(_lib.lib().rr_plan_direct_layout(self._h, *(ptr(out[k]) for k in ('tile_c0', 'tile_nc', 'tile_lag_lo', 'tile_span', 'delay', 'up3', 'xinfo'))))
        return out

    def last_kernel(self) -> str:
        """'tick' (streaming), 'tile' (time-tiled over records) or 'direct' (direct row path): what the last call ran."""
        return ('tick', 'tile', 'direct')[int(_lib.lib().rr_plan_last_kernel(self._h))]

    # -- device-pointer routing (enqueue only) --
    def rapid_route_dev(self, q_t, qlateral, ql_rows, discharge, out_rows, T, num_substeps, stream=None) -> None:
        self.reserve(MODE_RAPID, T, num_substeps)
        check(_lib.lib().rr_rapid_route_dev(self._h, ptr(q_t), ptr(qlateral), int(ql_rows), ptr(discharge),
                                            int(out_rows), int(T), int(num_substeps), stream))

    def muskingum_route_dev(self, q_t, discharge, out_rows, num_output_steps, num_routing_per_output,
                            stream=None) -> None:
        self.reserve(MODE_MUSKINGUM, num_output_steps, num_routing_per_output)
        check(_lib.lib().rr_muskingum_route_dev(self._h, ptr(q_t), ptr(discharge), int(out_rows),
                                                int(num_output_steps), int(num_routing_per_output), stream))

    def unit_route_dev(self, q_ch, q_full, convolved, conv_rows, discharge, out_rows, T, num_substeps,
                       stream=None) -> None:
        self.reserve(MODE_UNIT, T, num_substeps)
        check(_lib.lib().rr_unit_route_dev(self._h, ptr(q_ch), ptr(q_full), ptr(convolved), int(conv_rows),
                                           ptr(discharge), int(out_rows), int(T), int(num_substeps), stream))


    # -- device-pointer routing with the routers' post-processing fused in: float32 rows, `factor` routed rows averaged --
    def rapid_route_f32_dev(self, q_t, qlateral, ql_rows, discharge32, T, num_substeps, factor=1, stream=None) -> None:
        self.reserve(MODE_RAPID, T, num_substeps, f32_out=True)
        check(_lib.lib().rr_rapid_route_f32_dev(self._h, ptr(q_t), ptr(qlateral), int(ql_rows), ptr(discharge32), int(T),
                                                int(num_substeps), int(factor), stream))

    def rapid_route_f32in_dev(self, q_t, qlateral32, ql_rows, T, num_substeps, discharge=None, out_rows=0, discharge32=None, factor=1,
                              stream=None) -> None:
        """rr_rapid_route_f32in_dev: float32 lateral rows in (exact in float64); exactly one of discharge / discharge32."""
        self.reserve(MODE_RAPID, T, num_substeps, f32_out=discharge32 is not None)
        check(_lib.lib().rr_rapid_route_f32in_dev(self._h, ptr(q_t), ptr(qlateral32), int(ql_rows), ptr(discharge), int(out_rows),
                                                  ptr(discharge32), int(factor), int(T), int(num_substeps), stream))

    def muskingum_route_f32_dev(self, q_t, discharge32, num_output_steps, num_routing_per_output, stream=None) -> None:
        self.reserve(MODE_MUSKINGUM, num_output_steps, num_routing_per_output, f32_out=True)
        check(_lib.lib().rr_muskingum_route_f32_dev(self._h, ptr(q_t), ptr(discharge32), int(num_output_steps),
                                                    int(num_routing_per_output), stream))

    def unit_route_f32_dev(self, q_ch, q_full, convolved, conv_rows, discharge32, T, num_substeps, factor=1, stream=None) -> None:
        self.reserve(MODE_UNIT, T, num_substeps, f32_out=True)
        check(_lib.lib().rr_unit_route_f32_dev(self._h, ptr(q_ch), ptr(q_full), ptr(convolved), int(conv_rows),
                                               ptr(discharge32), int(T), int(num_substeps), int(factor), stream))

    def rapid_route_runoff_dev(self, q_t, n_points, indptr, indices, weights, runoff, runoff_is_f32, stride_t, stride_p, area, flags, T,
                               discharge=None, discharge32=None, factor=1, stream=None) -> None:
        """Gridded runoff -> records -> routing in one call (rr_rapid_route_runoff_dev); exactly one of discharge / discharge32."""
        self.reserve(MODE_RAPID, T, 1, plain_rows=False)
        check(_lib.lib().rr_rapid_route_runoff_dev(self._h, ptr(q_t), int(n_points), ptr(indptr), ptr(indices), ptr(weights), ptr(runoff),
                                                   int(bool(runoff_is_f32)), int(stride_t), int(stride_p), ptr(area), int(flags),
                                                   ptr(discharge), ptr(discharge32), int(factor), int(T), stream))

    def unit_route_uh_dev(self, q_ch, q_full, q_final, uh_kernel, uh_state, n_ks, depth, T, num_substeps, discharge=None,
                          discharge32=None, factor=1, stream=None) -> None:
        """Convolution + routing of one file in one call (rr_unit_route_uh_dev); exactly one of discharge / discharge32."""
        self.reserve(MODE_UNIT, T, num_substeps, f32_out=discharge32 is not None, uh=True)
        check(_lib.lib().rr_unit_route_uh_dev(self._h, ptr(q_ch), ptr(q_full), ptr(q_final), ptr(uh_kernel), ptr(uh_state),
                                              int(n_ks), ptr(depth), ptr(discharge), ptr(discharge32), int(factor), int(T),
                                              int(num_substeps), stream))

    def unit_route_uh_f32in_dev(self, q_ch, q_full, q_final, uh_kernel, uh_state, n_ks, depth32, T, num_substeps, discharge=None,
                                discharge32=None, factor=1, stream=None) -> None:
        """rr_unit_route_uh_f32in_dev: the same from float32 runoff depths (as runoff files store them)."""
        self.reserve(MODE_UNIT, T, num_substeps, f32_out=discharge32 is not None, uh=True)
        check(_lib.lib().rr_unit_route_uh_f32in_dev(self._h, ptr(q_ch), ptr(q_full), ptr(q_final), ptr(uh_kernel), ptr(uh_state),
                                                    int(n_ks), ptr(depth32), ptr(discharge), ptr(discharge32), int(factor), int(T),
                                                    int(num_substeps), stream))

    # -- partitioned networks: boundary reaches + streaming calls (include/rr_hip.h) --
    def set_boundary(self, ghost_reaches, export_reaches) -> None:
        g = np.ascontiguousarray(ghost_reaches, dtype=np.int64)
        e = np.ascontiguousarray(export_reaches, dtype=np.int64)
        check(_lib.lib().rr_plan_set_boundary(self._h, g.size, ptr(g) if g.size else None, e.size,
                                              ptr(e) if e.size else None))
        self.n_ghost, self.n_export = int(g.size), int(e.size)

    def stream_begin(self, q_t, lateral, lat_rows, discharge, out_rows, T, num_substeps, ghost_series=None,
                     export_series=None, stream=None) -> None:
        # rings shorter than 32 rows keep to records (include/rr_hip.h, rr_stream_begin): reserve for the schedule the call will get
        plain = (lateral is None or min(int(lat_rows), int(T)) >= min(32, int(T))) and min(int(out_rows), int(T)) >= min(32, int(T))
        self.reserve(MODE_MUSKINGUM if lateral is None else MODE_RAPID, T, num_substeps, plain_rows=plain)
        check(_lib.lib().rr_stream_begin(self._h, 0 if lateral is None else 1, ptr(q_t), ptr(lateral), int(lat_rows),
                                         ptr(discharge), int(out_rows), int(T), int(num_substeps), ptr(ghost_series),
                                         ptr(export_series), stream))

    def stream_advance(self, lateral_rows_ready: int, ghost_substeps_ready: int) -> int:
        ready = C.c_int64(0)
        check(_lib.lib().rr_stream_advance(self._h, int(lateral_rows_ready), int(ghost_substeps_ready), C.byref(ready)))
        return int(ready.value)

    def stream_end(self, q_t=None) -> None:
        check(_lib.lib().rr_stream_end(self._h, ptr(q_t)))

    def stream_begin_unit(self, q_ch, q_full, lateral, lat_rows, discharge, out_rows, T, num_substeps, ghost_series=None,
                          export_series=None, stream=None) -> None:
        self.reserve(MODE_UNIT, T, num_substeps, plain_rows=False)
        check(_lib.lib().rr_stream_begin_unit(self._h, ptr(q_ch), ptr(q_full), ptr(lateral), int(lat_rows), ptr(discharge),
                                              int(out_rows), int(T), int(num_substeps), ptr(ghost_series), ptr(export_series), stream))

    def stream_end_unit(self, q_ch=None, q_full=None) -> None:
        check(_lib.lib().rr_stream_end_unit(self._h, ptr(q_ch), ptr(q_full)))


def partition_forest(csc_indptr, csc_indices, n_parts: int):
    """(part_of int32[n], part_sizes int64[n_parts]) -- rr_partition_forest; host-only."""
    indptr = np.ascontiguousarray(csc_indptr, dtype=np.int32)
    indices = np.ascontiguousarray(csc_indices, dtype=np.int32)
    n = indptr.shape[0] - 1
    part_of = np.empty(n, dtype=np.int32)
    sizes = np.zeros(n_parts, dtype=np.int64)
    check(_lib.lib().rr_partition_forest(n, ptr(indptr), ptr(indices) if indices.size else None, int(n_parts),
                                         ptr(part_of), ptr(sizes)))
    return part_of, sizes


def uh_convolve(kernel, state, lateral, device: int = 0) -> np.ndarray:
    """UnitHydrograph.convolve (river_route/uhkernels/UnitHydrograph.py:77-107) on the GPU.
    kernel (n_ks, n); state (n_ks, n) updated in place; lateral (T, n) -> (T, n)."""
    k = np.ascontiguousarray(kernel, dtype=np.float64)
    lat = np.ascontiguousarray(lateral, dtype=np.float64)
    n_ks, n = k.shape
    _inplace_f64(state, 'state', (n_ks, n))
    if lat.ndim != 2 or lat.shape[1] != n:
        raise ValueError(f'lateral must have shape (T, {n})')
    out = np.empty_like(lat)
    check(_lib.lib().rr_uh_convolve(int(device), ptr(k), ptr(state), ptr(lat), ptr(out), lat.shape[0], n_ks, n))
    return out


def uh_convolve_dev(kernel, state, lateral, out, T, n_ks, n, device: int = 0, stream=None) -> None:
    check(_lib.lib().rr_uh_convolve_dev(int(device), ptr(kernel), ptr(state), ptr(lateral), ptr(out), int(T),
                                        int(n_ks), int(n), stream))


RUNOFF_CUMULATIVE, RUNOFF_FORCE_POSITIVE, RUNOFF_KEEP_NAN = 1, 2, 4


def runoff_to_qlateral(indptr, indices, weights, runoff_tp, area=None, flags: int = 0, device: int = 0) -> np.ndarray:
    """rr_runoff_to_qlateral: (T, n_rivers) float64 from a CSR weight matrix (n_rivers x n_points, float64 data,
    int32 structure) and a (T, n_points) float32/float64 runoff block (river_route/runoff.py:288-330).  The block is
    handed over point-major so every gathered grid point is one contiguous run of time steps."""
    indptr = np.ascontiguousarray(indptr, dtype=np.int32)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    weights = np.ascontiguousarray(weights, dtype=np.float64)
    runoff_tp = np.asarray(runoff_tp)
    if runoff_tp.ndim != 2:
        raise ValueError('runoff must be (time, points)')
    if runoff_tp.dtype not in (np.float32, np.float64):
        runoff_tp = runoff_tp.astype(np.float64)
    T, n_points = runoff_tp.shape
    n_rivers = indptr.shape[0] - 1
    if indices.shape[0] and (indices.min() < 0 or indices.max() >= n_points):
        raise ValueError('weight matrix refers to grid points outside the runoff block')
    t_pad = -(-T // 16) * 16                                  # rows padded to whole 16-step chunks: vector gathers
    point_major = np.zeros((n_points, t_pad), dtype=runoff_tp.dtype)
    point_major[:, :T] = runoff_tp.T                          # (n_points, t_pad): stride_t = 1, stride_p = t_pad
    out = np.empty((T, n_rivers), dtype=np.float64)
    if area is not None:
        area = np.ascontiguousarray(area, dtype=np.float64)
        if area.shape != (n_rivers,):
            raise ValueError('area must have one value per river')
    check(_lib.lib().rr_runoff_to_qlateral(int(device), n_rivers, n_points, T, ptr(indptr), ptr(indices), ptr(weights),
                                           ptr(point_major), int(point_major.dtype == np.float32), 1, t_pad,
                                           ptr(area) if area is not None else None, int(flags), ptr(out)))
    return out


def runoff_to_qlateral_dev(n_rivers, n_points, T, indptr, indices, weights, runoff, runoff_is_f32, stride_t, stride_p, area, flags, out,
                           device: int = 0, stream=None) -> None:
    """rr_runoff_to_qlateral_dev: device arrays in, (T, n_rivers) float64 device rows out; only enqueues."""
    check(_lib.lib().rr_runoff_to_qlateral_dev(int(device), int(n_rivers), int(n_points), int(T), ptr(indptr), ptr(indices), ptr(weights),
                                               ptr(runoff), int(bool(runoff_is_f32)), int(stride_t), int(stride_p),
                                               ptr(area) if area is not None else None, int(flags), ptr(out), stream))


def resample_cast_dev(discharge, num_rows, n, factor, out, device: int = 0, stream=None) -> None:
    """Device-side mean over `factor` rows + float32 cast (rr_resample_cast_dev)."""
    check(_lib.lib().rr_resample_cast_dev(int(device), ptr(discharge), int(num_rows), int(n), int(factor), ptr(out), stream))


class DeviceBuffer:
    """A raw hipMalloc'd buffer for callers that do not use torch."""

    def __init__(self, nbytes: int, device: int = 0):
        self.device, self.nbytes = int(device), int(nbytes)
        p = C.c_void_p()
        check(_lib.lib().rr_dev_malloc(self.device, self.nbytes, C.byref(p)))
        self.address = int(p.value or 0)

    def upload(self, array: np.ndarray, offset: int = 0) -> 'DeviceBuffer':
        a = np.ascontiguousarray(array)
        check(_lib.lib().rr_dev_upload(self.device, self.address + offset, ptr(a), a.nbytes))
        return self

    def download(self, dtype, shape, offset: int = 0) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        check(_lib.lib().rr_dev_download(self.device, ptr(out), self.address + offset, out.nbytes))
        return out

    def free(self) -> None:
        if self.address:
            _lib.lib().rr_dev_free(self.device, self.address)
            self.address = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def rows_upload(dst, dst_pitch: int, path, file_offset: int, file_pitch: int, row_bytes: int, n_rows: int, device: int = 0, stream=None) -> None:
    """rr_rows_upload: rows of a file straight into a device array (pinned staging, reader threads beside the copy engine)."""
    check(_lib.lib().rr_rows_upload(int(device), ptr(dst), int(dst_pitch), str(path).encode(), int(file_offset), int(file_pitch), int(row_bytes), int(n_rows), stream))


def rows_download(src, src_pitch: int, path, file_offset: int, file_pitch: int, row_bytes: int, n_rows: int, device: int = 0, stream=None) -> None:
    """rr_rows_download: device rows straight into (an existing region of) a file."""
    check(_lib.lib().rr_rows_download(int(device), ptr(src), int(src_pitch), str(path).encode(), int(file_offset), int(file_pitch), int(row_bytes), int(n_rows), stream))


def copy_bandwidth(device: int = 0, nbytes: int = 1 << 31, reps: int = 10) -> float:
    """Measured device copy rate in GB/s (read + write), see rr_copy_bandwidth."""
    out = C.c_double(0.0)
    check(_lib.lib().rr_copy_bandwidth(int(device), int(nbytes), int(reps), C.byref(out)))
    return float(out.value)


def synchronize(device: int = 0) -> None:
    check(_lib.lib().rr_dev_synchronize(int(device)))
