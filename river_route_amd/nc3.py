"""
Where the rows of a NetCDF-3 (classic / 64-bit offset) variable lie in the file, and a discharge file's header without its rows.

The routers' qlateral and discharge files (river_route/routers/TransformMuskingum.py:30-36, Muskingum.py:319-352;
docs/references/io-file-schema.md) hold one (time, river_id) variable that is the whole file.  Read through scipy's NetCDF-3 reader
(the backend of this image: netCDF4 / xarray are absent) it becomes a host array, then a byte-swapped copy (the format is big-endian),
then pageable transfers: 3.4 s and 24 GB of host memory for a 3 GB file at 1M reaches x 744 rows, 10 ms of which is routing
(profiles/r05_route_breakdown_before.txt).  With the variable's offset and row pitch known, `engine.rows_upload` / `rows_download` move the
file's bytes between the page cache and the device through pinned chunks, and the kernels that read and write the rows anyway swap
the bytes (`Plan.set_row_format`).  Only the classic format is laid out flat like this; HDF5-based files (netCDF-4) go through their
library, i.e. through `io.read_qlateral` / `io.write_discharge`.

Format: https://docs.unidata.org/nug/current/file_format_specifications.html (classic format spec): header = magic, numrecs, dim_list,
gatt_list, var_list; fixed-size variables in definition order, then the records, each the record variables' slices in definition order.
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass

import numpy as np

__all__ = ['RowBlock', 'locate_rows', 'read_vector', 'create_discharge_file']

_TYPES = {1: 'i1', 2: 'S1', 3: '>i2', 4: '>i4', 5: '>f4', 6: '>f8'}
_NC_DIMENSION, _NC_VARIABLE, _NC_ATTRIBUTE = 0x0A, 0x0B, 0x0C


@dataclass
class RowBlock:
    """rows x cols values of `dtype` in `path`: row r starts at offset + r * pitch."""
    path: str
    offset: int
    pitch: int
    rows: int
    cols: int
    dtype: np.dtype

    @property
    def row_bytes(self) -> int:
        return self.cols * self.dtype.itemsize

    @property
    def big_endian(self) -> bool:
        return self.dtype.byteorder == '>' or (self.dtype.byteorder == '=' and not np.little_endian)


class _Reader:
    def __init__(self, buf: bytes):
        self.b, self.p = buf, 0

    def u32(self) -> int:
        v = struct.unpack_from('>I', self.b, self.p)[0]
        self.p += 4
        return v

    def u64(self) -> int:
        v = struct.unpack_from('>Q', self.b, self.p)[0]
        self.p += 8
        return v

    def name(self) -> str:
        n = self.u32()
        s = self.b[self.p:self.p + n].decode('utf-8', 'replace')
        self.p += (n + 3) // 4 * 4
        return s

    def atts(self) -> dict:
        tag, count = self.u32(), self.u32()
        out = {}
        if tag == 0:
            return out
        if tag != _NC_ATTRIBUTE:
            raise ValueError('not a NetCDF-3 attribute list')
        for _ in range(count):
            key = self.name()
            t, n = self.u32(), self.u32()
            dt = np.dtype(_TYPES[t])
            raw = self.b[self.p:self.p + n * dt.itemsize]
            self.p += (n * dt.itemsize + 3) // 4 * 4
            out[key] = raw.decode('utf-8', 'replace').rstrip('\x00') if t == 2 else np.frombuffer(raw, dtype=dt)
        return out


def _header(path):
    """(version, numrecs, dims [(name, length)], variables {name: dict(dimids, atts, dtype, vsize, begin)}) or None if not NetCDF-3."""
    size = os.path.getsize(path)
    with open(path, 'rb') as f:
        head = f.read(min(size, 1 << 20))
    if len(head) < 8 or head[:3] != b'CDF' or head[3] not in (1, 2):
        return None
    version = head[3]
    for attempt in range(2):      # (a header longer than the first read: once more with the whole file's worth)
        try:
            r = _Reader(head)
            r.p = 4
            numrecs = r.u32()
            tag, count = r.u32(), r.u32()
            dims = []
            if tag == _NC_DIMENSION:
                for _ in range(count):
                    nm = r.name()
                    dims.append((nm, r.u32()))
            elif tag != 0:
                raise ValueError('not a NetCDF-3 dimension list')
            r.atts()
            tag, count = r.u32(), r.u32()
            variables = {}
            if tag == _NC_VARIABLE:
                for _ in range(count):
                    nm = r.name()
                    dimids = [r.u32() for _ in range(r.u32())]
                    atts = r.atts()
                    t = r.u32()
                    vsize = r.u32()
                    begin = r.u64() if version == 2 else r.u32()
                    variables[nm] = dict(dimids=dimids, atts=atts, dtype=np.dtype(_TYPES[t]), vsize=vsize, begin=begin)
            elif tag != 0:
                raise ValueError('not a NetCDF-3 variable list')
            return version, numrecs, dims, variables
        except struct.error:
            if attempt:
                raise ValueError(f'{path}: truncated NetCDF-3 header')
            with open(path, 'rb') as f:
                head = f.read(min(size, 1 << 26))
    return None


def _layout(path, var):
    h = _header(path)
    if h is None or var not in h[3]:
        return None
    version, numrecs, dims, variables = h
    if numrecs == 0xFFFFFFFF:
        return None      # streaming: the record count is not in the header
    v = variables[var]
    shape = [dims[d][1] for d in v['dimids']]
    is_rec = bool(v['dimids']) and dims[v['dimids'][0]][1] == 0
    rec_vars = [w for w in variables.values() if w['dimids'] and dims[w['dimids'][0]][1] == 0]
    item = v['dtype'].itemsize
    inner = int(np.prod(shape[1:], dtype=np.int64)) if len(shape) > 1 else 1
    if is_rec:
        recsize = inner * item if len(rec_vars) == 1 else sum((int(np.prod([dims[d][1] for d in w['dimids'][1:]], dtype=np.int64)) * w['dtype'].itemsize + 3) // 4 * 4 for w in rec_vars)
        return v, numrecs, inner, recsize
    rows = shape[0] if shape else 1
    return v, rows, inner, inner * item


def locate_rows(path, var: str):
    """RowBlock of a 2-D variable of a NetCDF-3 file (None: another format, or no such variable, or not 2-D)."""
    path = str(path)
    try:
        got = _layout(path, var)
    except (OSError, ValueError, KeyError):
        return None
    if got is None:
        return None
    v, rows, cols, pitch = got
    if len(v['dimids']) != 2:
        return None
    block = RowBlock(path, int(v['begin']), int(pitch), int(rows), int(cols), v['dtype'])
    if block.offset + (block.rows - 1) * block.pitch + block.row_bytes > os.path.getsize(path):
        return None
    return block


def read_vector(path, var: str):
    """(values in native byte order, attributes) of a 1-D variable, read straight from the file."""
    got = _layout(str(path), var)
    if got is None:
        raise KeyError(f'{var} not in {path}')
    v, rows, cols, pitch = got
    mm = np.memmap(str(path), dtype=np.uint8, mode='r')
    item = v['dtype'].itemsize
    raw = np.lib.stride_tricks.as_strided(mm[int(v['begin']):], shape=(rows, cols * item), strides=(pitch, 1))
    out = np.ascontiguousarray(raw).view(v['dtype']).reshape(rows * cols).astype(v['dtype'].newbyteorder('='))
    del raw, mm
    return out, v['atts']


def _name(s: str) -> bytes:
    b = s.encode()
    return struct.pack('>I', len(b)) + b + b'\x00' * (-len(b) % 4)


def _att(key: str, value) -> bytes:
    if isinstance(value, str):
        b = value.encode()
        return _name(key) + struct.pack('>II', 2, len(b)) + b + b'\x00' * (-len(b) % 4)
    a = np.atleast_1d(np.asarray(value))
    t = {'f4': 5, 'f8': 6, 'i4': 4, 'i2': 3, 'i1': 1}[a.dtype.newbyteorder('=').str[1:]]
    raw = a.astype(a.dtype.newbyteorder('>')).tobytes()
    return _name(key) + struct.pack('>II', t, a.size) + raw + b'\x00' * (-len(raw) % 4)


def _atts(d: dict) -> bytes:
    if not d:
        return struct.pack('>II', 0, 0)
    return struct.pack('>II', _NC_ATTRIBUTE, len(d)) + b''.join(_att(k, v) for k, v in d.items())


def create_discharge_file(path, dates, river_ids, var_river_id='river_id', var_discharge='Q', routed_file='', record_dim=None) -> RowBlock:
    """The discharge file of river_route/routers/Muskingum.py:319-352 (dims (time, river_id); time f8 'seconds since <first date>'; ids i4;
    Q f4 with its attributes; global attribute runoff_file) in the 64-bit-offset classic format WITHOUT the rows of Q: header, time and
    ids are written, the file is extended to its full size, and the returned RowBlock says where the (big-endian float32) rows go
    (engine.rows_download).  As io.write_discharge does: `time` is the record dimension once Q reaches 2 GiB."""
    dates = np.asarray(dates).astype('datetime64[s]')
    T, n = int(dates.shape[0]), int(np.asarray(river_ids).shape[0])
    origin = str(dates[0]).replace('T', ' ')
    seconds = (dates - dates[0]).astype('timedelta64[s]').astype(np.int64).astype('>f8')
    ids = np.asarray(river_ids).astype('>i4')
    big = T * n * 4 >= (1 << 31) - 4 if record_dim is None else bool(record_dim)
    q_atts = dict(long_name='Discharge at catchment outlet', standard_name='discharge', aggregation_method='mean', units='m3 s-1')

    def header(begin_time, begin_ids, begin_q):
        dims = struct.pack('>II', _NC_DIMENSION, 2) + _name('time') + struct.pack('>I', 0 if big else T) + _name(var_river_id) + struct.pack('>I', n)
        gatts = _atts({'runoff_file': str(routed_file)})
        def var(name, dimids, atts, nc_type, vsize, begin):
            return (_name(name) + struct.pack('>I', len(dimids)) + b''.join(struct.pack('>I', d) for d in dimids) + _atts(atts) +
                    struct.pack('>II', nc_type, min(vsize, 0xFFFFFFFF)) + struct.pack('>Q', begin))
        vs = (var('time', [0], {'units': f'seconds since {origin}'}, 6, 8 if big else T * 8, begin_time) +
              var(var_river_id, [1], {}, 4, n * 4, begin_ids) +
              var(var_discharge, [0, 1], q_atts, 5, n * 4 if big else T * n * 4, begin_q))
        return b'CDF\x02' + struct.pack('>I', T if big else 0) + dims + gatts + struct.pack('>II', _NC_VARIABLE, 3) + vs

    hlen = len(header(0, 0, 0))
    if big:      # fixed: ids; then the records: time value + Q row
        begin_ids = hlen
        begin_time = begin_ids + n * 4
        begin_q = begin_time + 8
        pitch = 8 + n * 4
        total = begin_time + T * pitch
    else:
        begin_time = hlen
        begin_ids = begin_time + T * 8
        begin_q = begin_ids + n * 4
        pitch = n * 4
        total = begin_q + T * pitch
    with open(path, 'wb') as f:
        f.write(header(begin_time, begin_ids, begin_q))
        f.truncate(total)
        f.seek(begin_ids)
        f.write(ids.tobytes())
        if not big:
            f.seek(begin_time)
            f.write(seconds.tobytes())
    if big:
        mm = np.memmap(str(path), dtype=np.uint8, mode='r+')
        np.lib.stride_tricks.as_strided(mm[begin_time:], shape=(T, 8), strides=(pitch, 1))[:] = seconds.view(np.uint8).reshape(T, 8)
        mm.flush()
        del mm
    return RowBlock(str(path), begin_q, pitch, T, n, np.dtype('>f4'))
