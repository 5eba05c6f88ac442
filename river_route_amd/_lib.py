"""
ctypes binding of librr_hip.so (include/rr_hip.h).  There is no CPU fallback: if the shared object is
missing this module raises, and every compute entry point raises RRError(RR_E_NO_DEVICE) without a GPU.
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('RR_LIB_PATH') or os.path.join(_HERE, 'librr_hip.so')      # RR_LIB_PATH: a development build
CSRC = os.path.join(_HERE, 'csrc')
SOURCES = ('rr_plan.cpp', 'rr_engine.hip')

RR_OK = 0
RR_E_INVALID, RR_E_NOT_TOPOLOGICAL, RR_E_HIP, RR_E_NO_DEVICE, RR_E_STATE, RR_E_ALLOC, RR_E_UNSUPPORTED = \
    -1, -2, -3, -4, -5, -6, -7
RR_DEVICE_NONE = -1


class RRError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f'librr_hip error {code}: {message}')
        self.code = code
        self.message = message


SANITIZED_LIB_PATH = os.path.join(_HERE, 'librr_hip_asan.so')


def sanitizer_runtime() -> str | None:
    """The AddressSanitizer runtime of the compiler hipcc drives (to be preloaded into a python that loads the sanitized build)."""
    import glob
    found = sorted(glob.glob('/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so'))
    return found[-1] if found else None


def build(force: bool = False, verbose: bool = False, sanitize: bool = False) -> str:
    """Compile librr_hip.so for gfx950 with hipcc (cross-compiles without a GPU). Returns the path.
    sanitize: the HOST code (network analysis, tile / direct planners, partitioner, executor: rr_plan.cpp and the host half of
    rr_engine.hip) under AddressSanitizer + UndefinedBehaviorSanitizer into librr_hip_asan.so -- device code is compiled as usual
    (-fno-gpu-sanitize: no GPU sanitizer on this pool).  tests/test_sanitize.py runs the host-only planner tests against it."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith('.hpp')] + [os.path.join(os.path.dirname(_HERE), 'include', 'rr_hip.h')]
    target = SANITIZED_LIB_PATH if sanitize else LIB_PATH
    if not force and os.path.exists(target) and all(os.path.getmtime(target) >= os.path.getmtime(d) for d in deps):
        return target
    hipcc = os.environ.get('HIPCC') or ('/opt/rocm/bin/hipcc' if os.path.exists('/opt/rocm/bin/hipcc') else 'hipcc')
    opt = ['-O1', '-g', '-fsanitize=address,undefined', '-fno-gpu-sanitize', '-fno-omit-frame-pointer', '-shared-libsan'] if sanitize else ['-O3']
    cmd = [hipcc, '--offload-arch=gfx950', *opt, '-std=c++17', '-fPIC', '-shared', '-pthread', '-Wall', '-Wno-unused-result',
           *srcs, '-o', target]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return target


def _preload_hip_runtime() -> None:
    """PyTorch-ROCm wheels carry their own libamdhip64.so (same SONAME as /opt/rocm's).  Two HIP runtimes in
    one process do not share streams or allocations, so when torch is installed its copy is mapped first and
    librr_hip.so then binds to it by SONAME, whichever of the two modules is imported first."""
    try:
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), 'lib', 'libamdhip64.so')
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


_i64 = C.c_int64
_vp = C.c_void_p
_lib: C.CDLL | None = None

# name -> (restype, argtypes); pointers are void* so numpy arrays and raw device addresses both pass
_SIGNATURES = {
    'rr_version': (C.c_int, []),
    'rr_last_error': (C.c_char_p, []),
    'rr_device_count': (C.c_int, []),
    'rr_plan_create': (C.c_int, [_i64, _vp, _vp, C.c_int, C.POINTER(_vp)]),
    'rr_plan_destroy': (None, [_vp]),
    'rr_plan_info': (C.c_int, [_vp, _vp]),
    'rr_plan_layout': (C.c_int, [_vp, _vp, _vp, _vp]),
    'rr_plan_tile_info': (C.c_int, [_vp, _vp]),
    'rr_plan_tile_layout': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'rr_plan_direct_info': (C.c_int, [_vp, _vp, _vp, _i64]),
    'rr_plan_direct_layout': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'rr_plan_last_kernel': (C.c_int, [_vp]),
    'rr_plan_set_coeffs': (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    'rr_plan_set_unit_weights': (C.c_int, [_vp, _vp, _vp]),
    'rr_plan_reserve': (C.c_int, [_vp, C.c_int, _i64, _i64, C.c_int, _vp]),
    'rr_plan_set_options': (C.c_int, [_vp, _i64, _i64]),
    'rr_plan_set_row_format': (C.c_int, [_vp, C.c_int, C.c_int]),
    'rr_plan_profile': (C.c_int, [_vp, _vp]),
    'rr_plan_profile_aux': (C.c_int, [_vp, _vp]),
    'rr_rapid_route': (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64]),
    'rr_muskingum_route': (C.c_int, [_vp, _vp, _vp, _i64, _i64]),
    'rr_unit_route': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64]),
    'rr_uh_convolve': (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _i64, _i64, _i64]),
    'rr_rapid_route_dev': (C.c_int, [_vp, _vp, _vp, _i64, _vp, _i64, _i64, _i64, _vp]),
    'rr_muskingum_route_dev': (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp]),
    'rr_unit_route_dev': (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i64, _vp]),
    'rr_uh_convolve_dev': (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp]),
    'rr_rapid_route_runoff_dev': (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, C.c_int, _i64, _i64, _vp, C.c_int, _vp, _vp, _i64, _i64, _vp]),
    'rr_unit_route_uh_dev': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _vp]),
    'rr_unit_route_uh_f32in_dev': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _vp]),
    'rr_rapid_route_f32_dev': (C.c_int, [_vp, _vp, _vp, _i64, _vp, _i64, _i64, _i64, _vp]),
    'rr_rapid_route_f32in_dev': (C.c_int, [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _vp]),
    'rr_muskingum_route_f32_dev': (C.c_int, [_vp, _vp, _vp, _i64, _i64, _vp]),
    'rr_unit_route_f32_dev': (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i64, _vp]),
    'rr_plan_set_boundary': (C.c_int, [_vp, _i64, _vp, _i64, _vp]),
    'rr_stream_begin': (C.c_int, [_vp, C.c_int, _vp, _vp, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    'rr_stream_advance': (C.c_int, [_vp, _i64, _i64, C.POINTER(_i64)]),
    'rr_stream_end': (C.c_int, [_vp, _vp]),
    'rr_stream_begin_unit': (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    'rr_stream_end_unit': (C.c_int, [_vp, _vp, _vp]),
    'rr_partition_forest': (C.c_int, [_i64, _vp, _vp, C.c_int32, _vp, _vp]),
    'rr_postorder': (C.c_int, [_i64, _vp, _vp]),
    'rr_resample_cast_dev': (C.c_int, [C.c_int, _vp, _i64, _i64, _i64, _vp, _vp]),
    'rr_runoff_to_qlateral': (C.c_int, [C.c_int, _i64, _i64, _i64, _vp, _vp, _vp, _vp, C.c_int, _i64, _i64, _vp, C.c_int, _vp]),
    'rr_runoff_to_qlateral_dev': (C.c_int, [C.c_int, _i64, _i64, _i64, _vp, _vp, _vp, _vp, C.c_int, _i64, _i64, _vp,
                                            C.c_int, _vp, _vp]),
    'rr_dev_malloc': (C.c_int, [C.c_int, _i64, C.POINTER(_vp)]),
    'rr_dev_free': (C.c_int, [C.c_int, _vp]),
    'rr_dev_upload': (C.c_int, [C.c_int, _vp, _vp, _i64]),
    'rr_dev_download': (C.c_int, [C.c_int, _vp, _vp, _i64]),
    'rr_dev_synchronize': (C.c_int, [C.c_int]),
    'rr_copy_bandwidth': (C.c_int, [C.c_int, _i64, C.c_int, C.POINTER(C.c_double)]),
    'rr_rows_upload': (C.c_int, [C.c_int, _vp, _i64, C.c_char_p, _i64, _i64, _i64, _i64, _vp]),
    'rr_rows_download': (C.c_int, [C.c_int, _vp, _i64, C.c_char_p, _i64, _i64, _i64, _i64, _vp]),
}
EXPORTS = tuple(_SIGNATURES)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} is missing: the HIP engine is not built and there is no CPU fallback. '
                'Run `python -c "import __graft_entry__ as g; g.build()"` (needs hipcc) first.')
        _preload_hip_runtime()
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc: int) -> None:
    if rc != RR_OK:
        raise RRError(rc, lib().rr_last_error().decode('utf-8', 'replace'))


def device_count() -> int:
    return int(lib().rr_device_count())


def ptr(a) -> int | None:
    """void* of a C-contiguous numpy array, a raw integer device address, a torch tensor, or None."""
    if a is None:
        return None
    if isinstance(a, int):
        return a
    if isinstance(a, np.ndarray):
        if not a.flags['C_CONTIGUOUS']:
            raise ValueError('array must be C-contiguous')
        return a.ctypes.data
    if hasattr(a, 'address'):        # engine.DeviceBuffer
        return int(a.address)
    if hasattr(a, 'data_ptr'):
        if not a.is_contiguous():
            raise ValueError('tensor must be contiguous')
        return a.data_ptr()
    raise TypeError(f'cannot take the address of {type(a).__name__}')
