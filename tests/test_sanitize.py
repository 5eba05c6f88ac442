"""The host C++ of the engine (rr_plan.cpp: network analysis, subtree tiles, direct row tiles, partitioner, post-order; the host half of
rr_engine.hip) under AddressSanitizer + UndefinedBehaviorSanitizer ("race detection / sanitizers"): the host-only
planner tests run again in a child python whose librr_hip is the sanitized build (_lib.build(sanitize=True) -> librr_hip_asan.so,
RR_LIB_PATH) with the sanitizer runtime preloaded; any report fails the test.  Device code is compiled as usual: no GPU sanitizer on
this pool, and none of these tests computes on a GPU."""
import os
import subprocess
import sys

import pytest

from river_route_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_ONLY = ['tests/test_planner_shapes.py', 'tests/test_tiles.py', 'tests/test_direct.py', 'tests/test_host.py', 'tests/test_config5.py']


def test_host_planners_under_address_and_ub_sanitizers(tmp_path):
    runtime = _lib.sanitizer_runtime()
    if runtime is None:
        pytest.skip('no AddressSanitizer runtime next to hipcc on this box')
    lib = _lib.build(sanitize=True)
    log = str(tmp_path / 'san')
    env = dict(os.environ, RR_LIB_PATH=lib, LD_PRELOAD=runtime,
               ASAN_OPTIONS=f'detect_leaks=0:abort_on_error=0:halt_on_error=0:log_path={log}:protect_shadow_gap=0',      # (python itself leaks by design)
               UBSAN_OPTIONS=f'print_stacktrace=1:halt_on_error=0:log_path={log}')
    env.pop('RR_DIRECT', None)
    res = subprocess.run([sys.executable, '-m', 'pytest', '-x', '-q', '-m', 'not gpu', '-p', 'no:cacheprovider', *HOST_ONLY], cwd=REPO, env=env,
                         capture_output=True, text=True, timeout=1500)
    reports = [f for f in os.listdir(tmp_path) if f.startswith('san')]
    text = '\n'.join(open(os.path.join(tmp_path, f)).read() for f in reports)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert 'passed' in res.stdout and ' failed' not in res.stdout
    assert not reports, f'sanitizer reports:\n{text[:6000]}'
    assert 'runtime error' not in res.stderr and 'AddressSanitizer' not in res.stderr, res.stderr[-3000:]


def test_the_sanitized_build_reports_a_planted_overflow():
    """The harness above is only worth something if a report would be seen: rr_postorder handed an output array that is too short must
    come back as a heap-buffer-overflow from rr_plan.cpp, in a child of its own."""
    runtime = _lib.sanitizer_runtime()
    if runtime is None:
        pytest.skip('no AddressSanitizer runtime next to hipcc on this box')
    lib = _lib.build(sanitize=True)
    code = ('import numpy as np\nfrom river_route_amd import _lib\nh = _lib.lib()\n'
            'down = np.array([1, 2, -1], dtype=np.int64); order = np.empty(1, dtype=np.int64)\n'
            'h.rr_postorder(3, down.ctypes.data, order.ctypes.data)\n')
    env = dict(os.environ, RR_LIB_PATH=lib, LD_PRELOAD=runtime, ASAN_OPTIONS='detect_leaks=0:protect_shadow_gap=0')
    res = subprocess.run([sys.executable, '-c', code], cwd=REPO, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and 'heap-buffer-overflow' in res.stderr and 'rr_plan.cpp' in res.stderr, res.stderr[-2000:]
