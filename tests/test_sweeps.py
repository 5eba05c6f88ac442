"""The randomised parity sweeps of profiles/microbench (round 5) as `-m gpu` tests, a few dozen cases each with fixed seeds: random small networks, modes, sub-steps, row types,
task and batch lengths against the oracle -- the partitioned one found the short-call ring bug of DESIGN.md section 3e that no hand-written case had.  Each sweep runs in one child
process (the scripts are stand-alone programs); the longer runs are logged in profiles/r05_direct_fuzz.txt."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, *args, ok='agree with the oracle'):
    env = dict(os.environ)
    for k in ('RR_WAVE', 'RR_WAVE_K', 'RR_TILE_BLOCK', 'RR_TILE_LEAN', 'RR_UH_PAIRS', 'RR_DIRECT'):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(REPO, 'profiles', 'microbench', script), *map(str, args)], capture_output=True, text=True, env=env, cwd=REPO, timeout=900)
    tail = '\n'.join((res.stdout + res.stderr).splitlines()[-15:])
    assert res.returncode == 0 and ok in res.stdout, tail


def test_sweep_direct_row_path():
    _run('direct_fuzz.py', 30, 7)


def test_sweep_record_path():
    _run('direct_fuzz.py', 30, 5, 'random')


def test_sweep_partitioned_networks():
    _run('parts_fuzz.py', 14, 11)


def test_sweep_unit_muskingum_with_convolution():
    _run('unit_fuzz.py', 14, 3)


def test_sweep_host_pointer_kernel_functions():
    _run('host_fuzz.py', 20, 17)
