import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, 'tests', 'golden')

# fp64 parity tolerance of the path: rtol 1e-10, atol 1e-10 * max|Q|
RTOL = 1e-10


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run on the GPU box via gpurun)')


def assert_close(got, want, what=''):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, f'{what}: shape {got.shape} != {want.shape}'
    scale = float(np.max(np.abs(want))) if want.size else 0.0
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=RTOL * max(scale, 1e-300), err_msg=what)


@pytest.fixture(scope='session')
def golden_kernels():
    return np.load(os.path.join(GOLDEN, 'kernels.npz'))


@pytest.fixture(scope='session')
def golden_routers():
    return np.load(os.path.join(GOLDEN, 'routers.npz'))


def unit_split(indptr, indices, n):
    """Headwater/inner split and inner sub-matrices exactly as UnitMuskingum._hook_before_route builds them
    (river_route/routers/UnitMuskingum.py:40-54), from the CSC adjacency."""
    import scipy.sparse
    A = scipy.sparse.csc_matrix((np.ones(len(indices)), indices, indptr), shape=(n, n))
    incoming = np.asarray(A.sum(axis=1)).flatten()
    hw_idx = np.where(incoming == 0)[0]
    inner_idx = np.where(incoming != 0)[0]
    A_in = A[np.ix_(inner_idx, inner_idx)].tocsc()
    A_hw = A[np.ix_(inner_idx, hw_idx)].tocsc()
    return hw_idx, inner_idx, A_in, A_hw


@pytest.fixture(autouse=True)
def _parquet_stand_in(monkeypatch):
    """The GPU box's image has pandas but no parquet engine (pyarrow/fastparquet).  Where that is the case the
    tests swap pandas' parquet reader/writer for pickle so the routers' state/params file handling still runs;
    the product code always calls read_parquet / to_parquet."""
    import pandas as pd
    try:
        import pyarrow  # noqa: F401
        return
    except ImportError:
        pass

    def read_parquet(path, columns=None, **kw):
        df = pd.read_pickle(path)
        return df[list(columns)] if columns is not None else df

    monkeypatch.setattr(pd, 'read_parquet', read_parquet)
    monkeypatch.setattr(pd.DataFrame, 'to_parquet', lambda self, path, **kw: self.to_pickle(path))
