"""
MI355X-native Muskingum routing engine behind river-route's Router API.

    import river_route_amd as rr
    rr.RapidMuskingum('config.yaml').route()

mirrors `import river_route as rr` of the reference for the routing hot path (river_route/__init__.py:11-29):
Configs, Muskingum, RapidMuskingum, UnitMuskingum, uhkernels.UnitHydrograph, runoff.runoff_to_qlateral, tools.adjacency_matrix.  The compute
runs in hand-written HIP kernels (csrc/, C ABI in include/rr_hip.h); there is no CPU fallback.
"""
__version__ = '0.1.0'

from . import runoff, synth, tools, uhkernels  # noqa: E402
from .routers import Configs, Muskingum, RapidMuskingum, UnitMuskingum  # noqa: E402

__all__ = ['Configs', 'Muskingum', 'RapidMuskingum', 'UnitMuskingum', 'uhkernels', 'runoff', 'tools', 'synth', '__version__']
