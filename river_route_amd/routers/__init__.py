from .config import Configs
from .muskingum import Muskingum
from .rapid import RapidMuskingum
from .transform import TransformMuskingum
from .unit import UnitMuskingum

__all__ = ['Configs', 'Muskingum', 'RapidMuskingum', 'UnitMuskingum', 'TransformMuskingum']
