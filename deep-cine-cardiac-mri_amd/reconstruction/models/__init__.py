from .varnet import SensitivityModel, VarNet, VarNetBlock
from .cinenet import CineNet, CineNetBlock

__all__ = ["SensitivityModel", "VarNet", "VarNetBlock", "CineNet", "CineNetBlock"]
