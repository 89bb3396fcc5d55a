from .varnet import SensitivityModel, VarNet, VarNetBlock
from .cinenet import CineNet, CineNetBlock
from .xpdnet import XPDNet, XPDNetBlock

__all__ = ["SensitivityModel", "VarNet", "VarNetBlock", "CineNet", "CineNetBlock", "XPDNet", "XPDNetBlock"]
