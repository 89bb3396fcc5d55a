from .varnet import SensitivityModel, VarNet, VarNetBlock
from .cinenet import CineNet, CineNetBlock
from .xpdnet import XPDNet, XPDNetBlock
from .recurrent_varnet import VarNet_RNN
from .recurrent_cinenet import CineNet_RNN
from .recurrent_xpdnet import XPDNet_RNN

__all__ = ["SensitivityModel", "VarNet", "VarNetBlock", "CineNet", "CineNetBlock", "XPDNet", "XPDNetBlock",
           "VarNet_RNN", "CineNet_RNN", "XPDNet_RNN"]
