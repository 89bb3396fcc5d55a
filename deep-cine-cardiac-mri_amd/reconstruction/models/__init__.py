from .varnet import SensitivityModel, VarNet, VarNetBlock

__all__ = ["SensitivityModel", "VarNet", "VarNetBlock"]
