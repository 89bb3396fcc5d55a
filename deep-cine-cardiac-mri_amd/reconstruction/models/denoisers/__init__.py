from .unet import Unet
from .norm_unet import NormUnet, NormUnet3D

__all__ = ["Unet", "NormUnet", "NormUnet3D"]
