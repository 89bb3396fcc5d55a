from .unet import Unet
from .norm_unet import NormUnet, NormUnet3D
from .mwcnn import MWCNN

__all__ = ["Unet", "NormUnet", "NormUnet3D", "MWCNN"]
