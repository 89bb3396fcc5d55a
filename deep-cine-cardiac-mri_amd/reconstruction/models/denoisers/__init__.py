from .unet import Unet
from .norm_unet import NormUnet, NormUnet3D
from .mwcnn import MWCNN
from .kspace_net import KSpaceCNN

__all__ = ["Unet", "NormUnet", "NormUnet3D", "MWCNN", "KSpaceCNN"]
