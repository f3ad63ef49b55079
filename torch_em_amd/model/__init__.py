"""U-Net family of the MI355X path (reference torch_em/model/__init__.py)."""
from .unet import AnisotropicUNet, UNet2d, UNet3d
