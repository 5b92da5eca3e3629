"""motion324_amd -- MI355X-native (gfx950) implementation of Motion324's per-frame motion-prediction hot path.

Public surface (mirrors the reference's model package):
    Motion_Latent_Model   drop-in for model/Pcd_motion.py::Motion_Latent_Model
    GraphedForward        hipGraph capture / replay of the forward (one host call per clip)
    set_precision         force 'bf16' / 'fp32' kernels (default: follow torch.autocast)
"""
from .easydict import EasyDict
from .prepared import set_precision, compute_dtype
from .Pcd_motion import Motion_Latent_Model
from .graph import GraphedForward

__all__ = ["Motion_Latent_Model", "GraphedForward", "EasyDict", "set_precision", "compute_dtype"]
