"""latticeurbanwind_amd -- MI355X-native D3Q19 lattice-Boltzmann core for LatticeUrbanWind.

Layout: csrc/ (HIP kernels + C-ABI, builds libluw_core.so), capi.py (ctypes binding of include/luw_core.h),
lbm.py (host-side mirror of the reference's `LBM` class over the C-ABI), distributed.py (one-process-per-GPU
domain decomposition driver over torch.distributed) with layout.py (who owns what, who talks to whom), transports.py
(how the halo messages travel) and hip_domain.py (one domain on one GPU: solver, streams, halo buffers).
"""
from .capi import build, load, LuwError  # noqa: F401
from .lbm import LBM, LBMGroup  # noqa: F401

__all__ = ["build", "load", "LuwError", "LBM", "LBMGroup"]
