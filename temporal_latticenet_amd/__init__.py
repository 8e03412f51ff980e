"""temporal_latticenet_amd — the permutohedral splat / lattice-conv / slice hot path of AIS-Bonn/temporal_latticenet
and its temporal fusion modules, written for MI355X (gfx950): hand-written HIP kernels behind a C ABI
(include/tln.h), driven from PyTorch-ROCm host code that keeps the reference's operator API.

    from temporal_latticenet_amd import Lattice, ModelParams, LNN_SEQ
    temporal_latticenet_amd.install_compat()   # makes `latticenet`, `latticenet_py`, `torch_scatter`, `seq_lattice` importable
"""
import os
import sys

__version__ = "0.1.0"


def install_compat():
    """puts the name shims (compat/) on sys.path so the reference's drivers import this implementation"""
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "compat")
    if d not in sys.path:
        sys.path.insert(0, d)
    return d


def __getattr__(name):
    if name in ("Lattice", "ModelParams", "HashTable"):
        from . import lattice
        return getattr(lattice, name)
    if name == "LNN_SEQ":
        from .models import LNN_SEQ
        return LNN_SEQ
    raise AttributeError(name)
