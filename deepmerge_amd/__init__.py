"""deepmerge_amd -- MI355X-native (gfx950) implementation of DeepMerge's pair-encoder hot path.

Layout
  csrc/                 hand-written HIP kernels + the C-ABI (include/deepmerge_hip.h)
  _lib.py, ops.py       ctypes binding and torch-facing wrappers / autograd Functions
  nets/ShfitScaleFormer.py, Losses.py
                        drop-in mirrors of the reference's Python modules for this path
The HIP library is required: importing the package is cheap, but any op raises
`DeepMergeLibraryError` if libdeepmerge_hip.so has not been built (no CPU fallback).
"""
from ._lib import DeepMergeLibraryError, lib  # noqa: F401
from .ops import get_numerics, set_numerics  # noqa: F401

__version__ = "0.1.0"
