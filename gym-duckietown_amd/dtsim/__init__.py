"""dtsim: MI355X-native batched Duckietown simulator (host side of libdtsim.so).

The HIP library is the product.  Importing this package does not need a GPU; creating a
BatchedSimulator does (dtsim_create fails with DTSIM_E_NOGPU otherwise -- there is no CPU
fallback).
"""
from . import _ffi, assets, maps  # noqa: F401
from ._ffi import DtsimError, DtsimLibraryError  # noqa: F401
from .batched import BatchedSimulator  # noqa: F401
from .vecenv import DuckietownVecEnv  # noqa: F401

__version__ = "0.1.0"
