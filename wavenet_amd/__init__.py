"""wavenet_amd: MI355X (gfx950) implementation of the Sardingfish/Wavenet training path.

The compute lives in wavenet_amd/lib/libwavenet_hip.so (hand-written HIP kernels behind the C
ABI of include/wavenet_hip.h).  This package holds the ctypes binding of that ABI and the thin
host-side logic above it; it has no CPU fallback and raises if the library is missing.
"""
from .capi import Context, WavenetError, lib_path, load_library  # noqa: F401
