"""lia_ral_amd -- MI355X-native GMM / i-vector compute engine behind LIA_RAL's hot-path call sites.

The product is the C-ABI shared library `csrc/libgmmiv.so` (hand-written HIP for gfx950, declared in
include/gmmiv.h).  This package is its Python host side: a ctypes binding (`capi`) and thin
mirrors of the reference's L3 driver functions (`host`).  There is no CPU fallback: importing
`capi` without the built library raises.
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
