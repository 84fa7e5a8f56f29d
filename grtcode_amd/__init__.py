"""grtcode_amd -- MI355X-native line-by-line radiative transfer hot path behind GRTCODE's C ABI.

The product is the C-ABI shared library ``grtcode_amd/lib/libgrtcode_hip.so`` (C99 host
layer + hand-written gfx950 kernels; headers in ``include/``).  This package only
builds it (``grtcode_amd.build``) and mirrors its interface for Python callers
(``grtcode_amd.api``); there is no Python or CPU fallback for any computation.
"""
from .api import load_library, LibraryMissing  # noqa: F401
