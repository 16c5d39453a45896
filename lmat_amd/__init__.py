"""lmat_amd -- MI355X-native read-labeling engine for LMAT's classification hot path.

The product is the C-ABI shared library ``liblmat_hip.so`` (see include/lmat_hip.h)
built from lmat_amd/csrc (hand-written HIP for gfx950).  This package is only the
thin ctypes mirror of that ABI used by the tests and bench.py; it has no CPU
implementation and raises if the library is missing.
"""
from .capi import Engine, Ingest, LmatError, Params, Reads, Stream, load_library, READ_RESULT_DTYPE, CAND_DTYPE  # noqa: F401
