"""knoxdb_amd -- MI355X (gfx950) implementation of KnoxDB's pack-scan hot path.

The product is libknoxgpu.so (hand-written HIP kernels behind the C ABI in include/knoxgpu.h);
this package is the loader and the thin host plumbing used by tests and bench.py.  It never
falls back to a CPU implementation: without the built library importing the engine fails.
"""
from .types import BlockType, FilterMode, ReducerFunc  # noqa: F401
from ._lib import KnoxError, SO_PATH  # noqa: F401
from .engine import (And, Bits, Column, Context, Leaf, Or, agg_combine, agg_value, device_count,  # noqa: F401
                     fill_windows, hash_bytes, hash_value, regexp_match_value)
