"""Enumerations shared with the reference (values are the reference's own).

BlockType   internal/types/block.go:20-36
FilterMode  internal/types/mode.go:13-27
ReducerFunc internal/reducer/reducer.go:25-45 (string names there; integer codes in the C ABI)
"""
import enum

import numpy as np


class BlockType(enum.IntEnum):
    Int64 = 1
    Int32 = 2
    Int16 = 3
    Int8 = 4
    Uint64 = 5
    Uint32 = 6
    Uint16 = 7
    Uint8 = 8
    Float64 = 9
    Float32 = 10
    Bool = 11
    Bytes = 12


class FilterMode(enum.IntEnum):
    Equal = 1
    NotEqual = 2
    Gt = 3
    Ge = 4
    Lt = 5
    Le = 6
    In = 7
    NotIn = 8
    Range = 9
    Regexp = 10   # bytes columns, value = pattern text (Go regexp syntax)
    True_ = 11
    False_ = 12


class ReducerFunc(enum.IntEnum):
    Sum = 0
    First = 1
    Last = 2
    Min = 3
    Max = 4
    Mean = 5
    Var = 6
    Std = 7
    Count = 8
    # join reducers (reducer.go:432-731): evaluated by Context.reduce_buckets_sequential only
    FirstJoin = 9
    LastJoin = 10
    MinJoin = 11
    MaxJoin = 12
    MeanJoin = 13
    VarJoin = 14
    StdJoin = 15


NP_DTYPE = {
    BlockType.Int64: np.int64, BlockType.Int32: np.int32, BlockType.Int16: np.int16, BlockType.Int8: np.int8,
    BlockType.Uint64: np.uint64, BlockType.Uint32: np.uint32, BlockType.Uint16: np.uint16, BlockType.Uint8: np.uint8,
    BlockType.Float64: np.float64, BlockType.Float32: np.float32,
}
FROM_NP = {np.dtype(v): k for k, v in NP_DTYPE.items()}


def value_bits(dtype, v):
    """Little-endian bit pattern of one value of the column type, zero-extended to 64 bit."""
    a = np.array([v], dtype=NP_DTYPE[BlockType(dtype)])
    return int.from_bytes(a.tobytes(), "little")
