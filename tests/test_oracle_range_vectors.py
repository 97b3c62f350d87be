"""Hand-worked cases for the oracle's restatement of the set / wide-integer MatchRangeVectors forms (the reference holds no
vectors for them: parity unpinned): internal/operator/filter/match_num.go:693-727,810-817, match_i128.go:78-375."""
import numpy as np

from oracle import oracle as O


def bits(x, n):
    return np.unpackbits(x, bitorder="little")[:n].tolist()


def test_set_range_vectors_follow_the_uint64_conversion():
    mins = np.array([-5, 10, 100, -3], dtype=np.int32)
    maxs = np.array([-1, 20, 200, 4], dtype=np.int32)
    # {15}: pack 1 = [10, 20]; pack 3 = (-3, 4): uint64(-3) > uint64(4), swapped to [4, 2^64 - 3] -- holds 15 (the reference's reading of a
    # pack that spans the sign change, match_num.go:704-708, restated literally)
    assert bits(O.range_vectors_set(O.I32, mins, maxs, [15]), 4) == [0, 1, 0, 1]
    # {-2}: pack 0 = [2^64 - 5, 2^64 - 1] holds 2^64 - 2; pack 3's swapped range ends at 2^64 - 3
    assert bits(O.range_vectors_set(O.I32, mins, maxs, [-2]), 4) == [1, 0, 0, 0]
    assert bits(O.range_vectors_set(O.I32, mins, maxs, [-4]), 4) == [1, 0, 0, 1]
    # masked rows only; NOT IN is undecided (mask or all ones); an empty set matches nothing
    mask = np.array([0b0110], dtype=np.uint8)
    assert bits(O.range_vectors_set(O.I32, mins, maxs, [15, -2], mask=mask), 4) == [0, 1, 0, 0]
    assert bits(O.range_vectors_set(O.I32, mins, maxs, [15], negate=True), 4) == [1, 1, 1, 1]
    assert bits(O.range_vectors_set(O.I32, mins, maxs, [15], negate=True, mask=mask), 4) == [0, 1, 1, 0]
    assert bits(O.range_vectors_set(O.I32, mins, maxs, []), 4) == [0, 0, 0, 0]
    # unsigned: zero extension, no swap ever
    assert bits(O.range_vectors_set(O.U8, np.array([0, 200], np.uint8), np.array([9, 255], np.uint8), [255, 9]), 2) == [1, 1]


def test_wide_range_vectors_compose_the_wide_compares():
    # Int128 as (hi int64, lo uint64): packs [ (0,5)..(0,9) ], [ (-1, 2^64-2)..(0,1) ], [ (1,0)..(1,7) ]
    mins = [np.array([0, -1, 1], dtype=np.int64), np.array([5, 2**64 - 2, 0], dtype=np.uint64)]
    maxs = [np.array([0, 0, 1], dtype=np.int64), np.array([9, 1, 7], dtype=np.uint64)]
    v = [0, 7]
    assert bits(O.range_vectors_wide(O.EQ, mins, maxs, v), 3) == [1, 0, 0]                    # 7 lies in pack 0 only (pack 1 ends at 1)
    assert bits(O.range_vectors_wide(O.EQ, mins, maxs, [0, 0]), 3) == [0, 1, 0]               # -2 <= 0 <= 1
    assert bits(O.range_vectors_wide(O.LT, mins, maxs, v), 3) == [1, 1, 0]                    # min < 7
    assert bits(O.range_vectors_wide(O.GE, mins, maxs, v), 3) == [1, 0, 1]                    # max >= 7
    assert bits(O.range_vectors_wide(O.RG, mins, maxs, [0, 1], [1, 0]), 3) == [1, 1, 1]       # [1, 2^64]: min <= to && max >= from
    assert bits(O.range_vectors_wide(O.RG, mins, maxs, [0, 10], [0, 20]), 3) == [0, 0, 0]
    mask = np.array([0b101], dtype=np.uint8)
    assert bits(O.range_vectors_wide(O.RG, mins, maxs, [0, 1], [1, 0], mask=mask), 3) == [1, 0, 1]
    assert bits(O.range_vectors_wide(O.NE, mins, maxs, v, mask=mask), 3) == [1, 0, 1]
    assert bits(O.range_vectors_wide(O.NIN, mins, maxs, v), 3) == [1, 1, 1]
