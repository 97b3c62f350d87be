"""The device-memory arena (knoxdb_amd/csrc/knox_arena.h) under the operations a long-lived host performs: allocate, free,
trim, allocate again -- every column must keep holding ITS OWN data.  Round 2 met a ROCm 7.2 virtual-memory hazard on the
way (a second physical handle mapped at a virtual address that had been mapped and unmapped before keeps reaching the
FIRST handle's memory: silent corruption); the arena avoids it by never reusing a virtual address, and this
test makes the failure impossible to reintroduce silently.  Also: columns that fit no single placement class (the mixed
region), the degraded case "no memory of another class within the budget", and the parked verified scratch buffers."""
import os

import numpy as np
import pytest

import knoxdb_amd as K
from knoxdb_amd import And, Leaf, Or
from knoxdb_amd.types import BlockType, FilterMode, ReducerFunc as R

pytestmark = pytest.mark.gpu

GIB = 1 << 30


def checksum_ok(ctx, col, first, n):
    """kind 3 columns hold first + row: count and wrapping sum of `v >= first` must be n and the arithmetic series"""
    res = ctx.scan(And(Leaf(0, FilterMode.Ge, first)), [col], aggs=[(R.Sum, 0), (R.Min, 0), (R.Max, 0)])
    want = (n * first + n * (n - 1) // 2) % (1 << 64)
    return (res.count == n and int(res.value(0)[0]) % (1 << 64) == want and int(res.value(1)[0]) == first and int(res.value(2)[0]) == first + n - 1)


def test_alloc_free_trim_realloc_keeps_every_columns_data():
    with K.Context(0) as ctx:
        ctx.set_arena(40 * GIB)
        rows = [(1 << 27) + 12345, 3 << 26, (1 << 28) + 777, 1 << 20, (1 << 27) - 4096, 5 << 25]   # 1 GiB+, 1.5, 2 GiB+, 8 MiB, <1 GiB, 1.25
        live = {}

        def make(tag, n):
            first = 1_000_000_007 * (tag + 1)
            live[tag] = (ctx.generate(BlockType.Int64, 3, 0, n, lo=first, pack_rows=1 << 16), first, n)

        def check_all(where):
            for tag, (col, first, n) in live.items():
                assert checksum_ok(ctx, col, first, n), (where, tag)

        for t, n in enumerate(rows):
            make(t, n)
        check_all("fresh")
        for t in (1, 3, 4):                       # free a middle, a small and a sub-GiB column: their ranges go back to the free lists
            live.pop(t)[0].free()
        make(10, 3 << 26)                         # reuses freed ranges (same virtual addresses, same physical chunks: fine)
        make(11, (1 << 27) - 4096)
        check_all("after reuse")
        for t in (0, 2, 5, 10, 11):               # everything large goes; trim hands whole zones back to the driver
            live.pop(t)[0].free()
        freed = ctx.arena_trim()
        assert freed >= 4 * GIB
        for t, n in enumerate(rows):              # new zones: new reservations at addresses this process never used before
            make(20 + t, n)
        check_all("after trim")
        bits = ctx.bits(rows[2], for_col=live[22][0])   # a verified write buffer in the re-grown arena, and a scan through it
        r = ctx.scan(And(Leaf(0, FilterMode.Lt, live[22][1] + 1000)), [live[22][0]], bits_out=bits, pack_counts=True)
        assert r.count == 1000 and bits.count() == 1000 and bits.indexes().tolist() == list(range(1000))
        make(40, 1 << 27)                         # one more while the old ones live, then a last look at all of them
        check_all("final")
        info = ctx.arena_info()
        assert info["enabled"] and info["used_gib"] <= info["mapped_gib"] <= info["budget_gib"] and ctx.placement_info()["held_bytes"] == 0
        for col, _, _ in live.values():
            col.free()
        bits.free()


def test_interleaved_classes_fall_back_to_the_mixed_region():
    """KNOX_ARENA_FAKE_LABELS=1 labels new chunks class 0, 1, 0, 1 ... (a device whose classes interleave chunk by chunk):
    class-pure zones then strand half of the free chunks each.  A column that fits no single zone must still be placed --
    its chunks are taken out of the zones into one mixed-class region -- instead of KNOX_E_NOMEM below the budget."""
    os.environ["KNOX_ARENA_FAKE_LABELS"] = "1"
    try:
        with K.Context(0) as ctx:
            ctx.set_arena(13 * GIB)               # the heap's chunk + 12
            n = (1 << 29) - (1 << 17)             # 4 GiB - 1 MiB: four chunks
            cols = []
            for t in range(3):
                first = 77 + 1000 * t
                cols.append((ctx.generate(BlockType.Int64, 3, 0, n, lo=first, pack_rows=1 << 20), first))
            info = ctx.arena_info()
            assert info["mixed_zone_gib"] == 4.0 and info["mapped_gib"] <= 13.0
            for col, first in cols:
                assert checksum_ok(ctx, col, first, n)
            # scans into a verified bitset work on the mixed column like on any other
            bits = ctx.bits(n, for_col=cols[2][0])
            r = ctx.scan(And(Leaf(0, FilterMode.Range, cols[2][1] + 5, cols[2][1] + 14)), [cols[2][0]], bits_out=bits)
            assert r.count == 10 and bits.indexes().tolist() == list(range(5, 15))
            # a fourth one really does not fit: the budget is a hard cap
            with pytest.raises(K.KnoxError):
                ctx.generate(BlockType.Int64, 3, 0, n, lo=1, pack_rows=1 << 20)
            for col, _ in cols:
                col.free()
            bits.free()
    finally:
        del os.environ["KNOX_ARENA_FAKE_LABELS"]


def test_no_memory_of_another_class_within_the_budget_is_reported_not_hidden():
    """once the column is placed the budget is lowered to what is mapped + room for ONE candidate write buffer: the search
    cannot reach further (no second candidate, no more chunks), the allocation still succeeds with the best there is, and
    knox_bits_pair_check says what the pair measured -- a host sees a degraded placement instead of guessing"""
    with K.Context(0) as ctx:
        n = (1 << 28) - (1 << 17)                 # 2 GiB - 1 MiB: two chunks
        ctx.set_arena(8 * GIB)
        col = ctx.generate(BlockType.Int64, 3, 0, n, lo=5, pack_rows=1 << 20)
        mapped = int(ctx.arena_info()["mapped_gib"] * GIB)
        ctx.set_arena(mapped + (48 << 20))        # a 34 MiB candidate fits once
        bits = ctx.bits(n, for_col=col)
        ratio, tried = bits.pair_check
        assert tried >= 1 and 0.5 < ratio < 1.1, (ratio, tried)
        r = ctx.scan(And(Leaf(0, FilterMode.Le, 104)), [col], aggs=[(R.Sum, 0)], bits_out=bits)
        assert r.count == 100 and int(r.value(0)[0]) == sum(range(5, 105)) and bits.count() == 100
        assert ctx.arena_info()["mapped_gib"] * GIB <= mapped + (48 << 20)
        col.free()
        bits.free()


def test_alternating_multi_pass_scans_keep_their_verified_scratch():
    """multi-pass plans (mixed widths -> scratch bitsets) over two columns of >= 1 GiB in turn: the scratch verified for
    column A is parked while B is scanned and comes back for A without new probe measurements"""
    with K.Context(0) as ctx:
        ctx.set_arena(24 * GIB)
        n = 1 << 27
        a64 = ctx.generate(BlockType.Int64, 3, 0, n, lo=0, pack_rows=1 << 20)
        b64 = ctx.generate(BlockType.Int64, 3, 0, n, lo=10, pack_rows=1 << 20)
        a32 = ctx.generate(BlockType.Int32, 0, 9, n, lo=0, span=100, pack_rows=1 << 20)
        tree = Or(Leaf(0, FilterMode.Lt, 50), Leaf(1, FilterMode.Gt, 1000))   # int64 OR int32 leaf: mixed widths = multi-pass
        bits = ctx.bits(n)
        ra = ctx.scan(tree, [a64, a32], bits_out=bits).count
        rb = ctx.scan(tree, [b64, a32], bits_out=bits).count
        assert ra == 50 and rb == 40
        probes = ctx.arena_info()["probes"]
        for _ in range(3):
            assert ctx.scan(tree, [a64, a32], bits_out=bits).count == 50
            assert ctx.scan(tree, [b64, a32], bits_out=bits).count == 40
        assert ctx.arena_info()["probes"] == probes
        for o in (a64, b64, a32, bits):
            o.free()


def test_columns_scanned_together_share_a_placement_class():
    """knox_ctx_reserve + the arena's "previous column's class first" rule: the two columns of a two-leaf scan and the
    pointer array / pools of a bytes column land in ONE placement class (two read streams of one scan run 4-5 % faster
    there), the bitset verified against both in ANOTHER one when the device offers one"""
    with K.Context(0) as ctx:
        ctx.set_arena(48 * GIB)
        n = 3 << 27                                # 3 GiB per float64 column
        assert ctx.reserve(2 * (n * 8 + (1 << 20)))
        a = ctx.generate(BlockType.Float64, 1, 3, n, pack_rows=1 << 20)
        b = ctx.generate(BlockType.Float64, 2, 4, n, pack_rows=1 << 20)
        assert a.placement_class() >= 0 and a.placement_class() == b.placement_class()
        bits = ctx.bits(n, for_col=[a, b])
        r = ctx.scan(And(Leaf(0, FilterMode.Ge, 256.0), Leaf(1, FilterMode.Lt, 0.5)), [a, b], aggs=[(R.Count, 0)], bits_out=bits)
        assert r.count == bits.count() and abs(r.count / n - 0.375) < 1e-3
        c = ctx.generate(BlockType.Int64, 3, 0, 1 << 27, lo=0, pack_rows=1 << 20)     # the next column follows while there is room
        addr = ctx.generate_bytes(6, 1 << 26, 1 << 20, 32, 16, 4)                    # 0.5 GiB of pointers + 2 GiB of pools
        assert addr.placement_class(pool=True) >= 0
        assert addr.placement_class() in (-1, addr.placement_class(pool=True))        # a sub-GiB pointer array lives in the heap
        big = ctx.generate_bytes(7, 1 << 28, 1 << 20, 32, 16, 4)                     # 2 GiB of pointers + 8 GiB of pools: one class
        assert big.placement_class() == big.placement_class(pool=True) >= 0
        for o in (a, b, c, addr, big, bits):
            o.free()
