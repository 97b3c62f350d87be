"""Host-side plumbing over the C ABI (include/knoxgpu.h) used by tests and bench.py.

Thin on purpose: every call goes straight into libknoxgpu.so; nothing here computes a result.
The names follow the reference: a Context scans Columns (runs of packs) with a filter tree of
Leaf / And / Or nodes (filter.Node, internal/operator/filter/node.go:29) and reducers
(internal/reducer/reducer.go:129).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Agg, AggState, KnoxError, Pred, ScanOut, check
from .types import FROM_NP, NP_DTYPE, BlockType, FilterMode, ReducerFunc, value_bits


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Leaf:
    """filter.Filter{Index, Mode, Value} (operator/filter/filter.go:22-30)."""

    def __init__(self, col, mode, a, b=None):
        self.col, self.mode, self.a, self.b = col, FilterMode(mode), a, b


class And:
    def __init__(self, *children):
        self.children = children


class Or:
    def __init__(self, *children):
        self.children = children


def _flatten(node, cols, out, keep):
    if isinstance(node, Leaf):
        c = cols[node.col]
        p = Pred()
        p.kind, p.nchild, p.col, p.op = 0, 0, node.col, int(node.mode)
        if c.dtype == BlockType.Bytes and node.mode in (FilterMode.In, FilterMode.NotIn):
            # the set as a packed list of [uint32 little-endian length][bytes] (include/knoxgpu.h)
            val = b"".join(len(bytes(v)).to_bytes(4, "little") + bytes(v) for v in node.a)
            keep.append(val)
            p.val, p.val_len = val, len(val)
        elif c.dtype == BlockType.Bytes:
            val = bytes(node.a)
            keep.append(val)
            p.val, p.val_len = val, len(val)
            if node.b is not None:
                val2 = bytes(node.b)
                keep.append(val2)
                p.val2, p.val2_len = val2, len(val2)
        elif node.mode in (FilterMode.In, FilterMode.NotIn):
            val = np.ascontiguousarray(node.a, dtype=NP_DTYPE[c.dtype]).tobytes()
            keep.append(val)
            p.val, p.val_len = val, len(val)
        else:
            p.a = value_bits(c.dtype, node.a)
            p.b = value_bits(c.dtype, node.b) if node.b is not None else 0
        out.append(p)
        return
    for ch in node.children:
        _flatten(ch, cols, out, keep)
    p = Pred()
    p.kind = 1 if isinstance(node, And) else 2
    p.nchild = len(node.children)
    out.append(p)


class Column:
    def __init__(self, ctx, handle, dtype, keep=None, pack_rows=0):
        self.ctx, self.h, self.dtype, self._keep = ctx, handle, BlockType(dtype), keep
        self.pack_rows = pack_rows or _lib.lib().knox_col_rows(handle)

    @property
    def rows(self):
        return _lib.lib().knox_col_rows(self.h)

    @property
    def device_ptr(self):
        return _lib.lib().knox_col_device_ptr(self.h)

    def download(self, first=0, n=None):
        n = self.rows - first if n is None else n
        out = np.empty(n, dtype=NP_DTYPE[self.dtype])
        check(_lib.lib().knox_col_download(self.ctx.h, self.h, first, n, _ptr(out)))
        return out

    def placement_class(self, pool=False):
        """the arena's placement class of the values / pointer array (pool=True: of a bytes column's pools), -1 = none"""
        return _lib.lib().knox_col_placement_class(self.h, 1 if pool else 0)

    def append_async(self, arr):
        """enqueue the copy of `arr` behind the rows present (knox_col_append_async); see Context.upload_async for lifetimes"""
        assert arr.flags.c_contiguous and FROM_NP[arr.dtype] == self.dtype
        check(_lib.lib().knox_col_append_async(self.ctx.h, self.h, _ptr(arr), arr.size), "knox_col_append_async")

    def reset(self):
        check(_lib.lib().knox_col_reset(self.ctx.h, self.h), "knox_col_reset")

    def wait(self):
        check(_lib.lib().knox_col_wait(self.ctx.h, self.h), "knox_col_wait")

    @property
    def ready(self):
        return _lib.lib().knox_col_ready(self.h) == 1

    def free(self):
        if self.h:
            _lib.lib().knox_col_free(self.h)
            self.h = None


class Bits:
    """bitset.Bitset resident on the device (internal/bitset/bitset.go:30-35)."""

    def __init__(self, ctx, size, for_col=None, wrap_ptr=None):
        self.ctx, self.size = ctx, size
        h = C.c_void_p()
        if wrap_ptr is not None:  # borrowed device memory (knox_bits_wrap)
            check(_lib.lib().knox_bits_wrap(ctx.h, C.c_void_p(wrap_ptr), size, C.byref(h)), "knox_bits_wrap")
        elif for_col is None:
            check(_lib.lib().knox_bits_alloc(ctx.h, size, C.byref(h)))
        elif isinstance(for_col, (list, tuple)):  # verified against every column of a multi-leaf scan
            arr = (C.c_void_p * len(for_col))(*[c.h for c in for_col])
            check(_lib.lib().knox_bits_alloc_for_cols(ctx.h, size, arr, len(for_col), C.byref(h)), "knox_bits_alloc_for_cols")
        else:  # placement class matched to scans of that column (include/knoxgpu.h knox_bits_alloc_for)
            check(_lib.lib().knox_bits_alloc_for(ctx.h, size, for_col.h, C.byref(h)), "knox_bits_alloc_for")
        self.h = h

    @property
    def pair_check(self):
        """(scan / read-only of the verified (column, bitset) pair, candidate windows looked at); (0.0, 0) = not verified"""
        r, k = C.c_double(), C.c_int()
        check(_lib.lib().knox_bits_pair_check(self.h, C.byref(r), C.byref(k)))
        return r.value, k.value

    @property
    def device_ptr(self):
        f = _lib.lib().knox_bits_device_ptr
        f.restype = C.c_void_p
        return f(self.h)

    def upload(self, host):
        host = np.ascontiguousarray(host, dtype=np.uint8)
        assert host.size == (self.size + 7) >> 3
        check(_lib.lib().knox_bits_upload(self.ctx.h, self.h, _ptr(host) if host.size else None))
        return self

    def upload_bitmap_container(self, container):
        c = bytes(container)
        check(_lib.lib().knox_bits_upload_bitmap_container(self.ctx.h, self.h, c, len(c)), "knox_bits_upload_bitmap_container")
        return self

    def bytes(self, poison=0):
        n = (self.size + 7) >> 3
        out = np.full(n + poison, 0xFA, dtype=np.uint8)
        check(_lib.lib().knox_bits_download(self.ctx.h, self.h, _ptr(out) if n else None))
        assert (out[n:] == 0xFA).all(), "boundary violation"
        return out[:n]

    def _un(self, name):
        check(getattr(_lib.lib(), "knox_bits_" + name)(self.ctx.h, self.h))
        return self

    def _bin(self, name, other):
        check(getattr(_lib.lib(), "knox_bits_" + name)(self.ctx.h, self.h, other.h))
        return self

    def zero(self):
        return self._un("zero")

    def one(self):
        return self._un("one")

    def neg(self):
        return self._un("neg")

    def and_(self, o):
        return self._bin("and", o)

    def andnot(self, o):
        return self._bin("andnot", o)

    def or_(self, o):
        return self._bin("or", o)

    def xor(self, o):
        return self._bin("xor", o)

    def copy_from(self, o):
        return self._bin("copy", o)

    def _flag(self, name, o):
        a, l = C.c_int(), C.c_int()
        check(getattr(_lib.lib(), "knox_bits_" + name)(self.ctx.h, self.h, o.h, C.byref(a), C.byref(l)))
        return bool(a.value), bool(l.value)

    def and_flag(self, o):
        return self._flag("and_flag", o)

    def or_flag(self, o):
        return self._flag("or_flag", o)

    def match_bool(self, mode, a, b=False, out=None):
        """Bool-column matcher (bitset/match.go:14-62): self is the column, returns the match bitset."""
        out = out if out is not None else Bits(self.ctx, self.size)
        check(_lib.lib().knox_bits_match_bool(self.ctx.h, self.h, int(mode), int(bool(a)), int(bool(b)), out.h))
        return out

    def count(self):
        c = C.c_int64()
        check(_lib.lib().knox_bits_popcount(self.ctx.h, self.h, C.byref(c)))
        return c.value

    def segment_counts(self, seg_bits):
        n = (self.size + seg_bits - 1) // seg_bits
        out = np.zeros(n, dtype=np.int64)
        check(_lib.lib().knox_bits_popcount_segments(self.ctx.h, self.h, seg_bits, _ptr(out), n))
        return out

    def indexes_view(self):
        """Bitset.Indexes into the context's page-locked result buffer: a zero-copy uint32 view, valid until the next
        indexes_view / reduce_buckets_values call on the context (copy it to keep it)."""
        p, n = C.c_void_p(), C.c_size_t()
        check(_lib.lib().knox_bits_indexes_view(self.ctx.h, self.h, C.byref(p), C.byref(n)), "knox_bits_indexes_view")
        if n.value == 0:
            return np.zeros(0, dtype=np.uint32)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(n.value,))

    def and_flag_async(self, o):
        check(_lib.lib().knox_bits_and_flag_async(self.ctx.h, self.h, o.h), "knox_bits_and_flag_async")

    def or_flag_async(self, o):
        check(_lib.lib().knox_bits_or_flag_async(self.ctx.h, self.h, o.h), "knox_bits_or_flag_async")

    def count_async(self):
        check(_lib.lib().knox_bits_popcount_async(self.ctx.h, self.h), "knox_bits_popcount_async")

    def indexes(self, capacity=None, out=None):
        """Bitset.Indexes.  out: caller's uint32 buffer (e.g. Context.host_array: page-locked, DMA'd in place) ->
        a view of it, no copy.  Without arguments: a copy of indexes_view()."""
        if out is not None:
            n = C.c_size_t()
            check(_lib.lib().knox_bits_indexes(self.ctx.h, self.h, _ptr(out), out.size, C.byref(n)))
            return out[:n.value]
        if capacity is None:
            return self.indexes_view().copy()
        cap = self.count() + 8 if capacity is None else capacity  # Bitset.Indexes pads the result by 7 (iterator.go:277-289)
        out = np.zeros(max(cap, 1), dtype=np.uint32)
        n = C.c_size_t()
        check(_lib.lib().knox_bits_indexes(self.ctx.h, self.h, _ptr(out), cap, C.byref(n)))
        return out[:n.value].copy()

    def free(self):
        if self.h:
            _lib.lib().knox_bits_free(self.h)
            self.h = None


class ScanResult:
    def __init__(self, count, rows_scanned, pack_counts, states, aggs, cols):
        self.count, self.rows_scanned, self.pack_counts = count, rows_scanned, pack_counts
        self.states, self._aggs, self._cols = states, aggs, cols

    def value(self, i):
        """Reducer.Value() of aggregate i -> (value, seen, row)."""
        func, col = self._aggs[i]
        return agg_value(self.states[i], func)


def agg_value(state, func):
    dtype = BlockType(state.dtype)
    out = np.zeros(1, dtype=NP_DTYPE[dtype])
    seen, row = C.c_int(), C.c_int64()
    check(_lib.lib().knox_agg_value(C.byref(state), int(func), _ptr(out), C.byref(seen), C.byref(row)))
    return out[0], bool(seen.value), row.value


def agg_combine(states):
    arr = states if isinstance(states, C.Array) else (AggState * len(states))(*states)
    out = AggState()
    check(_lib.lib().knox_agg_combine(arr, len(states), C.byref(out)))
    return out


class Context:
    def __init__(self, device=0, stream=None):
        L = _lib.lib()
        h = C.c_void_p()
        if stream is None:
            check(L.knox_ctx_open(device, C.byref(h)), "knox_ctx_open")
        else:
            check(L.knox_ctx_open_stream(device, C.c_void_p(stream), C.byref(h)), "knox_ctx_open_stream")
        self.h, self.L = h, L

    def close(self):
        if self.h:
            self.L.knox_ctx_close(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def sync(self):
        check(self.L.knox_ctx_sync(self.h))

    def set_tuning(self, blocks_per_cu=0):
        check(self.L.knox_ctx_set_tuning(self.h, blocks_per_cu))

    def diag_set_variant(self, variant):
        """diagnostics libraries only (`make diag`, loaded through KNOXGPU_SO): the product exports no such switch"""
        fn = getattr(self.L, "knox_diag_set_variant", None)
        if fn is None:
            raise RuntimeError("knox_diag_set_variant: not a diagnostics build (make -C knoxdb_amd/csrc diag; KNOXGPU_SO=knoxdb_amd/diag/libknoxgpu_p7.so)")
        fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int]
        check(fn(self.h, variant))

    def host_array(self, n, dtype):
        """numpy array over page-locked memory from knox_host_alloc (freed with .base_free())"""
        dt = np.dtype(dtype)
        p = C.c_void_p()
        check(self.L.knox_host_alloc(self.h, n * dt.itemsize, C.byref(p)), "knox_host_alloc")
        buf = (C.c_char * (n * dt.itemsize)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dt, count=n)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p
        return arr

    def host_free(self, arr):
        p = self._pinned.pop(arr.ctypes.data)
        check(self.L.knox_host_free(self.h, p), "knox_host_free")

    def set_placement(self, on=True):
        """device memory from the context's arena (on) or plain hipMalloc (off): include/knoxgpu.h knox_ctx_set_placement"""
        check(self.L.knox_ctx_set_placement(self.h, 1 if on else 0))

    def set_arena(self, budget_bytes):
        check(self.L.knox_ctx_set_arena(self.h, budget_bytes), "knox_ctx_set_arena")

    def reserve(self, nbytes):
        """room for columns that will be scanned together, in ONE placement class (include/knoxgpu.h knox_ctx_reserve);
        -> True when such room exists now"""
        return self.L.knox_ctx_reserve(self.h, int(nbytes)) == 0

    def set_probe_budget(self, max_launches, max_ms):
        """what the arena may spend on probe measurements from now on (include/knoxgpu.h knox_ctx_set_probe_budget)"""
        check(self.L.knox_ctx_set_probe_budget(self.h, int(max_launches), float(max_ms)), "knox_ctx_set_probe_budget")

    def arena_info(self):
        a = _lib.ArenaInfo()
        check(self.L.knox_ctx_arena_info(self.h, C.byref(a)))
        return {"budget_gib": a.budget_bytes / 2**30, "mapped_gib": a.mapped_bytes / 2**30, "used_gib": a.used_bytes / 2**30,
                "peak_used_gib": a.peak_used_bytes / 2**30, "write_buffers": a.write_buffers, "chunks": a.chunks, "probes": a.probes,
                "enabled": bool(a.enabled), "mixed_zone_gib": a.mixed_zone_bytes / 2**30,
                "spare_write_buffer_gib": a.spare_write_buffer_bytes / 2**30, "probe_launches": a.probe_launches, "probe_ms": a.probe_ms,
                "unprobed_chunks": a.unprobed_chunks, "unverified_write_buffers": a.unverified_write_buffers}

    def arena_classes(self):
        """-> (GiB of column space per placement class, classes met); include/knoxgpu.h knox_ctx_arena_classes"""
        b = (C.c_uint64 * 4)()
        n = C.c_int()
        check(self.L.knox_ctx_arena_classes(self.h, b, C.byref(n)))
        return [x / 2**30 for x in b], n.value

    def arena_trim(self):
        f = C.c_size_t()
        check(self.L.knox_ctx_arena_trim(self.h, C.byref(f)))
        return f.value

    def placement_info(self):
        b, r = C.c_int64(), C.c_int()
        check(self.L.knox_ctx_placement_info(self.h, C.byref(b), C.byref(r)))
        return {"held_bytes": b.value, "held_regions": r.value}

    def timer_start(self):
        check(self.L.knox_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_float()
        check(self.L.knox_timer_stop(self.h, C.byref(ms)))
        return ms.value

    # -- finest boundary: cmp.Int64Less(src, val, bits) int64 etc. (internal/cmp/cmp.go:6-114)
    def cmp(self, dtype, mode, src, a, b=None, poison=32):
        dtype = BlockType(dtype)
        src = np.ascontiguousarray(src, dtype=NP_DTYPE[dtype])
        n = src.size
        nb = (n + 7) >> 3
        buf = np.zeros(nb + poison, dtype=np.uint8)
        buf[nb:] = 0xFA
        av = np.array([a], dtype=NP_DTYPE[dtype])
        bv = np.array([b if b is not None else 0], dtype=NP_DTYPE[dtype])
        cnt = self.L.knox_cmp(self.h, int(dtype), int(mode), _ptr(src) if n else None, n, _ptr(av), _ptr(bv), _ptr(buf))
        if cnt < 0:
            raise KnoxError(int(cnt), "knox_cmp")
        assert (buf[nb:] == 0xFA).all(), "boundary violation"
        return buf[:nb], cnt

    # -- columns
    def upload(self, arr, pack_rows=0):
        arr = np.ascontiguousarray(arr)
        dtype = FROM_NP[arr.dtype]
        h = C.c_void_p()
        check(self.L.knox_col_upload(self.h, int(dtype), _ptr(arr) if arr.size else None, arr.size, pack_rows, C.byref(h)))
        return Column(self, h, dtype, pack_rows=pack_rows)

    # -- ingest that overlaps the scans (include/knoxgpu.h knox_col_create ... knox_col_ready)
    def col_create(self, dtype, capacity_rows, pack_rows=0):
        """an empty numeric column with room for capacity_rows rows, filled by Column.append_async"""
        h = C.c_void_p()
        check(self.L.knox_col_create(self.h, int(dtype), capacity_rows, pack_rows, C.byref(h)), "knox_col_create")
        return Column(self, h, dtype, pack_rows=pack_rows or capacity_rows)

    def upload_async(self, arr, pack_rows=0):
        """knox_col_upload_async: the copy runs on the context's copy stream; `arr` must stay alive (and, if page-locked,
        untouched) until Column.wait()"""
        assert arr.flags.c_contiguous
        dtype = FROM_NP[arr.dtype]
        h = C.c_void_p()
        check(self.L.knox_col_upload_async(self.h, int(dtype), _ptr(arr), arr.size, pack_rows, C.byref(h)), "knox_col_upload_async")
        return Column(self, h, dtype, keep=arr, pack_rows=pack_rows or arr.size)

    def upload_bytes(self, ptr, buf, pack_rows=0):
        ptr = np.ascontiguousarray(ptr, dtype=np.uint64)
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        h = C.c_void_p()
        check(self.L.knox_col_upload_bytes(self.h, _ptr(ptr) if ptr.size else None, ptr.size,
                                           _ptr(buf) if buf.size else None, buf.size, pack_rows, C.byref(h)))
        return Column(self, h, BlockType.Bytes, pack_rows=pack_rows)

    def upload_bytes_packs(self, packs, pack_rows):
        """One StringPool (ptr, buf) per pack, as the reference holds bytes blocks -> one resident bytes column."""
        packs = [(np.ascontiguousarray(p, dtype=np.uint64), np.ascontiguousarray(b, dtype=np.uint8)) for p, b in packs]
        h = C.c_void_p()
        check(self.L.knox_col_bytes_create(self.h, pack_rows, max(1, len(packs)), sum(b.size for _, b in packs), C.byref(h)),
              "knox_col_bytes_create")
        col = Column(self, h, BlockType.Bytes, pack_rows=pack_rows)
        for p, b in packs:
            check(self.L.knox_col_bytes_append_pack(self.h, h, _ptr(p), p.size, _ptr(b) if b.size else None, b.size),
                  "knox_col_bytes_append_pack")
        return col

    def generate_bytes(self, seed, n_rows, pack_rows, key_len=32, window_bits=16, packs_per_window=4):
        """synthetic bytes column, one pool per pack (knox_col_generate_bytes; SURVEY.md 8d cfg 4)"""
        h = C.c_void_p()
        check(self.L.knox_col_generate_bytes(self.h, seed, n_rows, pack_rows, key_len, window_bits, packs_per_window, C.byref(h)),
              "knox_col_generate_bytes")
        return Column(self, h, BlockType.Bytes, pack_rows=pack_rows)

    def download_bytes_pack(self, col, pack):
        """(ptr, buf) of one pack of a bytes column as the reference would hold it"""
        n, bl = C.c_size_t(), C.c_size_t()
        check(self.L.knox_col_bytes_download_pack(self.h, col.h, pack, None, 0, C.byref(n), None, 0, C.byref(bl)))
        ptr = np.zeros(n.value, dtype=np.uint64)
        buf = np.zeros(bl.value, dtype=np.uint8)
        check(self.L.knox_col_bytes_download_pack(self.h, col.h, pack, _ptr(ptr) if n.value else None, n.value, None,
                                                  _ptr(buf) if bl.value else None, bl.value, None), "knox_col_bytes_download_pack")
        return ptr, buf

    def upload_raw_containers(self, dtype, containers, pack_rows=0):
        """One on-disk raw container (bytes) per pack -> one contiguous HBM column, no host decode."""
        bufs = [np.frombuffer(bytes(c), dtype=np.uint8) for c in containers]
        ptrs = (C.c_void_p * max(1, len(bufs)))(*[b.ctypes.data for b in bufs])
        lens = (C.c_size_t * max(1, len(bufs)))(*[b.size for b in bufs])
        h = C.c_void_p()
        check(self.L.knox_col_upload_raw_containers(self.h, int(dtype), ptrs, lens, len(bufs), pack_rows, C.byref(h)),
              "knox_col_upload_raw_containers")
        return Column(self, h, dtype, keep=bufs, pack_rows=pack_rows)

    def packed_upload(self, dtype, containers, pack_rows):
        """bit-packed containers of consecutive packs, kept packed in HBM (knox_packed_upload)"""
        bufs = [np.frombuffer(bytes(c), dtype=np.uint8) for c in containers]
        ptrs = (C.c_void_p * max(1, len(bufs)))(*[b.ctypes.data for b in bufs])
        lens = (C.c_size_t * max(1, len(bufs)))(*[b.size for b in bufs])
        h = C.c_void_p()
        check(self.L.knox_packed_upload(self.h, int(dtype), ptrs, lens, len(bufs), pack_rows, C.byref(h)), "knox_packed_upload")
        return Packed(self, h, BlockType(dtype), len(bufs))

    def enc_upload(self, dtype, containers, pack_rows):
        """the integer containers of consecutive packs (any kinds, mixed), kept in the form they are cheapest to scan in (knox_enc_upload)"""
        bufs = [np.frombuffer(bytes(c), dtype=np.uint8) for c in containers]
        ptrs = (C.c_void_p * max(1, len(bufs)))(*[b.ctypes.data for b in bufs])
        lens = (C.c_size_t * max(1, len(bufs)))(*[b.size for b in bufs])
        h = C.c_void_p()
        check(self.L.knox_enc_upload(self.h, int(dtype), ptrs, lens, len(bufs), pack_rows, C.byref(h)), "knox_enc_upload")
        return Encoded(self, h, BlockType(dtype), len(bufs))

    def packed_create(self, dtype, pack_rows, max_packs, max_payload_bytes):
        """an empty packed column filled by Packed.append_async on the copy stream (knox_packed_create)"""
        h = C.c_void_p()
        check(self.L.knox_packed_create(self.h, int(dtype), pack_rows, max_packs, max_payload_bytes, C.byref(h)), "knox_packed_create")
        return Packed(self, h, BlockType(dtype), 0)

    def upload_wide_containers(self, n_limbs, containers, pack_rows=0):
        """Int128 / Int256 containers -> limb columns (most significant first) for match_wide"""
        bufs = [np.frombuffer(bytes(c), dtype=np.uint8) for c in containers]
        ptrs = (C.c_void_p * max(1, len(bufs)))(*[b.ctypes.data for b in bufs])
        lens = (C.c_size_t * max(1, len(bufs)))(*[b.size for b in bufs])
        outs = (C.c_void_p * n_limbs)()
        check(self.L.knox_col_upload_wide_containers(self.h, n_limbs, ptrs, lens, len(bufs), pack_rows, outs), "knox_col_upload_wide_containers")
        return [Column(self, C.c_void_p(outs[l]), BlockType.Int64 if l == 0 else BlockType.Uint64, keep=bufs, pack_rows=pack_rows)
                for l in range(n_limbs)]

    def upload_string_containers(self, containers, pack_rows=0):
        """One on-disk string container (bytes) per pack -> one bytes (StringPool) column."""
        bufs = [np.frombuffer(bytes(c), dtype=np.uint8) for c in containers]
        ptrs = (C.c_void_p * max(1, len(bufs)))(*[b.ctypes.data for b in bufs])
        lens = (C.c_size_t * max(1, len(bufs)))(*[b.size for b in bufs])
        h = C.c_void_p()
        check(self.L.knox_col_upload_string_containers(self.h, ptrs, lens, len(bufs), pack_rows, C.byref(h)),
              "knox_col_upload_string_containers")
        return Column(self, h, BlockType.Bytes, keep=bufs, pack_rows=pack_rows)

    def wrap(self, dtype, device_ptr, n_rows, pack_rows=0, keep=None):
        h = C.c_void_p()
        check(self.L.knox_col_wrap(self.h, int(dtype), C.c_void_p(device_ptr), n_rows, pack_rows, C.byref(h)))
        return Column(self, h, dtype, keep, pack_rows=pack_rows)

    def generate(self, dtype, kind, seed, n_rows, first_row=0, lo=0, span=0, pack_rows=0):
        h = C.c_void_p()
        check(self.L.knox_col_generate(self.h, int(dtype), kind, seed, first_row, n_rows, lo, span, pack_rows, C.byref(h)))
        return Column(self, h, dtype, pack_rows=pack_rows)

    def bits(self, size, for_col=None):
        return Bits(self, size, for_col)

    # -- filter.Match over every resident pack + reducers
    def _plan(self, tree, cols, aggs):
        nodes, keep = [], []
        if isinstance(tree, Leaf):
            tree = And(tree)  # root is always an AND node (operator/filter/node.go:25-28)
        _flatten(tree, cols, nodes, keep)
        parr = (Pred * len(nodes))(*nodes)
        carr = (C.c_void_p * len(cols))(*[c.h for c in cols])
        aggs = [(ReducerFunc(f), c) for f, c in (aggs or [])]
        aarr = (Agg * max(1, len(aggs)))()
        for i, (f, c) in enumerate(aggs):
            aarr[i].func, aarr[i].col = int(f), c
        return parr, len(nodes), carr, aggs, aarr, keep

    def scan(self, tree, cols, aggs=None, bits_out=None, pack_counts=False, pack_skip=None, row_base=0):
        parr, nn, carr, aggs, aarr, keep = self._plan(tree, cols, aggs)
        n_rows = cols[0].rows
        skip = np.ascontiguousarray(pack_skip, dtype=np.uint8) if pack_skip is not None else None
        pc = None
        if pack_counts:
            prow = self._pack_rows(cols[0])
            pc = np.zeros((n_rows + prow - 1) // prow if prow else 0, dtype=np.int64)
        states = (AggState * max(1, len(aggs)))()
        out = ScanOut()
        check(self.L.knox_scan(self.h, parr, nn, carr, len(cols), aarr, len(aggs), row_base, _ptr(skip),
                               bits_out.h if bits_out is not None else None, _ptr(pc), states, C.byref(out)), "knox_scan")
        return ScanResult(out.count, out.rows_scanned, pc, [states[i] for i in range(len(aggs))], aggs, cols)

    def scan_async(self, plan, bits_out=None, pack_counts=False, row_base=0, pack_skip=None):
        parr, nn, carr, aggs, aarr, keep = plan
        skip = np.ascontiguousarray(pack_skip, dtype=np.uint8) if pack_skip is not None else None
        check(self.L.knox_scan_async(self.h, parr, nn, carr, len(carr), aarr, len(aggs), row_base, _ptr(skip),
                                     bits_out.h if bits_out is not None else None, 1 if pack_counts else 0), "knox_scan_async")

    def scan_collect(self, plan, n_packs=0):
        parr, nn, carr, aggs, aarr, keep = plan
        states = (AggState * max(1, len(aggs)))()
        out = ScanOut()
        pc = np.zeros(n_packs, dtype=np.int64) if n_packs else None
        check(self.L.knox_scan_collect(self.h, _ptr(pc), states, len(aggs), C.byref(out)), "knox_scan_collect")
        return ScanResult(out.count, out.rows_scanned, pc, [states[i] for i in range(len(aggs))], aggs, None)

    def plan(self, tree, cols, aggs=None):
        return self._plan(tree, cols, aggs)

    def _pack_rows(self, col):
        return col.pack_rows or col.rows

    def match_range_vectors(self, mins, maxs, mode, a, b=None, mask=None, bits=None):
        """Matcher.MatchRangeVectors over per-pack min / max statistics columns -> packs that may match."""
        out = bits if bits is not None else self.bits(mins.rows)
        av = value_bits(mins.dtype, a)
        bv = value_bits(mins.dtype, b) if b is not None else 0
        check(self.L.knox_match_range_vectors(self.h, mins.h, maxs.h, int(mode), av, bv, mask.h if mask is not None else None, out.h),
              "knox_match_range_vectors")
        return out

    def match_range_vectors_bytes(self, mins, maxs, mode, val, val2=None, mask=None, bits=None):
        """MatchRangeVectors over bytes (StringPool) min / max statistics columns."""
        out = bits if bits is not None else self.bits(mins.rows)
        v, v2 = bytes(val), bytes(val2) if val2 is not None else b""
        check(self.L.knox_match_range_vectors_bytes(self.h, mins.h, maxs.h, int(mode), v, len(v), v2, len(v2),
                                                    mask.h if mask is not None else None, out.h), "knox_match_range_vectors_bytes")
        return out

    def match_range_vectors_set(self, mins, maxs, values, negate=False, mask=None, bits=None):
        """numInSetMatcher / numNotInSetMatcher.MatchRangeVectors over integer min / max statistics columns -> Bits"""
        vals = np.ascontiguousarray(values, dtype=NP_DTYPE[mins.dtype])
        out = bits if bits is not None else self.bits(mins.rows)
        check(self.L.knox_match_range_vectors_set(self.h, mins.h, maxs.h, 1 if negate else 0, _ptr(vals) if vals.size else None, vals.size,
                                                  mask.h if mask is not None else None, out.h), "knox_match_range_vectors_set")
        return out

    def match_range_vectors_wide(self, mins, maxs, mode, a, b=None, mask=None, bits=None):
        """Int128 / Int256 MatchRangeVectors over the limb columns (most significant first) of the min / max statistics -> Bits"""
        n = len(mins)
        amin = (C.c_void_p * n)(*[c.h for c in mins])
        amax = (C.c_void_p * n)(*[c.h for c in maxs])
        av = (C.c_uint64 * n)(*[int(v) & (2**64 - 1) for v in a]) if a is not None else None
        bv = (C.c_uint64 * n)(*[int(v) & (2**64 - 1) for v in b]) if b is not None else None
        out = bits if bits is not None else self.bits(mins[0].rows)
        check(self.L.knox_match_range_vectors_wide(self.h, amin, amax, n, int(mode), av, bv, mask.h if mask is not None else None, out.h),
              "knox_match_range_vectors_wide")
        return out

    def match_regexp(self, col, pattern, mask=None, bits=None):
        """bytesRegexpMatcher.MatchVector: Go regexp.Match per row of a bytes column -> (Bits, count)."""
        pat = pattern.encode() if isinstance(pattern, str) else bytes(pattern)
        out = bits if bits is not None else self.bits(col.rows)
        cnt = C.c_int64()
        check(self.L.knox_match_regexp(self.h, col.h, pat, len(pat), mask.h if mask is not None else None, out.h, C.byref(cnt)),
              "knox_match_regexp")
        return out, cnt.value

    def match_wide(self, limbs, mode, a, b=None, mask=None, bits=None):
        """Int128 / Int256 matcher over limb columns (most significant first) -> (Bits, count)."""
        n = len(limbs)
        arr = (C.c_void_p * n)(*[c.h for c in limbs])
        av = (C.c_uint64 * n)(*[int(v) & (2**64 - 1) for v in a])
        bv = (C.c_uint64 * n)(*[int(v) & (2**64 - 1) for v in b]) if b is not None else None
        out = bits if bits is not None else self.bits(limbs[0].rows)
        cnt = C.c_int64()
        check(self.L.knox_match_wide(self.h, arr, n, int(mode), av, bv, mask.h if mask is not None else None, out.h, C.byref(cnt)),
              "knox_match_wide")
        return out, cnt.value

    def match_wide_set(self, limbs, values, negate=False, mask=None, bits=None):
        """Int128 / Int256 IN / NOT IN: values = list of limb tuples (most significant first) -> Bits"""
        n = len(limbs)
        arr = (C.c_void_p * n)(*[c.h for c in limbs])
        flat = [int(x) & (2**64 - 1) for v in values for x in v]
        sv = (C.c_uint64 * max(1, len(flat)))(*flat)
        out = bits if bits is not None else self.bits(limbs[0].rows)
        check(self.L.knox_match_wide_set(self.h, arr, n, 1 if negate else 0, sv, len(values), mask.h if mask is not None else None, out.h),
              "knox_match_wide_set")
        return out

    def reduce(self, col, mask=None, row_base=0, moments=False):
        """Masked reduction of one column; moments=True adds the var/std second pass."""
        st = AggState()
        check(self.L.knox_reduce(self.h, col.h, mask.h if mask is not None else None, row_base, C.byref(st)))
        if moments:
            check(self.L.knox_reduce_moments(self.h, col.h, mask.h if mask is not None else None, C.byref(st)), "knox_reduce_moments")
        return st

    def reduce_buckets(self, time_col, value_col, t_from, step, n_buckets, mask=None, row_base=0, assume_sorted=False, moments=False, out=None, recheck=False):
        """One mergeable state per time window [t_from + k*step, +step): NativeBucket semantics.
        moments=True adds the var/std second pass per window.  out: a (AggState * n_buckets) array to fill (a caller that
        asks again and again keeps one; a fresh 144 B x n_buckets array is first-touch page faults on every call)."""
        states = out if out is not None else (AggState * n_buckets)()
        check(self.L.knox_reduce_buckets(self.h, time_col.h, value_col.h, mask.h if mask is not None else None, t_from, step,
                                         n_buckets, row_base, (1 if assume_sorted else 0) | (2 if moments else 0) | (16 if recheck else 0), states), "knox_reduce_buckets")
        return states   # ctypes array: indexable / iterable, no per-window Python objects until touched

    def reduce_buckets_view(self, time_col, value_col, t_from, step, n_buckets, mask=None, row_base=0, assume_sorted=False, moments=False):
        """reduce_buckets into the context's page-locked result buffer: a ctypes array VIEW of n_buckets states (valid
        until the next *_view / *_values call; no copy into pageable memory)"""
        p = C.POINTER(AggState)()
        check(self.L.knox_reduce_buckets_view(self.h, time_col.h, value_col.h, mask.h if mask is not None else None, t_from, step,
                                              n_buckets, row_base, (1 if assume_sorted else 0) | (2 if moments else 0), C.byref(p)),
              "knox_reduce_buckets_view")
        return C.cast(p, C.POINTER(AggState * n_buckets)).contents

    def reduce_buckets_values(self, time_col, value_col, func, t_from, step, n_buckets, mask=None, assume_sorted=False, recheck=False):
        """Reducer.Value() per window, finalised on the device -> (values[n_buckets] of the column's type, seen[n_buckets])
        as zero-copy views of the context's page-locked result buffer (valid until the next *_view / *_values call)."""
        pv, ps = C.c_void_p(), C.c_void_p()
        check(self.L.knox_reduce_buckets_values(self.h, time_col.h, value_col.h, mask.h if mask is not None else None, t_from, step,
                                                n_buckets, int(func), (1 if assume_sorted else 0) | (16 if recheck else 0), C.byref(pv), C.byref(ps)),
              "knox_reduce_buckets_values")
        dt = np.dtype(NP_DTYPE[value_col.dtype])
        vals = np.ctypeslib.as_array(C.cast(pv, C.POINTER(C.c_uint8)), shape=(n_buckets * dt.itemsize,)).view(dt)
        seen = np.ctypeslib.as_array(C.cast(ps, C.POINTER(C.c_uint8)), shape=(n_buckets,))
        return vals, seen

    def bits_collect(self):
        """Oldest uncollected async bitset result -> (any, all, count); the fields that do not apply are None."""
        a, l, c = C.c_int(), C.c_int(), C.c_int64()
        check(self.L.knox_bits_collect(self.h, C.byref(a), C.byref(l), C.byref(c)), "knox_bits_collect")
        return (None if a.value < 0 else bool(a.value), None if l.value < 0 else bool(l.value), None if c.value < 0 else c.value)

    def windows_last_path(self):
        """how the last reduce_buckets* call reduced (include/knoxgpu.h knox_windows_path): 1 ordered, 2 sorted, 3 sorted after a
        compaction, 4 / 5 / 6 without a sort (values / states / exact float sums)"""
        return self.L.knox_windows_last_path(self.h)

    def reduce_buckets_sequential(self, time_col, value_col, func, t_from, step, n_buckets, mask=None, assume_sorted=False, form=None):
        """The reference's own reducer per window, row by row (bit-exact Value() for every reducer, join
        reducers included) -> (values[n_buckets], seen[n_buckets])."""
        vals = np.zeros(n_buckets, dtype=NP_DTYPE[value_col.dtype])
        seen = np.zeros(n_buckets, dtype=np.uint8)
        check(self.L.knox_reduce_buckets_sequential(self.h, time_col.h, value_col.h, mask.h if mask is not None else None, t_from, step,
                                                    n_buckets, int(func), (1 if assume_sorted else 0) | {None: 0, "lane": 4, "wave": 8}[form],
                                                    _ptr(vals), _ptr(seen)),
              "knox_reduce_buckets_sequential")
        return vals, seen.astype(bool)

    def reduce_sequential(self, col, func, mask=None):
        out = np.zeros(1, dtype=NP_DTYPE[col.dtype])
        seen, row = C.c_int(), C.c_int64()
        check(self.L.knox_reduce_sequential(self.h, col.h, mask.h if mask is not None else None, int(func), _ptr(out),
                                            C.byref(seen), C.byref(row)))
        return out[0], bool(seen.value), row.value

    # -- bloom
    def bloom_upload(self, filters, offsets):
        filters = np.ascontiguousarray(filters, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        h = C.c_void_p()
        check(self.L.knox_bloom_upload(self.h, _ptr(filters) if filters.size else None, _ptr(offsets), offsets.size - 1, C.byref(h)))
        return Bloom(self, h)

    def bloom_build(self, col, factor):
        """per-pack bloom filters of a numeric column of any width, built on the device"""
        h = C.c_void_p()
        check(self.L.knox_bloom_build(self.h, col.h, factor, C.byref(h)), "knox_bloom_build")
        return Bloom(self, h)

    def bloom_build_u64(self, col, factor):
        h = C.c_void_p()
        check(self.L.knox_bloom_build_u64(self.h, col.h, factor, C.byref(h)))
        return Bloom(self, h)


class Packed:
    """A column of bit-packed containers resident in HBM, scanned without materialization."""

    def __init__(self, ctx, handle, dtype, n_packs):
        self.ctx, self.h, self.dtype, self.n_packs = ctx, handle, dtype, n_packs

    @property
    def rows(self):
        return _lib.lib().knox_packed_rows(self.h)

    @property
    def nbytes(self):
        return _lib.lib().knox_packed_bytes(self.h)

    def append_async(self, containers):
        """enqueue the bit-packed containers of the next packs (numpy uint8 arrays -- page-locked ones are read in place and must
        stay untouched until wait() -- or bytes)"""
        bufs = [c if isinstance(c, np.ndarray) else np.frombuffer(bytes(c), dtype=np.uint8) for c in containers]
        ptrs = (C.c_void_p * max(1, len(bufs)))(*[b.ctypes.data for b in bufs])
        lens = (C.c_size_t * max(1, len(bufs)))(*[b.size for b in bufs])
        check(_lib.lib().knox_packed_append_async(self.ctx.h, self.h, ptrs, lens, len(bufs)), "knox_packed_append_async")
        self.n_packs += len(bufs)
        self._keep = bufs

    def reset(self):
        check(_lib.lib().knox_packed_reset(self.ctx.h, self.h), "knox_packed_reset")
        self.n_packs = 0

    def wait(self):
        check(_lib.lib().knox_packed_wait(self.ctx.h, self.h), "knox_packed_wait")

    def scan(self, mode, a, b=None, want_sum=False, bits_out=None, pack_counts=False):
        """-> (count, sum state or None, per-pack counts or None)"""
        av = value_bits(self.dtype, a)
        bv = value_bits(self.dtype, b) if b is not None else 0
        cnt = C.c_int64()
        st = AggState()
        pc = np.zeros(self.n_packs, dtype=np.int64) if pack_counts else None
        check(_lib.lib().knox_packed_scan(self.ctx.h, self.h, int(mode), av, bv, 1 if want_sum else 0, bits_out.h if bits_out is not None else None,
                                          pc.ctypes.data_as(C.POINTER(C.c_int64)) if pc is not None else None, C.byref(cnt), C.byref(st)),
              "knox_packed_scan")
        return cnt.value, (st if want_sum else None), pc

    def free(self):
        if self.h:
            check(_lib.lib().knox_packed_free(self.h))
            self.h = None


class Encoded:
    """An integer column whose packs stay encoded in HBM -- dictionary / run-end / constant / delta as they are, everything else
    decoded on the device -- and are scanned in that form (knox_enc_*)."""

    def __init__(self, ctx, handle, dtype, n_packs):
        self.ctx, self.h, self.dtype, self.n_packs = ctx, handle, dtype, n_packs

    @property
    def rows(self):
        return _lib.lib().knox_enc_rows(self.h)

    @property
    def nbytes(self):
        return _lib.lib().knox_enc_bytes(self.h)

    @property
    def container_bytes(self):
        return _lib.lib().knox_enc_container_bytes(self.h)

    def kinds(self):
        """-> packs scanned as (dictionary, run-end, constant / delta, raw)"""
        n = [C.c_size_t() for _ in range(4)]
        check(_lib.lib().knox_enc_kinds(self.h, *[C.byref(x) for x in n]), "knox_enc_kinds")
        return tuple(x.value for x in n)

    def scan(self, mode, a, b=None, want_sum=False, bits_out=None, pack_counts=False):
        """-> (count, sum state or None, per-pack counts or None)"""
        av = value_bits(self.dtype, a)
        bv = value_bits(self.dtype, b) if b is not None else 0
        cnt = C.c_int64()
        st = AggState()
        pc = np.zeros(self.n_packs, dtype=np.int64) if pack_counts else None
        check(_lib.lib().knox_enc_scan(self.ctx.h, self.h, int(mode), av, bv, 1 if want_sum else 0, bits_out.h if bits_out is not None else None,
                                       pc.ctypes.data_as(C.POINTER(C.c_int64)) if pc is not None else None, C.byref(cnt), C.byref(st)),
              "knox_enc_scan")
        return cnt.value, (st if want_sum else None), pc

    def free(self):
        if self.h:
            check(_lib.lib().knox_enc_free(self.h))
            self.h = None


class Bloom:
    """Per-pack bloom filters of one column, resident on the device (bloom.Filter per pack)."""

    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle

    @property
    def packs(self):
        return _lib.lib().knox_bloom_packs(self.h)

    def build_info(self):
        """-> (batches the build ran in, batches that took the exact passes); include/knoxgpu.h knox_bloom_build_info"""
        a, b = C.c_int(), C.c_int()
        check(_lib.lib().knox_bloom_build_info(self.h, C.byref(a), C.byref(b)), "knox_bloom_build_info")
        return a.value, b.value

    def download(self):
        nbytes = _lib.lib().knox_bloom_bytes(self.h)
        filters = np.zeros(nbytes, dtype=np.uint8)
        offsets = np.zeros(self.packs + 1, dtype=np.uint64)
        check(_lib.lib().knox_bloom_download(self.ctx.h, self.h, _ptr(filters) if nbytes else None, nbytes, _ptr(offsets)))
        return filters, offsets

    def probe(self, h):
        out = np.zeros(self.packs, dtype=np.uint8)
        check(_lib.lib().knox_bloom_probe(self.ctx.h, self.h, h, _ptr(out)))
        return out

    def free(self):
        if self.h:
            _lib.lib().knox_bloom_free(self.h)
            self.h = None


def hash_bytes(val):
    """hash.Hash: XXH3-64 of a byte string (knox_hash_bytes)"""
    v = bytes(val)
    return _lib.lib().knox_hash_bytes(v, len(v))


FILL_MODES = {"none": 0, "null": 1, "last": 2, "linear": 3, "zero": 4}


def fill_windows(dtype, mode, values, seen, t_from, step, units_per_second=1, limit=None):
    """NativeBucket.Emit's gap filling over dense per-window values -> (values, flags 0/1/2, emitted count)"""
    vals = np.ascontiguousarray(values, dtype=NP_DTYPE[BlockType(dtype)])
    sn = np.ascontiguousarray(seen, dtype=np.uint8)
    out = np.zeros_like(vals)
    flags = np.zeros(vals.size, dtype=np.uint8)
    n = C.c_size_t()
    check(_lib.lib().knox_fill_windows(int(dtype), FILL_MODES[mode] if isinstance(mode, str) else int(mode), _ptr(vals), _ptr(sn), vals.size,
                                       t_from, step, units_per_second, vals.size if limit is None else limit, _ptr(out), _ptr(flags), C.byref(n)),
          "knox_fill_windows")
    return out, flags, n.value


def regexp_match_value(pattern, value):
    """bytesRegexpMatcher.MatchValue for one value (host): True / False; KnoxError(UNSUPPORTED) outside the subset"""
    pat = pattern.encode() if isinstance(pattern, str) else bytes(pattern)
    v = bytes(value)
    rc = _lib.lib().knox_regexp_match_value(pat, len(pat), v, len(v))
    if rc < 0:
        check(rc, "knox_regexp_match_value")
    return bool(rc)


def hash_value(dtype, v):
    a = np.array([v], dtype=NP_DTYPE[BlockType(dtype)])
    return _lib.lib().knox_hash_value(int(dtype), _ptr(a))


def device_count():
    n = C.c_int()
    rc = _lib.lib().knox_device_count(C.byref(n))
    return n.value if rc == 0 else 0
