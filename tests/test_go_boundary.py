"""The Go side of the boundary (integration/go) is source only -- the image has no Go toolchain -- so it is checked
statically: integration/check_go_calls.py compares every `C.knox_*` call site with include/knoxgpu.h, and the rebind
file must cover every exported compare func var of internal/cmp/cmp.go:6-114 (10 numeric types x 7 modes through
knox_cmp, Int128 / Int256 x 7 modes through knox_match_wide)."""
import importlib.util
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GO = os.path.join(ROOT, "integration", "go")


def _checker():
    spec = importlib.util.spec_from_file_location("check_go_calls", os.path.join(ROOT, "integration", "check_go_calls.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_every_c_call_in_the_go_sources_matches_the_header():
    errors, checked, bound, _ = _checker().check()
    assert errors == [], "\n".join(errors)
    assert checked >= 30
    for name in ("knox_cmp", "knox_match_wide", "knox_scan", "knox_bits_alloc_for", "knox_bits_indexes", "knox_agg_value",
                 "knox_agg_combine", "knox_bloom_build", "knox_bloom_probe", "knox_col_bytes_append_pack",
                 "knox_comm_allreduce_sum_count", "knox_comm_allgather_states", "knox_ctx_set_arena"):
        assert name in bound, name


def test_the_checker_rejects_wrong_arity_wrong_types_and_unknown_fields(tmp_path):
    mod = _checker()
    work = tmp_path / "go"
    shutil.copytree(GO, work)
    path = work / "knoxgpu" / "knoxgpu.go"
    src = path.read_text()
    bad = src.replace("C.knox_bits_alloc(c.p, C.size_t(size), &b)", "C.knox_bits_alloc(c.p, &b)")                   # arity
    bad = bad.replace("C.knox_bloom_probe(c.p, b.p, C.uint64_t(h),", "C.knox_bloom_probe(c.p, b.p, C.int(h),")     # scalar type
    bad = bad.replace("C.knox_ctx_set_arena(c.p, C.size_t(bytes))", "C.knox_ctx_set_arena(b.p, C.size_t(bytes))")  # untypable handle
    bad = bad.replace("out.rows_scanned", "out.rows_read")                                                          # struct field
    bad = bad.replace("C.knox_hash_bytes(", "C.knox_hash_blob(")                                                    # unknown entry
    assert bad != src
    path.write_text(bad)
    mod.GO_DIR = str(work)
    errors, _, _, _ = mod.check()
    text = "\n".join(errors)
    assert "knox_bits_alloc: 2 arguments, header takes 3" in text
    assert "knox_bloom_probe: argument 3" in text and "is int, header takes uint64_t" in text
    assert "knox_ctx_set_arena: argument 1" in text
    assert "rows_read" in text
    assert "knox_hash_blob: not declared" in text


def test_all_84_compare_func_vars_are_rebound():
    src = open(os.path.join(GO, "cmp_knoxgpu.go")).read()
    rebound = set(re.findall(r"^\t+(\w+) = (?:bind|func)", src, flags=re.M))
    modes = ["Equal", "NotEqual", "Less", "LessEqual", "Greater", "GreaterEqual", "Between"]
    types = ["Uint8", "Uint16", "Uint32", "Uint64", "Int8", "Int16", "Int32", "Int64", "Float32", "Float64", "Int128", "Int256"]
    want = {t + m for t in types for m in modes}
    assert len(want) == 84
    assert want - rebound == set(), sorted(want - rebound)
    # and the kill switch + the fallback are there
    assert "knoxgpu.Enabled()" in src and "return prev(" in src
    lib = open(os.path.join(GO, "knoxgpu", "knoxgpu.go")).read()
    assert 'os.Getenv("KNOX_NOGPU")' in lib


def test_matcher_and_operator_files_implement_the_reference_interfaces():
    flt = open(os.path.join(GO, "filter_knoxgpu.go")).read()
    assert "type gpuNumMatcher[T Number] struct" in flt
    assert re.search(r"func \(m \*gpuNumMatcher\[T\]\) MatchVector\(b \*block\.Block, bits, mask \*bitset\.Bitset\)", flt)
    byt = open(os.path.join(GO, "bytes_knoxgpu.go")).read()
    assert re.search(r"^package filter$", byt, flags=re.M) and "type gpuBytesMatcher struct" in byt
    assert re.search(r"func \(m \*gpuBytesMatcher\) MatchVector\(b \*block\.Block, bits, mask \*bitset\.Bitset\)", byt)
    assert re.search(r"func \(f gpuBytesMatcherFactory\) New\(m FilterMode\) Matcher", byt)
    # every knoxgpu.* name the matcher uses exists in the cgo package; table-dependent patterns are guarded by the version check
    lib = open(os.path.join(GO, "knoxgpu", "knoxgpu.go")).read()
    for name in set(re.findall(r"knoxgpu\.([A-Z]\w+)", byt)):
        assert re.search(r"^(func %s\b|type %s\b|\t%s\b)" % (name, name, name), lib, flags=re.M) or re.search(r"\b%s\b.*=" % name, lib), name
    assert "knoxgpu.RegexpOnDevice(v)" in byt and "unicode.Version == RegexpUnicodeVersion()" in lib
    assert byt.count("m.Matcher.MatchVector(b, bits, mask)") >= 5      # every failure path ends in the CPU matcher
    # the arena hook: the reference's Allocator interface (internal/arena/allocator.go:14-17), page-locked slabs from the library
    ar = open(os.path.join(GO, "arena_knoxgpu.go")).read()
    assert re.search(r"^package arena$", ar, flags=re.M)
    assert re.search(r"func \(a \*pinnedAllocator\[T\]\) Alloc\(sz int\) any", ar) and re.search(r"func \(a \*pinnedAllocator\[T\]\) Free\(val any, sz int\)", ar)
    assert "a.gpu.HostAlloc(n * each)" in ar and ar.count("_arena.alloc[allocType") == 10 and "allocTypeBytesSlice" not in ar.split("func init()")[1]
    assert "func (c *Ctx) HostAlloc(bytes int) (unsafe.Pointer, error)" in lib and "func Default() *Ctx" in lib
    op = open(os.path.join(GO, "operator_knoxgpu.go")).read()
    # the tree plan takes bytes blocks, bytes / regexp leaves and IN / NOT IN sets (types the reference really has: no BlockString)
    for needle in ("case types.BlockBytes:", "knoxgpu.UploadPool(g, ptr, buf)", "case types.FilterModeIn, types.FilterModeNotIn:",
                   "knoxgpu.SetBytes(f.Matcher.Value())", "case types.FilterModeRegexp:", "knoxgpu.RegexpOnDevice(pat)", "p.Val, okA = rg[0].([]byte)"):
        assert needle in op, needle
    ref_types = open(os.path.join(ROOT, "include", "knoxgpu.h")).read()
    assert "KNOX_IN = 7, KNOX_NOTIN = 8" in ref_types and "KNOX_REGEXP = 10" in ref_types and "KNOX_BYTES = 12" in ref_types   # = FilterModeIn .. / BlockBytes
    assert "BlockString" not in op
    for sig in (r"func \(op \*PhysicalGpuFilter\) Process\(ctx context\.Context, src \*pack\.Package\) \(\*pack\.Package, Result\)",
                r"func \(op \*PhysicalGpuFilter\) Finalize\(ctx context\.Context\) error",
                r"func \(op \*PhysicalGpuFilter\) Err\(\) error", r"func \(op \*PhysicalGpuFilter\) Close\(\)",
                r"func \(r \*GpuAggregateResult\) Append\(ctx context\.Context, src \*pack\.Package\) error",
                r"func \(r \*GpuAggregateResult\) Len\(\) int"):
        assert re.search(sig, op), sig


def test_run_level_reader_is_committed_source_with_the_readers_contract():
    """integration/go/run_knoxgpu.go: many packs resident, ONE scan per run, per-pack results served to the unchanged tail of
    the reference's loop (reader.go:288-450); QueryStats gets the device's share (internal/query/stats.go:58-74)"""
    run = open(os.path.join(GO, "run_knoxgpu.go")).read()
    assert re.search(r"^package table$", run, flags=re.M)
    assert re.search(r"func \(r \*Reader\) nextQueryMatchGpu\(ctx context\.Context\) \(\*pack\.Package, error\)", run)
    assert re.search(r"func \(g \*GpuRun\) WithQuery\(n \*filter\.Node, aggs \[\]knoxgpu\.Agg\) bool", run)
    body = run[run.index("func (g *GpuRun) scan("):run.index("// reducer functions knox_enc_scan folds by itself")]
    assert body.count("g.gpu.Scan(") == 1 and "true, len(g.packs))" in body          # ONE scan per run, with per-pack counts
    assert body.count("bits.Download(") == 1                                           # ONE copy of the run's bitset
    assert "UploadRawContainers" in body and "knoxgpu.AppendAsync(" in run             # encoded and materialized blocks
    for key in ("GPU_RUNS_KEY", "GPU_ROWS_SCANNED_KEY", "GPU_SCAN_USEC_KEY", "GPU_INGEST_USEC_KEY", "PACKS_SCHEDULED_KEY"):
        assert "Count(%s" % key in run, key
    assert "r.finishMatch(ctx, pk.key, pk.ver, pk.n)" in run and "ResetCount(int(pk.count))" in run
    # round 4: a run whose one filter column arrived as containers and whose plan is one numeric leaf stays encoded on the device
    enc = run[run.index("func (g *GpuRun) scanEncoded("):run.index("// nextQueryMatchGpu has nextQueryMatch's contract")]
    assert "knoxgpu.UploadEncoded(g.gpu, int(bt), g.encoded[0], g.packRows)" in enc and "defer enc.Free()" in enc
    assert enc.count("enc.Scan(") == 1 and enc.count("bits.Download(") == 1 and "GPU_ENCODED_RUNS_KEY" in enc
    assert "return false, nil" in enc                       # every shape it does not take goes to the decoding path
    assert body.index("g.scanEncoded(r, rows)") < body.index("knoxgpu.UploadRawContainers(")
    op = open(os.path.join(GO, "operator_knoxgpu.go")).read()
    assert "evictOthers(k.pack, k.version" in op and "func (d *deviceCols) evict(" not in op   # no eviction of the package being planned


def test_the_checker_rejects_a_context_call_outside_the_lock(tmp_path):
    mod = _checker()
    work = tmp_path / "go"
    shutil.copytree(GO, work)
    path = work / "knoxgpu" / "knoxgpu.go"
    src = path.read_text()
    i = src.index("func (b *Bits) Free() {")
    j = src.index("}", i)
    bad = src[:i] + "func (b *Bits) Free() {\n\tC.knox_bits_free(b.p)\n" + src[j:]
    k = bad.index("func NewBits(c *Ctx, size int) (*Bits, error) {")
    bad = bad[:k] + bad[k:].replace("\tc.mu.Lock()\n\tdefer c.mu.Unlock()\n", "", 1)
    assert bad != src
    path.write_text(bad)
    mod.GO_DIR = str(work)
    errors, _, _, _ = mod.check()
    text = "\n".join(errors)
    assert "knox_bits_free: called without the context's lock" in text
    assert "knox_bits_alloc: called without the context's lock" in text


def test_cmp_rebinding_has_a_row_floor_and_a_context_pool(tmp_path):
    """round-3 review item 5: a rebound cmp func var below knoxgpu.CmpMinRows rows calls the kernel installed before (a
    device round trip cannot answer a 2^16-row pack in the 4.5-7.4 us the AVX2 kernel takes), and the calls above it take
    a context from a pool instead of queueing behind one mutex; the checker enforces Get / deferred Put pairs."""
    src = open(os.path.join(GO, "cmp_knoxgpu.go")).read()
    lib = open(os.path.join(GO, "knoxgpu", "knoxgpu.go")).read()
    assert "var CmpMinRows = func() int {" in lib and 'os.Getenv("KNOX_CMP_MIN_ROWS")' in lib and "type Pool struct" in lib
    for fn in ("bind1", "bind2", "bind128", "bind256"):
        body = src[src.index("func %s" % fn):]
        body = body[:body.index("\n}\n")]
        assert "knoxgpu.CmpMinRows" in body and "gpus.Get()" in body and "defer gpus.Put(gpu)" in body, fn
        assert body.index("knoxgpu.CmpMinRows") < body.index("gpus.Get()"), fn     # the floor comes first: short vectors never wait for a context
    assert src.count("gpus.Get()") == src.count("defer gpus.Put(gpu)") == 6         # + the two Between closures of Int128 / Int256
    mod = _checker()
    work = tmp_path / "go"
    shutil.copytree(GO, work)
    path = work / "cmp_knoxgpu.go"
    path.write_text(path.read_text().replace("\t\tgpu := gpus.Get()\n\t\tdefer gpus.Put(gpu)\n", "\t\tgpu := gpus.Get()\n", 1))
    mod.GO_DIR = str(work)
    errors, _, _, _ = mod.check()
    assert any("is not followed by `defer gpus.Put(gpu)`" in e for e in errors), errors
