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
    op = open(os.path.join(GO, "operator_knoxgpu.go")).read()
    for sig in (r"func \(op \*PhysicalGpuFilter\) Process\(ctx context\.Context, src \*pack\.Package\) \(\*pack\.Package, Result\)",
                r"func \(op \*PhysicalGpuFilter\) Finalize\(ctx context\.Context\) error",
                r"func \(op \*PhysicalGpuFilter\) Err\(\) error", r"func \(op \*PhysicalGpuFilter\) Close\(\)",
                r"func \(r \*GpuAggregateResult\) Append\(ctx context\.Context, src \*pack\.Package\) error",
                r"func \(r \*GpuAggregateResult\) Len\(\) int"):
        assert re.search(sig, op), sig
