"""float64 sum / mean: the distance between what the GPU returns (the correctly rounded EXACT sum / mean)
and the LITERAL reference -- SumReducer's left-to-right `r.v += v` and MeanReducer's Welford recurrence
(internal/reducer/reducer.go:168-178, 318-349), restated by the oracle and folded sequentially on the host -- measured on
BASELINE configs[2]'s two datasets at 2^20, 2^26, 2^28 and 2^30 rows (KNOX_ULP_FAST=1: without the last), and the relations that can be
proved on the exact grid, asserted:

  * dataset "grid"   A = k * 2^-10, k < 2^20 (every partial sum is exact in float64 at these sizes): the literal sum IS the
                     exact sum, so fused sum == literal sum bit for bit, and fused mean == literal_sum / count (one IEEE
                     division: the correctly rounded exact mean).  Welford's mean rounds once per row and drifts.
  * dataset "random" A uniform in [0, 1): nothing is exact; fused sum / mean are within 1 ulp (in practice 0) of the exact
                     values, the literal results are wherever their rounding took them -- the distance is REPORTED
                     (north_star asks for <= 1 ulp of the Go path: that holds against the exact value, not against a
                     sequential fold of 10^9 roundings).

The measured distances are written to gpurun_out/ulp_report.json (copied to profiles/r03_float_distance.json)."""
import json
import os

import numpy as np
import pytest

import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType, FilterMode, ReducerFunc
from oracle import oracle as O

pytestmark = pytest.mark.gpu
R = ReducerFunc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def ulps(a, b):
    ia, ib = np.array([a, b], dtype=np.float64).view(np.int64)
    ia = ia if ia >= 0 else np.int64(-(2**63)) - ia
    ib = ib if ib >= 0 else np.int64(-(2**63)) - ib
    return abs(int(ia) - int(ib))


def test_float_sum_mean_distance_to_the_literal_reference():
    # 2^30 rows -- the size the north star's "within 1 ulp" is quoted on -- is part of the default suite since round 4 (about a
    # minute of sequential folding on the host; the driver's GPU run used 88 of its 900 s); KNOX_ULP_FAST=1 stops at 2^28
    sizes = [20, 26, 28] + ([] if os.environ.get("KNOX_ULP_FAST") == "1" else [30])
    report = []
    with K.Context(0) as ctx:
        for lg in sizes:
            n, pack = 1 << lg, 1 << min(lg, 20)
            for name, kind, seed, tree in (("grid", 1, 3, And(Leaf(0, FilterMode.Ge, 256.0), Leaf(1, FilterMode.Lt, 0.5))),
                                           ("random", 2, 7, And(Leaf(0, FilterMode.Lt, 0.75), Leaf(1, FilterMode.Lt, 0.5)))):
                A = ctx.generate(BlockType.Float64, kind, seed, n, pack_rows=pack)
                B = ctx.generate(BlockType.Float64, 2, 4, n, pack_rows=pack)
                bits = ctx.bits(n)
                res = ctx.scan(tree, [A, B], aggs=[(R.Sum, 0), (R.Mean, 0)], bits_out=bits)
                f_sum, f_mean = float(res.value(0)[0]), float(res.value(1)[0])
                # the same rows and the same selection on the host (the generator's spec, the oracle's compare kernels)
                xa = O.generate_mt(O.F64, kind, seed, 0, n)
                wb = bits.bytes()
                lim = 256.0 if name == "grid" else 0.75
                sel_a = (xa >= lim) if name == "grid" else (xa < lim)
                xb = O.generate_mt(O.F64, 2, 4, 0, n)
                assert (np.packbits(sel_a & (xb < 0.5), bitorder="little") == wb).all(), "selection differs from the host's"
                del xb
                lit_sum = float(O.reduce(O.F64, O.R_SUM, xa, wb)[0])      # r.v += v, row order
                lit_mean = float(O.reduce(O.F64, O.R_MEAN, xa, wb)[0])    # Welford, row order
                ex_sum, ex_mean, cnt = O.exact_sum_f64(xa, wb)            # correctly rounded exact values
                assert cnt == res.count
                row = {"dataset": name, "rows_log2": lg, "selected": int(cnt),
                       "ulp_fused_sum_vs_exact": ulps(f_sum, ex_sum), "ulp_fused_mean_vs_exact": ulps(f_mean, ex_mean),
                       "ulp_fused_sum_vs_literal": ulps(f_sum, lit_sum), "ulp_fused_mean_vs_literal": ulps(f_mean, lit_mean),
                       "ulp_literal_sum_vs_exact": ulps(lit_sum, ex_sum), "ulp_literal_mean_vs_exact": ulps(lit_mean, ex_mean)}
                report.append(row)
                # what the product promises everywhere: the correctly rounded exact result
                assert row["ulp_fused_sum_vs_exact"] <= 1 and row["ulp_fused_mean_vs_exact"] <= 1, row
                if name == "grid":
                    # provable on the exact grid: the literal sum is exact, so the results coincide bit for bit, and the
                    # fused mean is ONE IEEE division of that sum by the count
                    assert f_sum == lit_sum == ex_sum, row
                    assert f_mean == lit_sum / cnt, row
                for o in (A, B, bits):
                    o.free()
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump(report, open(os.path.join(out, "ulp_report.json"), "w"), indent=1)
    for r in report:
        print(r)
