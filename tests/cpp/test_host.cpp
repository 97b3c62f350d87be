// test_host.cpp -- the C++ host mirror (knox_host.hpp) exercised the way the reference's own
// tests exercise its Go interfaces: internal/operator/filter/match_test.go (TestMatchValue,
// TestMatchVector), internal/bitset/bitset_test.go (method-level cnt short-cuts) and the
// operator pipeline's PhysicalFilter (operator/filter.go:29-37).  Expected values are the
// reference's golden vector i64_s0 with its results (internal/cmp/tests/int64.go:14-45).
// Runs on the GPU box: g++ tests/cpp/test_host.cpp -Iknoxdb_amd/csrc/host -Lknoxdb_amd -lknoxgpu
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "knox_host.hpp"

using namespace knox;

static int fails = 0;
#define EXPECT(c)                                                        \
    do {                                                                 \
        if (!(c)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); fails++; } \
    } while (0)

int main() {
    Context ctx(0);
    const std::vector<int64_t> s0 = {0, 5, 3, 5, 7, 5, 5, 9, 3, 5, 5, 5, 5, 0, 113, 12, 4, 2, 3, 5, 7, 3, 5, 9, 3, 13, 5, 5, 42, 5, 113, 12};
    std::vector<double> f0(s0.begin(), s0.end());
    pack::Package pkg;
    pkg.AddBlock(std::make_shared<block::Block>(ctx, s0));
    pkg.AddBlock(std::make_shared<block::Block>(ctx, f0));

    // --- TestMatchVector + golden bytes: Between [5,10] on i64_s0 -> fa 1e d8 2c, count 17
    {
        auto m = filter::NumMatcherFactory<int64_t>().New(KNOX_RANGE);
        m->WithValue(filter::Value::Range<int64_t>(5, 10));
        bitset::Bitset bits(ctx, pkg.Len());
        m->MatchVector(ctx, pkg.Block(0), bits, nullptr);
        EXPECT(bits.Len() == 32);
        EXPECT(bits.Count() == 17);
        auto b = bits.Bytes();
        EXPECT(b.size() == 4 && b[0] == 0xfa && b[1] == 0x1e && b[2] == 0xd8 && b[3] == 0x2c);
    }
    // --- TestMatchValue: the scalar side agrees with the vector side
    {
        filter::numMatcher<int64_t> m(KNOX_LT);
        m.WithValue(filter::Value::Of<int64_t>(5));
        bitset::Bitset bits(ctx, pkg.Len());
        m.MatchVector(ctx, pkg.Block(0), bits, nullptr);
        auto b = bits.Bytes();
        for (int i = 0; i < 32; i++) EXPECT(((b[i >> 3] >> (i & 7)) & 1) == (m.MatchValue(s0[i]) ? 1 : 0));
        EXPECT(b[0] == 0x05 && b[1] == 0x21 && b[2] == 0x27 && b[3] == 0x01);  // i64_lt_res_0
    }
    // --- filter tree: Match (fused) == MatchAnd/MatchOr walk (reference data flow)
    {
        auto root = filter::Node::New();
        auto orn = filter::Node::New();
        orn->SetOr(true);
        orn->AddLeaf(filter::NewFilter<int64_t>("a", 0, KNOX_EQ, filter::Value::Of<int64_t>(5)));
        orn->AddLeaf(filter::NewFilter<double>("b", 1, KNOX_GT, filter::Value::Of<double>(40.0)));
        root->AddLeaf(filter::NewFilter<int64_t>("a", 0, KNOX_NE, filter::Value::Of<int64_t>(113)));
        root->AddChild(orn);
        bitset::Bitset fused(ctx, pkg.Len()), walked(ctx, pkg.Len());
        filter::Match(ctx, *root, pkg, fused);
        filter::MatchTree(ctx, *root, pkg, walked);
        EXPECT(fused.Bytes() == walked.Bytes());
        EXPECT(fused.Count() == walked.Count());
        int want = 0;
        for (int i = 0; i < 32; i++) want += (s0[i] != 113 && (s0[i] == 5 || f0[i] > 40.0)) ? 1 : 0;
        EXPECT(fused.Count() == want);

        // --- PhysicalFilter.Process attaches the selection vector (operator/filter.go:29-37)
        op::PhysicalFilter pf(ctx, root);
        EXPECT(pf.Process(pkg) == op::ResultOK);
        EXPECT(!pkg.SelectsAll());
        EXPECT((int)pkg.Selected().size() == want);
        for (uint32_t r : pkg.Selected()) EXPECT(s0[r] != 113 && (s0[r] == 5 || f0[r] > 40.0));

        // --- reducers over the same tree
        auto sum = reducer::Aggregate<int64_t>(ctx, *root, pkg, KNOX_R_SUM, 0);
        int64_t wsum = 0;
        for (uint32_t r : pkg.Selected()) wsum += s0[r];
        EXPECT(sum.seen && sum.value == wsum);
        auto mx = reducer::Aggregate<double>(ctx, *root, pkg, KNOX_R_MAX, 1);
        EXPECT(mx.seen && mx.value == 42.0 && mx.row == 28);
        // VarReducer / StdReducer (reducer.go:352-429): m2 / (n - 1) over the selected rows
        double m = 0, m2 = 0;
        for (uint32_t r : pkg.Selected()) m += f0[r];
        m /= (double)pkg.Selected().size();
        for (uint32_t r : pkg.Selected()) m2 += (f0[r] - m) * (f0[r] - m);
        const double wvar = m2 / (double)(pkg.Selected().size() - 1);
        auto var = reducer::Aggregate<double>(ctx, *root, pkg, KNOX_R_VAR, 1);
        auto sd = reducer::Aggregate<double>(ctx, *root, pkg, KNOX_R_STD, 1);
        EXPECT(var.seen && std::fabs(var.value - wvar) <= 1e-12 * wvar);
        EXPECT(sd.seen && std::fabs(sd.value - std::sqrt(wvar)) <= 1e-12 * std::sqrt(wvar));
        auto ivar = reducer::Aggregate<int64_t>(ctx, *root, pkg, KNOX_R_VAR, 0);
        EXPECT(ivar.seen && ivar.value == (int64_t)wvar);  // T(m2/(n-1)) truncates
    }
    // --- Bitset method semantics (bitset.go:400-540): cnt short-cuts, flags, size mismatch no-op
    {
        bitset::Bitset a(ctx, 100), b(ctx, 100), c(ctx, 64);
        a.One();
        EXPECT(a.All() && a.Count() == 100);
        auto f = a.AndFlag(b);  // r.cnt == 0 -> Zero, (false,false)
        EXPECT(!f.first && !f.second && a.None());
        b.One();
        a.Or(b);
        EXPECT(a.All());
        a.Neg();
        EXPECT(a.None());
        a.And(c);  // size mismatch: no-op
        EXPECT(a.Count() == 0);
        f = a.OrFlag(b);
        EXPECT(f.first && f.second);
        std::vector<uint8_t> pat(13, 0xfa);
        pat[12] &= 0x0f;
        b.SetBytes(pat);
        EXPECT(b.Count() == 12 * 6 + 2);
        auto idx = b.Indexes();
        EXPECT((int)idx.size() == b.Count() && idx[0] == 1 && idx[1] == 3);
    }
    // --- an all-true filter selects all rows: WithSelection(nil)
    {
        auto root = filter::Node::New();
        root->AddLeaf(filter::NewFilter<int64_t>("a", 0, KNOX_GE, filter::Value::Of<int64_t>(0)));
        op::PhysicalFilter pf(ctx, root);
        EXPECT(pf.Process(pkg) == op::ResultOK && pkg.SelectsAll());
    }
    std::printf(fails ? "FAILED: %d\n" : "host mirror ok\n", fails);
    return fails ? 1 : 0;
}
