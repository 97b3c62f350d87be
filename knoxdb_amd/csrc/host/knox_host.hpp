// knox_host.hpp -- C++ host-side mirror of the reference's operator/plugin interface for the
// pack-scan path, implemented ONLY on top of the C ABI (include/knoxgpu.h).
//
// The reference host is Go; this image has no Go toolchain, so the layer a KnoxDB query would
// talk to is mirrored in C++ with the reference's names, argument meaning and error behaviour:
//
//   knox::bitset::Bitset            internal/bitset/bitset.go:30-540   (device resident, cached cnt)
//   knox::block::Block              internal/block/block.go:83-96      (materialized column vector in HBM)
//   knox::pack::Package             internal/pack/package.go:23        (N blocks + selection vector)
//   knox::filter::Matcher           internal/operator/filter/match.go:15-72
//   knox::filter::NumMatcherFactory internal/operator/filter/match_num.go:253-304
//   knox::filter::Filter / Node     internal/operator/filter/filter.go:22-30, node.go:29-80
//   knox::filter::Match / MatchSingle / MatchAnd / MatchOr   match_core.go:14-215
//   knox::op::PhysicalFilter        internal/operator/filter.go:16-58, operator.go:15-42
//   knox::reducer::Aggregate        Reducer.Value()/Time() of internal/reducer/reducer.go:129-349
//
// Header only; link with -lknoxgpu.  No computation happens here: every result comes from the
// device through the ABI.
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../../include/knoxgpu.h"

namespace knox {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &what) : std::runtime_error(what + ": " + knox_strerror(c)), code(c) {}
};
inline void check(int rc, const char *what) {
    if (rc != KNOX_OK) throw Error(rc, what);
}

// types.BlockType / types.FilterMode (internal/types/block.go:20-36, mode.go:13-27)
using BlockType = knox_dtype;
using FilterMode = knox_op;
template <typename T> struct BlockTypeOf;
template <> struct BlockTypeOf<int64_t> { static constexpr BlockType v = KNOX_INT64; };
template <> struct BlockTypeOf<int32_t> { static constexpr BlockType v = KNOX_INT32; };
template <> struct BlockTypeOf<int16_t> { static constexpr BlockType v = KNOX_INT16; };
template <> struct BlockTypeOf<int8_t> { static constexpr BlockType v = KNOX_INT8; };
template <> struct BlockTypeOf<uint64_t> { static constexpr BlockType v = KNOX_UINT64; };
template <> struct BlockTypeOf<uint32_t> { static constexpr BlockType v = KNOX_UINT32; };
template <> struct BlockTypeOf<uint16_t> { static constexpr BlockType v = KNOX_UINT16; };
template <> struct BlockTypeOf<uint8_t> { static constexpr BlockType v = KNOX_UINT8; };
template <> struct BlockTypeOf<double> { static constexpr BlockType v = KNOX_FLOAT64; };
template <> struct BlockTypeOf<float> { static constexpr BlockType v = KNOX_FLOAT32; };

class Context {
  public:
    explicit Context(int device = 0) { check(knox_ctx_open(device, &ctx_), "knox_ctx_open"); }
    ~Context() { if (ctx_) knox_ctx_close(ctx_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    knox_ctx *get() const { return ctx_; }

  private:
    knox_ctx *ctx_ = nullptr;
};

namespace bitset {

// bitset.Bitset: size, cached count (-1 = unknown), buffer on the device.
class Bitset {
  public:
    Bitset(Context &c, int size) : ctx_(&c), size_(size), cnt_(0) { check(knox_bits_alloc(c.get(), (size_t)size, &b_), "knox_bits_alloc"); }
    ~Bitset() { Close(); }
    Bitset(const Bitset &) = delete;
    Bitset &operator=(const Bitset &) = delete;
    void Close() { if (b_) { knox_bits_free(b_); b_ = nullptr; } }

    int Len() const { return size_; }
    knox_bits *handle() const { return b_; }
    void ResetCount(int n) { cnt_ = n; }                       // bitset.go:667
    int Count() {                                              // :286
        if (cnt_ < 0) { int64_t c = 0; check(knox_bits_popcount(ctx_->get(), b_, &c), "popcount"); cnt_ = (int)c; }
        return cnt_;
    }
    bool All() { return size_ > 0 && Count() == size_; }        // :306
    bool None() { return Count() == 0; }                       // :312 (count based here)
    bool Any() { return !None(); }
    Bitset &Zero() {                                           // :533
        if (size_ == 0 || cnt_ == 0) return *this;
        check(knox_bits_zero(ctx_->get(), b_), "zero"); cnt_ = 0; return *this;
    }
    Bitset &One() {                                            // :520
        if (size_ == 0) return *this;
        check(knox_bits_one(ctx_->get(), b_), "one"); cnt_ = size_; return *this;
    }
    Bitset &Neg() {                                            // :509
        if (size_ == 0) return *this;
        check(knox_bits_neg(ctx_->get(), b_), "neg");
        if (cnt_ >= 0) cnt_ = size_ - cnt_;
        return *this;
    }
    // And / AndNot / Or / Xor with the reference's cnt short-cuts; a size mismatch is a no-op (:400-518)
    Bitset &And(Bitset &r) {
        if (size_ == r.size_ && size_ > 0) {
            if (cnt_ == 0) return *this;
            if (r.cnt_ == 0) return Zero();
            check(knox_bits_and(ctx_->get(), b_, r.b_), "and"); cnt_ = -1;
        }
        return *this;
    }
    Bitset &AndNot(Bitset &r) {
        if (size_ == r.size_ && size_ > 0) {
            if (cnt_ == 0) return *this;
            check(knox_bits_andnot(ctx_->get(), b_, r.b_), "andnot"); cnt_ = -1;
        }
        return *this;
    }
    Bitset &Or(Bitset &r) {
        if (size_ == r.size_ && size_ > 0) {
            if (cnt_ == size_) return *this;
            if (r.cnt_ == r.size_) return One();
            check(knox_bits_or(ctx_->get(), b_, r.b_), "or"); cnt_ = -1;
        }
        return *this;
    }
    Bitset &Xor(Bitset &r) {
        if (size_ == r.size_ && size_ > 0) { check(knox_bits_xor(ctx_->get(), b_, r.b_), "xor"); cnt_ = -1; }
        return *this;
    }
    // AndFlag :410-445 -> (any, all)
    std::pair<bool, bool> AndFlag(Bitset &r) {
        if (size_ == 0) return {false, true};
        if (size_ != r.size_) return cnt_ == 0 ? std::make_pair(false, false) : cnt_ == size_ ? std::make_pair(true, true) : std::make_pair(true, false);
        if (cnt_ == 0) return {false, false};
        if (r.cnt_ == 0) { Zero(); return {false, false}; }
        int any = 0, all = 0;
        check(knox_bits_and_flag(ctx_->get(), b_, r.b_, &any, &all), "and_flag");
        cnt_ = -1;
        if (!any) cnt_ = 0; else if (all) cnt_ = size_;
        return {any != 0, all != 0};
    }
    std::pair<bool, bool> OrFlag(Bitset &r) {                  // :471-505
        if (size_ == 0) return {false, true};
        if (size_ != r.size_) return cnt_ == 0 ? std::make_pair(false, false) : cnt_ == size_ ? std::make_pair(true, true) : std::make_pair(true, false);
        if (cnt_ == size_) return {true, true};
        if (r.cnt_ == r.size_) { One(); return {true, true}; }
        int any = 0, all = 0;
        check(knox_bits_or_flag(ctx_->get(), b_, r.b_, &any, &all), "or_flag");
        cnt_ = -1;
        if (!any) cnt_ = 0; else if (all) cnt_ = size_;
        return {any != 0, all != 0};
    }
    Bitset &Copy(Bitset &r) { check(knox_bits_copy(ctx_->get(), b_, r.b_), "copy"); cnt_ = r.cnt_; return *this; }
    std::vector<uint8_t> Bytes() const {                       // :651 (host copy of the device buffer)
        std::vector<uint8_t> out(((size_t)size_ + 7) >> 3);
        check(knox_bits_download(ctx_->get(), b_, out.data()), "download");
        return out;
    }
    void SetBytes(const std::vector<uint8_t> &v) { check(knox_bits_upload(ctx_->get(), b_, v.data()), "upload"); cnt_ = -1; }
    std::vector<uint32_t> Indexes() {                          // iterator.go:269-290
        std::vector<uint32_t> out((size_t)Count() + 8);
        size_t n = 0;
        check(knox_bits_indexes(ctx_->get(), b_, out.data(), out.size(), &n), "indexes");
        out.resize(n);
        return out;
    }

  private:
    Context *ctx_;
    knox_bits *b_ = nullptr;
    int size_;
    int cnt_;
};

}  // namespace bitset

namespace block {

// A materialized block: one typed column vector resident in HBM (block.go:83-96).
class Block {
  public:
    template <typename T> Block(Context &c, const std::vector<T> &v, size_t pack_rows = 0) : ctx_(&c), type_(BlockTypeOf<T>::v) {
        check(knox_col_upload(c.get(), type_, v.data(), v.size(), pack_rows, &col_), "knox_col_upload");
    }
    ~Block() { if (col_) knox_col_free(col_); }
    Block(const Block &) = delete;
    Block &operator=(const Block &) = delete;
    BlockType Type() const { return type_; }
    int Len() const { return (int)knox_col_rows(col_); }
    bool IsMaterialized() const { return true; }
    knox_col *handle() const { return col_; }

  private:
    Context *ctx_;
    BlockType type_;
    knox_col *col_ = nullptr;
};

}  // namespace block

namespace pack {

// pack.Package: N blocks of equal length + optional selection vector (package.go:23, :85).
class Package {
  public:
    void AddBlock(std::shared_ptr<block::Block> b) { blocks_.push_back(std::move(b)); }
    block::Block &Block(int i) { return *blocks_.at((size_t)i); }
    int Len() const { return blocks_.empty() ? 0 : blocks_[0]->Len(); }
    int Cols() const { return (int)blocks_.size(); }
    // WithSelection(nil) = all rows selected (package.go:85)
    Package &WithSelection(std::vector<uint32_t> sel, bool all) { sel_ = std::move(sel); all_ = all; return *this; }
    bool SelectsAll() const { return all_; }
    const std::vector<uint32_t> &Selected() const { return sel_; }

  private:
    std::vector<std::shared_ptr<block::Block>> blocks_;
    std::vector<uint32_t> sel_;
    bool all_ = true;
};

}  // namespace pack

namespace filter {

// Value of a filter: one scalar or a RangeValue [from, to] (match_num.go:545-559), as raw bits.
struct Value {
    uint64_t a = 0, b = 0;
    template <typename T> static Value Of(T v) { Value x; std::memcpy(&x.a, &v, sizeof(T)); return x; }
    template <typename T> static Value Range(T from, T to) { Value x; std::memcpy(&x.a, &from, sizeof(T)); std::memcpy(&x.b, &to, sizeof(T)); return x; }
};

// Matcher (match.go:15-72): the vector part runs on the device, the scalar parts on the host.
class Matcher {
  public:
    virtual ~Matcher() = default;
    virtual void WithValue(const Value &v) = 0;
    virtual int Weight() const = 0;
    virtual int Len() const = 0;
    // fills `res` (pre-sized to block.Len(), zeroed by the caller) and calls res.ResetCount(n);
    // `mask` may be ignored for compare modes, exactly as in the reference (match_num.go:348-355)
    virtual void MatchVector(Context &c, block::Block &b, bitset::Bitset &res, bitset::Bitset *mask) = 0;
};

template <typename T> class numMatcher : public Matcher {
  public:
    explicit numMatcher(FilterMode m) : mode_(m) {}
    void WithValue(const Value &v) override { val_ = v; }
    int Weight() const override { return mode_ == KNOX_RANGE ? 2 : 1; }
    int Len() const override { return mode_ == KNOX_RANGE ? 2 : 1; }
    FilterMode Mode() const { return mode_; }
    const Value &Val() const { return val_; }
    bool MatchValue(T v) const {  // numEqualMatcher.MatchValue etc. (match_num.go:329-566)
        T a, b;
        std::memcpy(&a, &val_.a, sizeof(T));
        std::memcpy(&b, &val_.b, sizeof(T));
        switch (mode_) {
        case KNOX_EQ: return v == a;
        case KNOX_NE: return v != a;
        case KNOX_GT: return v > a;
        case KNOX_GE: return v >= a;
        case KNOX_LT: return v < a;
        case KNOX_LE: return v <= a;
        case KNOX_RANGE: return a <= v && b >= v;
        default: return false;
        }
    }
    void MatchVector(Context &c, block::Block &b, bitset::Bitset &res, bitset::Bitset *) override {
        knox_pred p;
        std::memset(&p, 0, sizeof p);
        p.kind = KNOX_NODE_LEAF;
        p.col = 0;
        p.op = mode_;
        p.a = val_.a;
        p.b = val_.b;
        knox_col *cols[1] = {b.handle()};
        knox_scan_out out;
        check(knox_scan(c.get(), &p, 1, cols, 1, nullptr, 0, 0, nullptr, res.handle(), nullptr, nullptr, &out), "knox_scan");
        res.ResetCount((int)out.count);
    }

  private:
    FilterMode mode_;
    Value val_;
};

// NumMatcherFactory[T].New(mode) (match_num.go:257-304); IN/NIN/regexp are not on this path.
template <typename T> struct NumMatcherFactory {
    std::shared_ptr<Matcher> New(FilterMode m) const {
        switch (m) {
        case KNOX_EQ: case KNOX_NE: case KNOX_GT: case KNOX_GE: case KNOX_LT: case KNOX_LE: case KNOX_RANGE:
            return std::make_shared<numMatcher<T>>(m);
        default: throw Error(KNOX_E_UNSUPPORTED, "NumMatcherFactory.New");
        }
    }
};

// Filter (filter.go:22-30)
struct Filter {
    std::string Name;
    BlockType Type;
    FilterMode Mode;
    int Index;
    std::shared_ptr<Matcher> Matcher_;
    Value Val;
};
template <typename T> inline std::shared_ptr<Filter> NewFilter(const std::string &name, int idx, FilterMode mode, const Value &v) {
    auto f = std::make_shared<Filter>();
    f->Name = name; f->Type = BlockTypeOf<T>::v; f->Mode = mode; f->Index = idx; f->Val = v;
    f->Matcher_ = NumMatcherFactory<T>().New(mode);
    f->Matcher_->WithValue(v);
    return f;
}

// Node (node.go:29-80): root is always an AND node.
struct Node {
    std::vector<std::shared_ptr<Node>> Children;
    std::shared_ptr<Filter> Filter_;
    bool OrKind = false;
    static std::shared_ptr<Node> New() { return std::make_shared<Node>(); }
    bool IsLeaf() const { return Filter_ && Children.empty(); }
    Node &SetFilter(std::shared_ptr<Filter> f) { Filter_ = std::move(f); Children.clear(); return *this; }
    Node &AddChild(std::shared_ptr<Node> c) { Children.push_back(std::move(c)); return *this; }
    Node &AddLeaf(std::shared_ptr<Filter> f) { auto n = New(); n->SetFilter(std::move(f)); Children.push_back(n); return *this; }
    Node &SetOr(bool b) { OrKind = b; return *this; }
};

inline void flatten(const Node &n, std::vector<knox_pred> &out) {
    knox_pred p;
    std::memset(&p, 0, sizeof p);
    if (n.IsLeaf()) {
        p.kind = KNOX_NODE_LEAF; p.col = n.Filter_->Index; p.op = n.Filter_->Mode; p.a = n.Filter_->Val.a; p.b = n.Filter_->Val.b;
        out.push_back(p);
        return;
    }
    for (auto &c : n.Children) flatten(*c, out);
    p.kind = n.OrKind ? KNOX_NODE_OR : KNOX_NODE_AND;
    p.nchild = (int)n.Children.size();
    out.push_back(p);
}

// MatchSingle (match_core.go:32-43)
inline bitset::Bitset &MatchSingle(Context &c, Filter &f, pack::Package &pkg, bitset::Bitset &bits, bitset::Bitset *mask) {
    f.Matcher_->MatchVector(c, pkg.Block(f.Index), bits, mask);
    return bits;
}

// Match (match_core.go:14-26): ONE fused device pass over the whole tree.  Early exits of
// MatchAnd / MatchOr change work, never results.
inline bitset::Bitset &Match(Context &c, const Node &n, pack::Package &pkg, bitset::Bitset &bits) {
    std::vector<knox_pred> preds;
    flatten(n, preds);
    std::vector<knox_col *> cols;
    for (int i = 0; i < pkg.Cols(); i++) cols.push_back(pkg.Block(i).handle());
    knox_scan_out out;
    check(knox_scan(c.get(), preds.data(), (int)preds.size(), cols.data(), (int)cols.size(), nullptr, 0, 0, nullptr, bits.handle(), nullptr,
                    nullptr, &out), "knox_scan");
    bits.ResetCount((int)out.count);
    return bits;
}

// MatchAnd / MatchOr exactly as the reference walks them (match_core.go:45-215): leaf by leaf into a
// scratch bitset, AndFlag / Or merges, early exits.  Same final bitset as Match.
inline bitset::Bitset &MatchTree(Context &c, const Node &n, pack::Package &pkg, bitset::Bitset &bits);
inline bitset::Bitset &MatchAnd(Context &c, const Node &n, pack::Package &pkg, bitset::Bitset &bits) {
    bits.One();
    for (auto &node : n.Children) {
        bitset::Bitset scratch(c, pkg.Len());
        if (!node->IsLeaf()) MatchTree(c, *node, pkg, scratch);
        else MatchSingle(c, *node->Filter_, pkg, scratch, &bits);
        auto flags = bits.AndFlag(scratch);
        if (!flags.first) break;
    }
    return bits;
}
inline bitset::Bitset &MatchOr(Context &c, const Node &n, pack::Package &pkg, bitset::Bitset &bits) {
    bits.Zero();
    for (size_t i = 0; i < n.Children.size(); i++) {
        auto &node = n.Children[i];
        bitset::Bitset scratch(c, pkg.Len());
        if (!node->IsLeaf()) MatchTree(c, *node, pkg, scratch);
        else MatchSingle(c, *node->Filter_, pkg, scratch, nullptr);
        bits.Or(scratch);
        if (i + 1 < n.Children.size() && bits.All()) break;
    }
    return bits;
}
inline bitset::Bitset &MatchTree(Context &c, const Node &n, pack::Package &pkg, bitset::Bitset &bits) {
    if (n.IsLeaf()) return MatchSingle(c, *n.Filter_, pkg, bits, nullptr);
    return n.OrKind ? MatchOr(c, n, pkg, bits) : MatchAnd(c, n, pkg, bits);
}

}  // namespace filter

namespace op {

// operator.Result (operator.go:15-21)
enum Result { ResultOK = 0, ResultMore, ResultDone, ResultError };

// PhysicalFilter (operator/filter.go:16-58): Process attaches the selection vector.
class PhysicalFilter {
  public:
    PhysicalFilter(Context &c, std::shared_ptr<filter::Node> node) : ctx_(&c), node_(std::move(node)) {}
    Result Process(pack::Package &src) {
        try {
            bitset::Bitset bits(*ctx_, src.Len());
            filter::Match(*ctx_, *node_, src, bits);
            if (bits.All()) src.WithSelection({}, true);
            else src.WithSelection(bits.Indexes(), false);
            return ResultOK;
        } catch (const Error &e) {
            err_ = e.what();
            return ResultError;
        }
    }
    int Finalize() { return 0; }
    const std::string &Err() const { return err_; }
    void Close() { node_.reset(); }

  private:
    Context *ctx_;
    std::shared_ptr<filter::Node> node_;
    std::string err_;
};

}  // namespace op

namespace reducer {

// Reducer.Value()/Time() of one reducer function over the rows a filter tree selects
// (reducer.go:129-349), computed in the same device pass as the filter.
template <typename T> struct Result {
    T value{};
    bool seen = false;
    int64_t row = -1;  // the row whose timestamp Time() would report
};
template <typename T>
inline Result<T> Aggregate(Context &c, const filter::Node &n, pack::Package &pkg, knox_reducer func, int col) {
    std::vector<knox_pred> preds;
    filter::flatten(n, preds);
    std::vector<knox_col *> cols;
    for (int i = 0; i < pkg.Cols(); i++) cols.push_back(pkg.Block(i).handle());
    knox_agg agg{(int32_t)func, (int32_t)col};
    knox_agg_state st;
    knox_scan_out out;
    check(knox_scan(c.get(), preds.data(), (int)preds.size(), cols.data(), (int)cols.size(), &agg, 1, 0, nullptr, nullptr, nullptr, &st, &out),
          "knox_scan");
    Result<T> r;
    int seen = 0;
    check(knox_agg_value(&st, func, &r.value, &seen, &r.row), "knox_agg_value");
    r.seen = seen != 0;
    return r;
}

}  // namespace reducer
}  // namespace knox
