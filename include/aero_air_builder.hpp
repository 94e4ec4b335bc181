// aero_air_builder.hpp — header-only recorder that turns an AIR written as C++ expressions into the AEROAIR bytes aero_air_load takes
// (format: include/aero_air.h). The C++ counterpart of aero_amd/air.py, and the shape of what a Rust host does with a symbolic
// `FieldElement`: winter-air's `Air::evaluate_transition<E>(frame, periodic_values, result)` is generic over the element type, so
// calling it ONCE with an element whose arithmetic appends nodes instead of computing records the constraint system the reference
// instantiates at aero-sdk/miden-wasm/src/constraints_worker.rs:32-43 (`ProcessorAir::new` -> `ConstraintEvaluator::new`).
//
//     aero_air_builder::Builder b(/*main*/ 2, /*aux*/ 0, /*rands*/ 0, /*public inputs*/ 1);
//     auto a = b.main(0), bb = b.main(1), na = b.main_next(0), nb = b.main_next(1);
//     b.transition(na - (a + bb), 1);
//     b.transition(nb - (bb + na), 1);
//     b.assert_single(0, 0, 1); b.assert_single(1, 0, 2); b.assert_single(1, -1, b.pub(0));
//     std::vector<uint8_t> program = b.to_bytes();          // -> aero_air_load(program.data(), program.size(), ...)
//
// Same hash-consing rules as the Python builder (a node per distinct (op, a, b), commutative operands ordered, constants by value):
// the same construction order gives the same bytes (tests/test_air_cpu.py compares them). Version 1 is written unless the AIR uses a
// sequence assertion or an affine / general auxiliary builder, which make it version 2. No dependency on the library: plain C++17.
#pragma once
#include <cstdint>
#include <cstring>
#include <map>
#include <stdexcept>
#include <tuple>
#include <vector>

namespace aero_air_builder {

constexpr uint64_t P = 0xFFFFFFFF00000001ull;
constexpr uint32_t NONE = 0xFFFFFFFFu, GENERAL = 0xFFFFFFFEu;
enum Kind : uint32_t { NODE = 0, MAIN_CUR, MAIN_NXT, AUX_CUR, AUX_NXT, PERIODIC, CONST, PUB, RAND, SEQ };
enum Op : uint32_t { ADD = 1, SUB = 2, MUL = 3 };

class Builder;

// An operand reference (kind << 24 | index) bound to its builder; arithmetic appends nodes.
class Expr {
public:
    Expr() = default;
    Expr(Builder* b, uint32_t ref) : b_(b), ref_(ref) {}
    uint32_t ref() const { return ref_; }
    Builder* builder() const { return b_; }
    Expr pow(unsigned e) const;
private:
    Builder* b_ = nullptr;
    uint32_t ref_ = NONE;
};

class Builder {
public:
    Builder(uint32_t main_width, uint32_t aux_width = 0, uint32_t aux_rands = 0, uint32_t num_pub = 0, uint32_t exemptions = 1)
        : W(main_width), A(aux_width), R(aux_rands), num_pub_(num_pub), exemptions_(exemptions) {
        if (W < 1 || W > 255 || A > 255 - W || (A == 0) != (R == 0) || R > 255) throw std::invalid_argument("aero_air_builder::Builder: bad shape");
        builders_.assign(A, BuilderRec{});
    }
    // ---- operands
    Expr main(uint32_t c) { check(c < W); return ref(MAIN_CUR, c); }
    Expr main_next(uint32_t c) { check(c < W); return ref(MAIN_NXT, c); }
    Expr aux(uint32_t c) { check(c < A); return ref(AUX_CUR, c); }
    Expr aux_next(uint32_t c) { check(c < A); return ref(AUX_NXT, c); }
    Expr pub(uint32_t i) { check(i < num_pub_); return ref(PUB, i); }
    Expr rand(uint32_t i) { check(i < R); return ref(RAND, i); }
    Expr constant(uint64_t v) {
        v %= P;
        auto it = const_idx_.find(v);
        if (it == const_idx_.end()) { it = const_idx_.emplace(v, (uint32_t)consts_.size()).first; consts_.push_back(v); }
        return ref(CONST, it->second);
    }
    // `get_periodic_column_values`: one cycle of the column (length a power of two >= 2)
    Expr periodic(const std::vector<uint64_t>& cycle) {
        check(cycle.size() >= 2 && (cycle.size() & (cycle.size() - 1)) == 0);
        periodics_.push_back(reduced(cycle));
        return ref(PERIODIC, (uint32_t)periodics_.size() - 1);
    }
    Expr node(uint32_t op, Expr a, Expr b) {
        uint32_t x = a.ref(), y = b.ref();
        if ((op == ADD || op == MUL) && y < x) std::swap(x, y);
        auto key = std::make_tuple(op, x, y);
        auto it = node_idx_.find(key);
        if (it == node_idx_.end()) { it = node_idx_.emplace(key, (uint32_t)nodes_.size()).first; nodes_.push_back({op, x, y}); }
        return ref(NODE, it->second);
    }
    // ---- constraints: `TransitionConstraintDegree::with_cycles(degree, cycles)`
    void transition(Expr e, uint32_t degree, std::vector<uint32_t> cycles = {}) { main_trans_.push_back({e.ref(), degree, std::move(cycles)}); }
    void aux_transition(Expr e, uint32_t degree, std::vector<uint32_t> cycles = {}) { aux_trans_.push_back({e.ref(), degree, std::move(cycles)}); }
    // ---- assertions: `Assertion::single / periodic / sequence`; step < 0 counts from the end (-1 = last row)
    void assert_single(uint32_t col, int32_t step, Expr v) { main_asserts_.push_back({col, step, 0, v.ref()}); }
    void assert_single(uint32_t col, int32_t step, uint64_t v) { assert_single(col, step, constant(v)); }
    void assert_periodic(uint32_t col, int32_t first, uint32_t stride, Expr v) { check_stride(first, stride); main_asserts_.push_back({col, first, stride, v.ref()}); }
    void assert_sequence(uint32_t col, int32_t first, uint32_t stride, const std::vector<uint64_t>& values) { main_asserts_.push_back({col, first, stride, sequence(first, stride, values)}); }
    void aux_assert_single(uint32_t col, int32_t step, Expr v) { aux_asserts_.push_back({col, step, 0, v.ref()}); }
    void aux_assert_single(uint32_t col, int32_t step, uint64_t v) { aux_assert_single(col, step, constant(v)); }
    void aux_assert_periodic(uint32_t col, int32_t first, uint32_t stride, Expr v) { check_stride(first, stride); aux_asserts_.push_back({col, first, stride, v.ref()}); }
    void aux_assert_sequence(uint32_t col, int32_t first, uint32_t stride, const std::vector<uint64_t>& values) { aux_asserts_.push_back({col, first, stride, sequence(first, stride, values)}); }
    // ---- how the prover builds the auxiliary columns (`Trace::build_aux_segment`):
    //      column(0) = init, column(i+1) = column(i) * num / den + add / add_den, every term on (row i, row i+1) of the main segment
    void aux_builder(uint32_t col, Expr init, Expr num, Expr den = Expr(), Expr add = Expr(), Expr add_den = Expr()) {
        check(col < A && (add.ref() != NONE || add_den.ref() == NONE));
        builders_[col] = BuilderRec{true, init.ref(), num.ref(), den.ref(), add.ref(), add_den.ref()};
    }
    //      column(i+1) = next evaluated on (main row i, main row i+1, CURRENT row of the auxiliary columns up to `col`): any recurrence
    void aux_builder_general(uint32_t col, Expr init, Expr next) { check(col < A); builders_[col] = BuilderRec{true, init.ref(), next.ref(), GENERAL, NONE, NONE}; }

    std::vector<uint8_t> to_bytes() const {
        uint32_t nb = 0;
        for (auto& r : builders_) nb += r.set;
        if (nb != 0 && nb != A) throw std::logic_error("aero_air_builder::Builder: one aux builder per auxiliary column, or none");
        bool v2 = !sequences_.empty();
        for (auto& r : builders_) v2 |= r.set && (r.add_num != NONE || r.den == GENERAL);
        std::vector<uint8_t> out;
        const char magic[8] = {'A', 'E', 'R', 'O', 'A', 'I', 'R', (char)(v2 ? 2 : 1)};
        out.insert(out.end(), magic, magic + 8);
        const uint32_t head[16] = {W, A, R, num_pub_, exemptions_, (uint32_t)consts_.size(), (uint32_t)periodics_.size(), (uint32_t)nodes_.size(),
                                   (uint32_t)main_trans_.size(), (uint32_t)aux_trans_.size(), (uint32_t)main_asserts_.size(), (uint32_t)aux_asserts_.size(),
                                   nb, (uint32_t)sequences_.size(), 0, 0};
        for (uint32_t v : head) u32(out, v);
        for (uint64_t v : consts_) u64(out, v);
        for (auto& cyc : periodics_) { u32(out, (uint32_t)cyc.size()); for (uint64_t v : cyc) u64(out, v); }
        for (auto& seq : sequences_) { u32(out, (uint32_t)seq.size()); for (uint64_t v : seq) u64(out, v); }
        for (auto& n : nodes_) { u32(out, n.op); u32(out, n.a); u32(out, n.b); }
        for (auto* list : {&main_trans_, &aux_trans_})
            for (auto& t : *list) { u32(out, t.root); u32(out, t.degree); u32(out, (uint32_t)t.cycles.size()); for (uint32_t c : t.cycles) u32(out, c); }
        for (auto* list : {&main_asserts_, &aux_asserts_})
            for (auto& a : *list) { u32(out, a.col); u32(out, (uint32_t)a.first); u32(out, a.stride); u32(out, a.value); }
        if (nb)
            for (auto& r : builders_) {
                u32(out, r.init); u32(out, r.num); u32(out, r.den);
                if (v2) { u32(out, r.add_num); u32(out, r.add_den); }
            }
        return out;
    }

private:
    struct NodeRec { uint32_t op, a, b; };
    struct TransRec { uint32_t root, degree; std::vector<uint32_t> cycles; };
    struct AssertRec { uint32_t col; int32_t first; uint32_t stride, value; };
    struct BuilderRec { bool set = false; uint32_t init = NONE, num = NONE, den = NONE, add_num = NONE, add_den = NONE; };
    uint32_t W, A, R, num_pub_, exemptions_;
    std::vector<uint64_t> consts_;
    std::map<uint64_t, uint32_t> const_idx_;
    std::vector<std::vector<uint64_t>> periodics_, sequences_;
    std::vector<NodeRec> nodes_;
    std::map<std::tuple<uint32_t, uint32_t, uint32_t>, uint32_t> node_idx_;
    std::vector<TransRec> main_trans_, aux_trans_;
    std::vector<AssertRec> main_asserts_, aux_asserts_;
    std::vector<BuilderRec> builders_;

    Expr ref(uint32_t kind, uint32_t idx) { return Expr(this, (kind << 24) | idx); }
    static void check(bool ok) { if (!ok) throw std::out_of_range("aero_air_builder::Builder: operand or argument out of range"); }
    static void check_stride(int32_t first, uint32_t stride) { check(stride >= 2 && (stride & (stride - 1)) == 0 && first >= 0 && (uint32_t)first < stride); }
    static std::vector<uint64_t> reduced(const std::vector<uint64_t>& v) { std::vector<uint64_t> r(v); for (auto& x : r) x %= P; return r; }
    uint32_t sequence(int32_t first, uint32_t stride, const std::vector<uint64_t>& values) {
        check_stride(first, stride);
        check(values.size() >= 2 && (values.size() & (values.size() - 1)) == 0);
        sequences_.push_back(reduced(values));
        return (SEQ << 24) | ((uint32_t)sequences_.size() - 1);
    }
    static void u32(std::vector<uint8_t>& o, uint32_t v) { for (int i = 0; i < 4; i++) o.push_back((uint8_t)(v >> (8 * i))); }
    static void u64(std::vector<uint8_t>& o, uint64_t v) { for (int i = 0; i < 8; i++) o.push_back((uint8_t)(v >> (8 * i))); }
};

inline Expr lift(const Expr& like, uint64_t v) { return like.builder()->constant(v); }
inline Expr operator+(Expr a, Expr b) { return a.builder()->node(ADD, a, b); }
inline Expr operator-(Expr a, Expr b) { return a.builder()->node(SUB, a, b); }
inline Expr operator*(Expr a, Expr b) { return a.builder()->node(MUL, a, b); }
inline Expr operator+(Expr a, uint64_t b) { return a + lift(a, b); }
inline Expr operator-(Expr a, uint64_t b) { return a - lift(a, b); }
inline Expr operator*(Expr a, uint64_t b) { return a * lift(a, b); }
inline Expr operator+(uint64_t a, Expr b) { return lift(b, a) + b; }
inline Expr operator-(uint64_t a, Expr b) { return lift(b, a) - b; }
inline Expr operator*(uint64_t a, Expr b) { return lift(b, a) * b; }
inline Expr Expr::pow(unsigned e) const {          // square-and-multiply, as the Python builder's `**`
    if (e < 1) throw std::invalid_argument("aero_air_builder::Expr::pow: exponent >= 1");
    Expr r, base = *this;
    bool have = false;
    while (e) {
        if (e & 1) { r = have ? r * base : base; have = true; }
        e >>= 1;
        if (e) base = base * base;
    }
    return r;
}

}  // namespace aero_air_builder
