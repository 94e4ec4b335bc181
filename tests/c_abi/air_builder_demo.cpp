// The C++ AIR recorder (include/aero_air_builder.hpp) on two systems that tests/ also writes with the Python builder:
//   air_builder_demo fib <width> <aux_width> <aux_rands> <aux_degree>      = aero_amd.air.fib_air(width, (A, R, D))
//   air_builder_demo v2 <log_n> <seq_stride>                                = tests/air_examples.py: v2_air(log_n, seq_stride)
// Prints the program as hex; tests/test_air_cpu.py checks it byte for byte against the Python builder and loads it with aero_air_load.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/aero_air_builder.hpp"

using aero_air_builder::Builder;
using aero_air_builder::Expr;
static const uint64_t P = aero_air_builder::P;

static std::vector<uint8_t> fib(uint32_t width, uint32_t A, uint32_t R, uint32_t D) {
    Builder b(width, A, A ? R : 0, width / 2);
    for (uint32_t k = 0; k < width / 2; k++) {
        Expr a = b.main(2 * k), bb = b.main(2 * k + 1), na = b.main_next(2 * k), nb = b.main_next(2 * k + 1);
        b.transition(na - (a + bb), 1);
        b.transition(nb - (bb + na), 1);
    }
    for (uint32_t c = 0; c < width; c++) b.assert_single(c, 0, (uint64_t)(1 + c));
    for (uint32_t k = 0; k < width / 2; k++) b.assert_single(2 * k + 1, -1, b.pub(k));
    for (uint32_t c = 0; c < A; c++) {
        Expr f = (b.rand(c % R) + b.main(c % width)).pow(D - 1);
        b.aux_transition(b.aux_next(c) - b.aux(c) * f, D);
        b.aux_assert_single(c, 0, (uint64_t)1);
        b.aux_builder(c, b.constant(1), f);
    }
    return b.to_bytes();
}

static uint64_t addp(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a + b) % P); }

static std::vector<uint8_t> v2(int log_n, uint32_t stride) {
    const size_t n = (size_t)1 << log_n;
    Builder b(4, 5, 4, 1);
    auto m = [&](uint32_t c) { return b.main(c); };
    auto mn = [&](uint32_t c) { return b.main_next(c); };
    auto a = [&](uint32_t c) { return b.aux(c); };
    auto an = [&](uint32_t c) { return b.aux_next(c); };
    auto r = [&](uint32_t i) { return b.rand(i); };
    // C++ leaves the evaluation order of a binary operator's operands to the compiler, Python evaluates left to right: the recorder
    // numbers nodes and constants in the order it meets them, so for BYTE equality with the Python builder the sub-expressions are
    // sequenced here (any order gives an equivalent program - the proofs do not depend on the numbering).
    { Expr s = m(0) + m(1); b.transition(mn(0) - s, 1); }
    { Expr s = m(1) + mn(0); b.transition(mn(1) - s, 1); }
    { Expr d = mn(2) - m(2); b.transition(d - 3, 1); }
    { Expr x1 = m(3) - 1; Expr p1 = m(3) * x1; Expr x2 = m(3) - 2; b.transition(p1 * x2, 3); }
    { Expr u = r(0) + m(2); Expr l = an(0) * u; Expr v = r(0) + m(0); Expr rr = a(0) * v; b.aux_transition(l - rr, 2); }
    { Expr d = an(1) - a(1); Expr u = r(1) + m(2); Expr l = d * u; b.aux_transition(l - m(3), 2); }
    { Expr u = r(2) + m(1); Expr p1 = a(2) * u; Expr d = an(2) - p1; Expr q = m(0) * m(3); b.aux_transition(d - q, 2); }
    b.aux_transition(an(3) - a(3), 1);
    { Expr sq = a(4) * a(4); Expr s1 = sq + m(0); Expr q = r(0) * a(1); Expr s2 = s1 + q; b.aux_transition(an(4) - s2, 2); }
    std::vector<uint64_t> t0(n), t1(n), t2(n);
    uint64_t x = 1, y = 2;
    for (size_t i = 0; i < n; i++) {
        t0[i] = x; t1[i] = y; t2[i] = (5 + 3 * (uint64_t)i) % P;
        const uint64_t nx = addp(x, y), ny = addp(y, nx);
        x = nx; y = ny;
    }
    b.assert_single(0, 0, (uint64_t)1);
    b.assert_single(1, 0, (uint64_t)2);
    b.assert_single(1, -1, b.pub(0));
    std::vector<uint64_t> seq;
    for (size_t i = 0; i < n / stride; i++) seq.push_back(t2[1 + stride * i]);
    b.assert_sequence(2, 1, stride, seq);
    b.assert_sequence(0, 0, (uint32_t)(n / 2), {t0[0], t0[n / 2]});
    b.aux_assert_single(0, 0, (uint64_t)1);
    b.aux_assert_single(1, 0, (uint64_t)0);
    b.aux_assert_single(2, 0, b.rand(3));
    b.aux_assert_sequence(3, 1, 2, std::vector<uint64_t>(n / 2, 7));
    { Expr nu = r(0) + m(0); Expr de = r(0) + m(2); b.aux_builder(0, b.constant(1), nu, de); }
    { Expr c0 = b.constant(0); Expr c1 = b.constant(1); Expr de = r(1) + m(2); b.aux_builder(1, c0, c1, Expr(), m(3), de); }
    { Expr nu = r(2) + m(1); Expr ad = m(0) * m(3); b.aux_builder(2, b.rand(3), nu, Expr(), ad); }
    { Expr c7 = b.constant(7); Expr c1 = b.constant(1); b.aux_builder(3, c7, c1); }
    b.aux_assert_single(4, 0, (uint64_t)3);
    { Expr sq = a(4) * a(4); Expr s1 = sq + m(0); Expr q = r(0) * a(1); b.aux_builder_general(4, b.constant(3), s1 + q); }
    return b.to_bytes();
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    std::vector<uint8_t> prog;
    if (!strcmp(argv[1], "fib") && argc == 6) prog = fib(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]));
    else if (!strcmp(argv[1], "v2") && argc == 4) prog = v2(atoi(argv[2]), atoi(argv[3]));
    else return 2;
    for (uint8_t v : prog) printf("%02x", v);
    printf("\n");
    return 0;
}
