"""Example program AIRs for the AIR-as-data tests: builders (aero_amd.air) together with traces that satisfy them.

`synth_vm(log_n, pairs, aux)` is a synthetic AIR in the SHAPE of a VM's (Miden's ProcessorAir is absent from the reference
mount: SURVEY.md section 0): a clock, a binary counter, periodic selector / round-constant columns of cycle 8, power-map state
columns of degree 2..7 gated by the selector, degree 5..8 accumulators, Fibonacci pairs, a permutation argument with numerator
and denominator (auxiliary running product that must return to 1), two transition exemptions, single assertions at the first,
last and an interior step, periodic assertions. More than 49 transition constraints in >= 8 degree groups, degrees 1..8.
TEST INFRASTRUCTURE: nothing in aero_amd/ imports this."""
import numpy as np

from aero_amd import air as A

P = A.P
SEL = [1, 1, 1, 0, 1, 0, 0, 1]


def rc(j):
    return [(0x9E3779B97F4A7C15 * (8 * j + i + 1) + 12345) % P for i in range(8)]


def synth_vm(log_n, pairs=2, aux=3, rands=4):
    """Returns (AirBuilder, trace (W, n) uint64, pub list). Columns: 0 clk | 1-4 bits | 5 m8 | 6-13 s_0..7 | 14-17 acc (deg 5..8) |
    18 b (s_0 shifted by one step, cyclically over the first n-1 rows) | 19 b2 (s_5 likewise) | 20.. Fibonacci pairs."""
    n = 1 << log_n
    W = 20 + 2 * pairs
    R = rands if aux else 0
    b = A.AirBuilder(W, aux, R, num_pub=pairs + 1, exemptions=2)
    CLK, BIT, M8, S, ACC, B1, B2, FIB = 0, 1, 5, 6, 14, 18, 19, 20
    cur, nxt = b.main, b.main_next
    sel = b.periodic(SEL)
    rcs = [b.periodic(rc(j)) for j in range(8)]
    deg_s = [2, 3, 4, 5, 6, 7, 2, 3]
    deg_acc = [5, 6, 7, 8]
    # ---- constraints
    b.transition(nxt(CLK) - cur(CLK) - 1, 1)
    bits = [cur(BIT + i) for i in range(4)]
    for i in range(4):
        b.transition(bits[i] * (bits[i] - 1), 2)
    carry = None
    for i in range(4):                       # binary counter: bit_i' = bit_i XOR (bit_0 ... bit_(i-1))
        t = b.const(1) if carry is None else carry
        b.transition(nxt(BIT + i) - (bits[i] + t - 2 * bits[i] * t), i + 1 if i else 1)
        carry = bits[i] if carry is None else carry * bits[i]
    b.transition(cur(M8) - (bits[0] + 2 * bits[1] + 4 * bits[2]), 1)
    s = [cur(S + j) for j in range(8)]
    for j in range(8):                       # s_j' = sel (s_j^d + rc_j) + (1 - sel)(s_j + s_(j+1))
        b.transition(nxt(S + j) - (sel * (s[j] ** deg_s[j] + rcs[j]) + (1 - sel) * (s[j] + s[(j + 1) % 8])), deg_s[j], cycles=[8])
    for k in range(4):                       # acc_k' = acc_k^d + s_k
        b.transition(nxt(ACC + k) - (cur(ACC + k) ** deg_acc[k] + s[k]), deg_acc[k])
    b.transition(cur(B1) - nxt(S), 1)
    b.transition(cur(B2) - nxt(S + 5), 1)
    for k in range(pairs):
        a_, b_, na, nb = cur(FIB + 2 * k), cur(FIB + 2 * k + 1), nxt(FIB + 2 * k), nxt(FIB + 2 * k + 1)
        b.transition(na - (a_ + b_), 1)
        b.transition(nb - (b_ + na), 1)
    # ---- trace
    t = [[0] * n for _ in range(W)]
    for j in range(8):
        t[S + j][0] = 3 + j
    for k in range(4):
        t[ACC + k][0] = 11 + k
    for k in range(pairs):
        t[FIB + 2 * k][0], t[FIB + 2 * k + 1][0] = 1 + 2 * k, 2 + 2 * k
    for i in range(n):
        t[CLK][i] = i
        for q in range(4):
            t[BIT + q][i] = (i >> q) & 1
        t[M8][i] = i & 7
        if i + 1 < n:
            sl = SEL[i % 8]
            for j in range(8):
                v = t[S + j][i]
                t[S + j][i + 1] = (pow(v, deg_s[j], P) + rc(j)[i % 8]) % P if sl else (v + t[S + (j + 1) % 8][i]) % P
            for k in range(4):
                t[ACC + k][i + 1] = (pow(t[ACC + k][i], deg_acc[k], P) + t[S + k][i]) % P
            for k in range(pairs):
                a_, b_ = t[FIB + 2 * k][i], t[FIB + 2 * k + 1][i]
                t[FIB + 2 * k][i + 1] = (a_ + b_) % P
                t[FIB + 2 * k + 1][i + 1] = (b_ + a_ + b_) % P
    for i in range(n - 1):
        t[B1][i] = t[S][(i + 1) % (n - 1)]
        t[B2][i] = t[S + 5][(i + 1) % (n - 1)]
    pub = [t[FIB + 2 * k + 1][n - 1] for k in range(pairs)] + [t[ACC + 3][n - 1]]
    # ---- assertions
    b.assert_single(CLK, 0, 0)
    b.assert_single(CLK, n // 2, n // 2)                 # an interior step
    b.assert_single(B1, -2, t[S][0])                     # where the cyclic shift wraps (covered by the second exemption)
    b.assert_single(B2, -2, t[S + 5][0])
    for j in range(8):
        b.assert_single(S + j, 0, 3 + j)
    for k in range(4):
        b.assert_single(ACC + k, 0, 11 + k)
    b.assert_single(ACC + 3, -1, b.pub(pairs))
    for k in range(pairs):
        b.assert_single(FIB + 2 * k, 0, 1 + 2 * k)
        b.assert_single(FIB + 2 * k + 1, 0, 2 + 2 * k)
        b.assert_single(FIB + 2 * k + 1, -1, b.pub(k))
    b.assert_periodic(M8, 0, 8, 0)
    b.assert_periodic(M8, 3, 8, 3)
    b.assert_periodic(BIT, 1, 2, 1)
    # ---- auxiliary segment
    if aux:
        r = [b.rand(i) for i in range(R)]
        # p0: permutation argument s_0 <-> b over the first n-1 rows: p' (r0 + b) = p (r0 + s_0), returns to 1
        b.aux_transition(b.aux_next(0) * (r[0] + cur(B1)) - b.aux(0) * (r[0] + s[0]), 2)
        b.aux_builder(0, 1, r[0] + s[0], r[0] + cur(B1))
        b.aux_assert_single(0, 0, 1)
        b.aux_assert_single(0, -1, 1)
        for c in range(1, aux):
            e = 1 + c % 3
            col = s[(c + 1) % 8] + r[(c + 1) % R] * cur(CLK)
            if c == 1:                        # a second argument with a denominator, tuples compressed with r1
                num, den = r[1 % R] + s[5] + r[2 % R] * cur(CLK), r[1 % R] + cur(B2) + r[2 % R] * cur(CLK)
                b.aux_transition(b.aux_next(c) * den - b.aux(c) * num, 2)
                b.aux_builder(c, r[0] * r[0] + 1, num, den)
                b.aux_assert_single(c, 0, r[0] * r[0] + 1)      # a boundary value that depends on the random elements
                continue
            f = (r[c % R] + col) ** e
            if c % 2 == 0:                    # gated by the periodic selector
                f = sel * f + (1 - sel)
                b.aux_transition(b.aux_next(c) - b.aux(c) * f, e + 1, cycles=[8])
            else:
                b.aux_transition(b.aux_next(c) - b.aux(c) * f, e + 1)
            b.aux_builder(c, 1, f)
            b.aux_assert_single(c, 0, 1)
    return b, np.array(t, dtype=np.uint64), pub


def tiny_no_assertion_groups(log_n):
    """One column doubling every step, one assertion: the smallest program (1 column, 1 constraint, 1 divisor group)."""
    n = 1 << log_n
    b = A.AirBuilder(1, num_pub=1)
    b.transition(b.main_next(0) - 2 * b.main(0), 1)
    b.assert_single(0, 0, b.pub(0))
    col = [pow(2, i, P) * 5 % P for i in range(n)]
    return b, np.array([col], dtype=np.uint64), [5]


def v2_air(log_n, seq_stride=4):
    """An AIR that needs AEROAIR version 2: `Assertion::sequence` on main and auxiliary columns and affine auxiliary builders.
    Main: 0,1 Fibonacci pair | 2 counter 5 + 3 i | 3 multiplicity (i mod 3). Auxiliary (4 random elements):
      0  running product with denominator   p' = p (r0 + m0) / (r0 + m2)
      1  running SUM (log-derivative shape) s' = s + m3 / (r1 + m2)
      2  mixed affine                       u' = u (r2 + m1) + m0 m3
      3  constant 7                         c' = c
      4  general (cannot be scanned)        g' = g g + m0 + r0 a1      (reads its own and an earlier auxiliary column)
    Assertions: singles, a sequence on the counter (every seq_stride-th step from step 1), a two-value sequence on column 0,
    a sequence on the constant auxiliary column. Returns (builder, trace, pub)."""
    n = 1 << log_n
    b = A.AirBuilder(4, 5, 4, num_pub=1)
    m, mn, a, an, r = b.main, b.main_next, b.aux, b.aux_next, b.rand
    b.transition(mn(0) - (m(0) + m(1)), 1)
    b.transition(mn(1) - (m(1) + mn(0)), 1)
    b.transition(mn(2) - m(2) - 3, 1)
    b.transition(m(3) * (m(3) - 1) * (m(3) - 2), 3)
    b.aux_transition(an(0) * (r(0) + m(2)) - a(0) * (r(0) + m(0)), 2)
    b.aux_transition((an(1) - a(1)) * (r(1) + m(2)) - m(3), 2)
    b.aux_transition(an(2) - a(2) * (r(2) + m(1)) - m(0) * m(3), 2)
    b.aux_transition(an(3) - a(3), 1)
    b.aux_transition(an(4) - (a(4) * a(4) + m(0) + r(0) * a(1)), 2)
    t = np.zeros((4, n), np.uint64)
    x, y = 1, 2
    for i in range(n):
        t[0][i], t[1][i], t[2][i], t[3][i] = x, y, (5 + 3 * i) % P, i % 3
        x, y = (x + y) % P, (y + x + y) % P
    b.assert_single(0, 0, 1)
    b.assert_single(1, 0, 2)
    b.assert_single(1, -1, b.pub(0))
    b.assert_sequence(2, 1, seq_stride, [int(t[2][1 + seq_stride * i]) for i in range(n // seq_stride)])
    b.assert_sequence(0, 0, n // 2, [int(t[0][0]), int(t[0][n // 2])])
    b.aux_assert_single(0, 0, 1)
    b.aux_assert_single(1, 0, 0)
    b.aux_assert_single(2, 0, b.rand(3))
    b.aux_assert_sequence(3, 1, 2, [7] * (n // 2))
    b.aux_builder(0, 1, r(0) + m(0), r(0) + m(2))
    b.aux_builder(1, 0, 1, None, m(3), r(1) + m(2))
    b.aux_builder(2, b.rand(3), r(2) + m(1), None, m(0) * m(3))
    b.aux_builder(3, 7, 1)
    b.aux_assert_single(4, 0, 3)
    b.aux_builder_general(4, 3, a(4) * a(4) + m(0) + r(0) * a(1))
    return b, t, [int(t[1][-1])]


def general_chain_air(log_n):
    """Three general auxiliary recurrences that read each other, every operand kind a general builder may use (own column, earlier general
    columns, main current / next row, periodic column, public input, constant, random element), and no other auxiliary column:
      0  g0' = g0 g0 + m0                       (init 3)
      1  g1' = g1 g0 + k m1 + pub0 + 5          (k = periodic 1, 2, 3, 4; reads general column 0)
      2  g2' = (g2 + r0) (g1 + m0')             (reads general column 1 and the NEXT main row)
    Main: a Fibonacci pair. Returns (builder, trace, pub)."""
    n = 1 << log_n
    b = A.AirBuilder(2, 3, 2, num_pub=2)
    m, mn, a, an, r = b.main, b.main_next, b.aux, b.aux_next, b.rand
    k = b.periodic([1, 2, 3, 4])
    b.transition(mn(0) - (m(0) + m(1)), 1)
    b.transition(mn(1) - (m(1) + mn(0)), 1)
    e0 = a(0) * a(0) + m(0)
    e1 = a(1) * a(0) + k * m(1) + b.pub(0) + 5
    e2 = (a(2) + r(0)) * (a(1) + mn(0))
    b.aux_transition(an(0) - e0, 2)
    b.aux_transition(an(1) - e1, 2, cycles=[4])
    b.aux_transition(an(2) - e2, 2)
    t = np.zeros((2, n), np.uint64)
    x, y = 1, 2
    for i in range(n):
        t[0][i], t[1][i] = x, y
        x, y = (x + y) % P, (y + x + y) % P
    b.assert_single(0, 0, 1)
    b.assert_single(1, 0, 2)
    b.assert_single(1, -1, b.pub(1))
    b.aux_assert_single(0, 0, 3)
    b.aux_assert_single(1, 0, b.pub(0))
    b.aux_assert_single(2, 0, b.rand(1))
    b.aux_builder_general(0, 3, e0)
    b.aux_builder_general(1, b.pub(0), e1)
    b.aux_builder_general(2, b.rand(1), e2)
    return b, t, [11, int(t[1][-1])]
