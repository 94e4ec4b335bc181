"""AEROAIR program builder: a constraint system written as Python expressions -> the bytes aero_air_load takes
(format: include/aero_air.h). Host-side convenience only; the library parses, validates and compiles the bytes itself.

The surface follows winter-air's `Air` (whose implementation for Miden, ProcessorAir, the reference instantiates at
aero-sdk/miden-wasm/src/constraints_worker.rs:32-36): transition constraints with a declared `TransitionConstraintDegree`,
`Assertion::single / periodic / sequence`, periodic columns, one auxiliary segment with random elements (winter-air 0.4's
`TraceLayout` holds exactly one auxiliary segment: NUM_AUX_SEGMENTS = 1, and its proof bytes one (width, rands) pair). Programs
that use a sequence assertion or an affine auxiliary builder are written as AEROAIR version 2, everything else as version 1.

    b = AirBuilder(main_width=2, num_pub=1)
    a, bb, na, nb = b.main(0), b.main(1), b.main_next(0), b.main_next(1)
    b.transition(na - (a + bb), degree=1)
    b.transition(nb - (bb + na), degree=1)
    b.assert_single(0, 0, 1); b.assert_single(1, 0, 2); b.assert_single(1, -1, b.pub(0))
    program = b.to_bytes()
"""
import struct

P = 0xFFFFFFFF00000001
MAGIC = b"AEROAIR\x01"
OP_ADD, OP_SUB, OP_MUL = 1, 2, 3
NODE, MAIN_CUR, MAIN_NXT, AUX_CUR, AUX_NXT, PERIODIC, CONST, PUB, RAND, SEQ = range(10)
NONE = 0xFFFFFFFF
GENERAL = 0xFFFFFFFE


class Expr:
    """An operand reference (kind << 24 | index) bound to its builder; arithmetic appends nodes (hash-consed)."""
    __slots__ = ("b", "ref")

    def __init__(self, b, ref):
        self.b, self.ref = b, ref

    def _lift(self, o):
        return o if isinstance(o, Expr) else self.b.const(o)

    def __add__(self, o): return self.b._node(OP_ADD, self, self._lift(o))
    def __radd__(self, o): return self.b._node(OP_ADD, self._lift(o), self)
    def __sub__(self, o): return self.b._node(OP_SUB, self, self._lift(o))
    def __rsub__(self, o): return self.b._node(OP_SUB, self._lift(o), self)
    def __mul__(self, o): return self.b._node(OP_MUL, self, self._lift(o))
    def __rmul__(self, o): return self.b._node(OP_MUL, self._lift(o), self)

    def __pow__(self, e):
        assert isinstance(e, int) and e >= 1
        r, base = None, self
        while e:
            if e & 1:
                r = base if r is None else r * base
            e >>= 1
            if e:
                base = base * base
        return r


class AirBuilder:
    def __init__(self, main_width, aux_width=0, aux_rands=0, num_pub=0, exemptions=1):
        assert 1 <= main_width <= 255 and 0 <= aux_width <= 255 - main_width
        assert (aux_width == 0) == (aux_rands == 0) and aux_rands <= 255
        self.W, self.A, self.R, self.num_pub, self.exemptions = main_width, aux_width, aux_rands, num_pub, exemptions
        self.consts, self._const_idx = [], {}
        self.periodics = []
        self.nodes, self._node_idx = [], {}
        self.main_trans, self.aux_trans = [], []       # (root ref, degree_base, cycles)
        self.main_asserts, self.aux_asserts = [], []   # (column, first_step, stride, value ref)
        self.builders = {}                             # aux column -> (init ref, num ref, den ref, add_num ref, add_den ref)
        self.sequences = []                            # value lists of the sequence assertions

    # ---- operands
    def _ref(self, kind, idx): return Expr(self, (kind << 24) | idx)
    def main(self, c): assert 0 <= c < self.W; return self._ref(MAIN_CUR, c)
    def main_next(self, c): assert 0 <= c < self.W; return self._ref(MAIN_NXT, c)
    def aux(self, c): assert 0 <= c < self.A; return self._ref(AUX_CUR, c)
    def aux_next(self, c): assert 0 <= c < self.A; return self._ref(AUX_NXT, c)
    def pub(self, i): assert 0 <= i < self.num_pub; return self._ref(PUB, i)
    def rand(self, i): assert 0 <= i < self.R; return self._ref(RAND, i)

    def const(self, v):
        v = int(v) % P
        if v not in self._const_idx:
            self._const_idx[v] = len(self.consts)
            self.consts.append(v)
        return self._ref(CONST, self._const_idx[v])

    def periodic(self, values):
        """`get_periodic_column_values`: one cycle of the column (length a power of two >= 2)."""
        vals = [int(v) % P for v in values]
        assert len(vals) >= 2 and len(vals) & (len(vals) - 1) == 0
        self.periodics.append(vals)
        return self._ref(PERIODIC, len(self.periodics) - 1)

    def _node(self, op, a, b):
        key = (op, a.ref, b.ref)
        if op in (OP_ADD, OP_MUL) and b.ref < a.ref:
            key = (op, b.ref, a.ref)
        if key not in self._node_idx:
            self._node_idx[key] = len(self.nodes)
            self.nodes.append(key)
        return self._ref(NODE, self._node_idx[key])

    def _e(self, v): return v if isinstance(v, Expr) else self.const(v)

    # ---- constraints
    def transition(self, expr, degree, cycles=()):
        """`TransitionConstraintDegree::with_cycles(degree, cycles)` for a constraint over the main segment."""
        self.main_trans.append((self._e(expr).ref, degree, tuple(cycles)))

    def aux_transition(self, expr, degree, cycles=()):
        self.aux_trans.append((self._e(expr).ref, degree, tuple(cycles)))

    def assert_single(self, column, step, value):
        """`Assertion::single(column, step, value)`; step < 0 counts from the end (-1 = last row)."""
        self.main_asserts.append((column, step, 0, self._e(value).ref))

    def assert_periodic(self, column, first_step, stride, value):
        """`Assertion::periodic(column, first_step, stride, value)`."""
        assert stride >= 2 and stride & (stride - 1) == 0 and 0 <= first_step < stride
        self.main_asserts.append((column, first_step, stride, self._e(value).ref))

    def _sequence(self, first_step, stride, values):
        vals = [int(v) % P for v in values]
        assert len(vals) >= 2 and len(vals) & (len(vals) - 1) == 0, "a sequence needs a power-of-two number of values >= 2"
        assert stride >= 2 and stride & (stride - 1) == 0 and 0 <= first_step < stride
        self.sequences.append(vals)
        return (SEQ << 24) | (len(self.sequences) - 1)

    def assert_sequence(self, column, first_step, stride, values):
        """`Assertion::sequence(column, first_step, stride, values)`: the column equals values[i] at step first_step + i * stride;
        the trace length must be stride * len(values)."""
        self.main_asserts.append((column, first_step, stride, self._sequence(first_step, stride, values)))

    def aux_assert_sequence(self, column, first_step, stride, values):
        self.aux_asserts.append((column, first_step, stride, self._sequence(first_step, stride, values)))

    def aux_assert_single(self, column, step, value):
        self.aux_asserts.append((column, step, 0, self._e(value).ref))

    def aux_assert_periodic(self, column, first_step, stride, value):
        assert stride >= 2 and stride & (stride - 1) == 0 and 0 <= first_step < stride
        self.aux_asserts.append((column, first_step, stride, self._e(value).ref))

    def aux_builder(self, column, init, num=1, den=None, add=None, add_den=None):
        """aux column(0) = init, column(i+1) = column(i) * num / den + add / add_den, every term evaluated on (row i, row i+1) of the
        main segment: running products (multiset / permutation arguments), running sums (log-derivative arguments: num = 1,
        add = multiplicity, add_den = alpha + value) and mixed forms. add / add_den make the program version 2."""
        assert add is not None or add_den is None
        self.builders[column] = (self._e(init).ref, self._e(num).ref, NONE if den is None else self._e(den).ref,
                                 NONE if add is None else self._e(add).ref, NONE if add_den is None else self._e(add_den).ref)

    def aux_builder_general(self, column, init, nxt):
        """aux column(0) = init, column(i+1) = nxt evaluated on (main row i, main row i+1, CURRENT row of the auxiliary columns up to
        `column` itself): any recurrence - e.g. one that squares its own previous value. It cannot be scanned: the library walks such a
        column row after row on the host (a serial chain has no parallel form; products, sums and affine forms are scanned on the device;
        AERO_AIR_GENERAL_DEVICE=1 walks it with one wavefront per column on the device instead - correct, and about 80 times slower)."""
        self.builders[column] = (self._e(init).ref, self._e(nxt).ref, GENERAL, NONE, NONE)

    def to_bytes(self):
        nb = len(self.builders)
        assert nb in (0, self.A) and sorted(self.builders) == list(range(nb)), "one aux builder per aux column, or none"
        v2 = bool(self.sequences) or any(b[3] != NONE or b[2] == GENERAL for b in self.builders.values())
        out = bytearray(b"AEROAIR\x02" if v2 else MAGIC)
        out += struct.pack("<16I", self.W, self.A, self.R, self.num_pub, self.exemptions, len(self.consts), len(self.periodics),
                           len(self.nodes), len(self.main_trans), len(self.aux_trans), len(self.main_asserts), len(self.aux_asserts),
                           nb, len(self.sequences), 0, 0)
        out += struct.pack(f"<{len(self.consts)}Q", *self.consts)
        for vals in self.periodics:
            out += struct.pack(f"<I{len(vals)}Q", len(vals), *vals)
        for vals in self.sequences:
            out += struct.pack(f"<I{len(vals)}Q", len(vals), *vals)
        for op, a, b in self.nodes:
            out += struct.pack("<3I", op, a, b)
        for root, deg, cyc in self.main_trans + self.aux_trans:
            out += struct.pack(f"<3I{len(cyc)}I", root, deg, len(cyc), *cyc)
        for col, first, stride, val in self.main_asserts + self.aux_asserts:
            out += struct.pack("<IiII", col, first, stride, val)
        for c in range(nb):
            out += struct.pack("<5I", *self.builders[c]) if v2 else struct.pack("<3I", *self.builders[c][:3])
        return bytes(out)


def fib_air(width, aux=(0, 0, 2)):
    """The built-in FibAir(width) with its optional auxiliary segment (include/aero_stark.h: aero_fib_air) as a program:
    pair k = columns (2k, 2k+1) = (a, b): a' = a + b, b' = b + a'; a(0) = 1 + 2k, b(0) = 2 + 2k, b(n-1) = pub[k];
    aux column c: p(0) = 1, p' = p * (rand[c mod R] + main[c mod W])^(D-1)."""
    A, R, D = aux if aux and aux[0] else (0, 0, 2)
    b = AirBuilder(width, A, R, num_pub=width // 2)
    for k in range(width // 2):
        a, bb, na, nb = b.main(2 * k), b.main(2 * k + 1), b.main_next(2 * k), b.main_next(2 * k + 1)
        b.transition(na - (a + bb), 1)
        b.transition(nb - (bb + na), 1)
    for c in range(width):
        b.assert_single(c, 0, 1 + c)
    for k in range(width // 2):
        b.assert_single(2 * k + 1, -1, b.pub(k))
    for c in range(A):
        f = (b.rand(c % R) + b.main(c % width)) ** (D - 1)
        b.aux_transition(b.aux_next(c) - b.aux(c) * f, D)
        b.aux_assert_single(c, 0, 1)
        b.aux_builder(c, 1, f)
    return b
