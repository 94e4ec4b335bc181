"""AIR-as-data without a GPU: the oracle's program evaluator against the oracle's hard-wired FibAir (byte-identical proofs), the
library's host side (parser / validator / compiler, and the verifier that runs the COMPILED program over E for the
out-of-domain check) against proofs the oracle produced. Format: include/aero_air.h; reference seam:
aero-sdk/miden-wasm/src/constraints_worker.rs:32-59."""
import os

import numpy as np
import pytest

import aero_amd
from aero_amd import air as A
from tests import air_examples as ex


@pytest.mark.parametrize("width,log_n,aux,opt", [
    (2, 6, (0, 0, 2), [8, 8, 4, 4, 1, 8, 5]),
    (4, 7, (3, 2, 2), [8, 8, 4, 4, 1, 4, 5]),
    (2, 6, (2, 3, 5), [8, 8, 4, 4, 2, 8, 5]),
    (6, 8, (9, 16, 8), [10, 8, 8, 4, 1, 4, 6]),
    (2, 5, (1, 1, 3), [6, 4, 0, 4, 2, 2, 4]),
])
def test_oracle_program_evaluator_reproduces_the_hard_wired_fibair(oracle, width, log_n, aux, opt):
    trace = oracle.fib_trace(width, log_n)
    want, pub, _ = oracle.prove_fib_aux(width, log_n, aux[0], aux[1], opt, D=aux[2]) if aux[0] else oracle.prove_fib(width, log_n, opt)
    for program in (A.fib_air(width, aux).to_bytes(), aero_amd.fib_program(width, aux)):
        got, _ = oracle.prove_air(program, trace, pub, opt)
        assert got == want
        oracle.verify_air(got, program, pub, log_n)
        air = aero_amd.Air(program)
        aero_amd.verify_air(got, pub, air, min_query_security_bits=0)       # the library's verifier: compiled program on the host
        aero_amd.verify_fib(got, pub, aux, min_query_security_bits=0)       # and its hard-wired FibAir check


@pytest.mark.parametrize("log_n,pairs,aux,opt", [
    (5, 1, 0, [6, 8, 0, 4, 1, 2, 4]),
    (6, 2, 3, [8, 8, 4, 4, 1, 4, 5]),
    (7, 2, 4, [8, 8, 4, 4, 2, 8, 5]),
    (8, 13, 9, [10, 16, 4, 4, 1, 4, 6]),
])
def test_vm_shaped_program_oracle_and_library_verifier(oracle, log_n, pairs, aux, opt):
    b, trace, pub = ex.synth_vm(log_n, pairs, aux)
    program = b.to_bytes()
    oracle.air_check_trace(program, trace, pub)
    proof, _ = oracle.prove_air(program, trace, pub, opt)
    oracle.verify_air(proof, program, pub, log_n)
    air = aero_amd.Air(program)
    info = air.info()
    assert info["ce_blowup"] == 8 and info["main_transition"] == 24 + 2 * pairs and info["aux_transition"] == aux
    assert air.num_divisors(log_n) == oracle.air_info(program, log_n)["columns"] == 8
    assert info["registers_base"] + info["registers_ext"] <= 40           # register allocation (frame values included), not one slot per node
    aero_amd.verify_air(proof, pub, air, min_query_security_bits=0, expected_log_n=log_n)
    # a trace that breaks one constraint: both verifiers refuse the proof the (non-validating) prover makes of it
    bad = trace.copy()
    bad[7][3] ^= 1
    with pytest.raises(RuntimeError):
        oracle.air_check_trace(program, bad, pub)
    forged, _ = oracle.prove_air(program, bad, pub, opt)
    with pytest.raises(RuntimeError):
        oracle.verify_air(forged, program, pub, log_n)
    with pytest.raises(aero_amd.AeroError) as e:
        aero_amd.verify_air(forged, pub, air, min_query_security_bits=0)
    assert e.value.code == -7
    with pytest.raises(aero_amd.AeroError):
        aero_amd.verify_air(proof, pub, air, min_query_security_bits=0, expected_log_n=log_n + 1)


def test_conjectured_security_policy(oracle):
    # 41 field bits over the base field at 2^20 x 8 ... here 2^6 x 8: 64 - 9 = 55 field bits; the quadratic extension has 119
    for ext, ok in ((1, False), (2, True)):
        opt = [27, 8, 16, 4, ext, 8, 5]
        proof, pub, _ = oracle.prove_fib(2, 6, opt)
        q, f = aero_amd.proof_security_bits(proof)
        assert q == 97 and f == 64 * ext - 9
        aero_amd.verify_fib(proof, pub, (0, 0, 2))                                              # query term alone: 97 >= 96
        if ok:
            aero_amd.verify_fib(proof, pub, (0, 0, 2), min_conjectured_security_bits=96)
        else:
            with pytest.raises(aero_amd.AeroError) as e:
                aero_amd.verify_fib(proof, pub, (0, 0, 2), min_conjectured_security_bits=96)
            assert e.value.code == -7 and "field bits" in str(e.value)


def test_malformed_programs_are_refused_on_the_host():
    good = A.fib_air(2).to_bytes()
    aero_amd.Air(good)
    cases = {
        "magic": b"AEROAIX\x01" + good[8:],
        "truncated": good[:-3],
        "trailing": good + b"\0",
        "opcode": None,
    }
    b = A.AirBuilder(2)
    b.transition(b.main_next(0) - b.main(1), 1)
    raw = bytearray(b.to_bytes())
    raw[8 + 64] = 9                       # first node's opcode (no constants, no periodic columns in front of it)
    cases["opcode"] = bytes(raw)
    bb = A.AirBuilder(2, 1, 1)
    bb.transition(bb.main_next(0) - bb.rand(0), 1)        # a main constraint reading a random element
    bb.aux_transition(bb.aux_next(0) - bb.aux(0), 1)
    cases["main constraint over E"] = bb.to_bytes()
    b3 = A.AirBuilder(1)
    b3.transition(b3.main_next(0) - b3.main(0), 1)
    b3.main_asserts.append((0, 0, 0, b3.main(0).ref))     # an assertion whose value depends on the trace
    cases["row-dependent assertion"] = b3.to_bytes()
    for name, data in cases.items():
        with pytest.raises(aero_amd.AeroError) as e:
            aero_amd.Air(data)
        assert e.value.code == -1, name


def test_loader_survives_mutated_programs_under_sanitizers(tmp_path):
    # tests/c_abi/fuzz_air_load.cpp: the AEROAIR loader (host code of the library, it parses bytes a host hands over) built with
    # g++ -fsanitize=address,undefined; bit flips, truncations, insertions and wild counts must end in aero::Error or a clean load
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = str(tmp_path / "fuzz_air_load")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
           "-I", os.path.join(root, "aero_amd", "csrc"), "-I", os.path.join(root, "include"), os.path.join(root, "tests", "c_abi", "fuzz_air_load.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "sanitizer" in r.stderr.lower():
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe, "4000", "3"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "no memory error" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])


def test_message_and_proof_parsers_survive_mutations_under_sanitizers(tmp_path, oracle):
    # tests/c_abi/fuzz_parsers.cpp: the worker-message, proof-layout and public-input parsers (host-only headers of the library) built
    # with g++ -fsanitize=address,undefined, fed mutated copies of well-formed inputs
    import os
    import shutil
    import subprocess
    from aero_amd import messages
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = str(tmp_path / "fuzz_parsers")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
           "-I", os.path.join(root, "aero_amd", "csrc"), "-I", os.path.join(root, "include"), os.path.join(root, "tests", "c_abi", "fuzz_parsers.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "sanitizer" in r.stderr.lower():
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0, r.stderr[-2000:]
    opt, width, log_n = [8, 8, 4, 4, 1, 8, 5], 2, 5
    n = 1 << log_n
    proof, pub, _ = oracle.prove_fib(width, log_n, opt)
    pub_bytes = messages.miden_public_inputs([1, 2, 3, 4], [0, 1], pub)
    lde = [np.arange(8 * n, dtype=np.uint64) + c for c in range(width)]
    coeffs = np.arange(2 * (width + width + width // 2), dtype=np.uint64).reshape(-1, 2)
    files = {"hash": messages.encode_hashing_work_item([[1, 2, 3], [4, 5], [], [7, 8, 9, 10, 11]], 3),
             "cons": messages.encode_constraint_work_item((width, 0, 0), n, pub_bytes, opt, [], coeffs[:width], coeffs[width:], lde, [], 8, 0, 2),
             "proof": proof, "pub": pub_bytes}
    paths = []
    for k, v in files.items():
        p = tmp_path / (k + ".bin")
        p.write_bytes(v)
        paths.append(str(p))
    r = subprocess.run([exe, *paths, "6000", "5"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "no memory error" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])


# ---- AEROAIR version 2 (no GPU: the oracle proves, the library's verifier and loader are host code) ---------------------------------
def test_version2_oracle_proof_is_accepted_by_the_library_verifier(oracle):
    from tests import air_examples as ex
    for log_n, opt in ((4, [27, 8, 8, 4, 1, 8, 4]), (6, [16, 8, 4, 4, 2, 4, 5]), (7, [20, 16, 4, 4, 1, 2, 5])):
        b, trace, pub = ex.v2_air(log_n)
        program = b.to_bytes()
        assert program[:8] == b"AEROAIR\x02"
        proof, _ = oracle.prove_air(program, trace, pub, opt)
        oracle.verify_air(proof, program, pub, log_n)
        air = aero_amd.Air(program)
        aero_amd.verify_air(proof, pub, air, min_query_security_bits=0, expected_log_n=log_n)
        for which in (0, 2):                     # a main sequence, the auxiliary one
            b2, _, _ = ex.v2_air(log_n)
            b2.sequences[which][-1] = (b2.sequences[which][-1] + 1) % ex.P
            with pytest.raises(aero_amd.AeroError):
                aero_amd.verify_air(proof, pub, aero_amd.Air(b2.to_bytes()), min_query_security_bits=0)
            with pytest.raises(RuntimeError):
                oracle.verify_air(proof, b2.to_bytes(), pub, log_n)


def test_version2_loader_rules():
    from aero_amd import air as A
    b = A.AirBuilder(1)
    b.transition(b.main_next(0) - b.main(0) - 1, 1)
    b.assert_sequence(0, 0, 4, [0, 4, 8, 12])
    program = b.to_bytes()
    air = aero_amd.Air(program)                              # loads: the length rule is checked against a trace length
    assert air.num_divisors(4) == 2                          # 16 rows = 4 values x stride 4
    with pytest.raises(aero_amd.AeroError):
        air.num_divisors(5)                                  # 32 rows: stride * values != trace length
    v1 = bytearray(program)
    v1[7] = 1                                                # the same bytes under the version-1 magic: a reserved header word is set
    with pytest.raises(aero_amd.AeroError):
        aero_amd.Air(bytes(v1))
    bad = A.AirBuilder(1)
    bad.transition(bad.main_next(0) - bad.main(0) - 1, 1)
    bad.main_asserts.append((0, 0, 0, (A.SEQ << 24) | 0))    # a sequence value on a single-step assertion
    bad.sequences.append([1, 2])
    with pytest.raises(aero_amd.AeroError):
        aero_amd.Air(bad.to_bytes())
    # affine builders: an additive denominator needs an additive numerator
    c = A.AirBuilder(1, 1, 1)
    c.transition(c.main_next(0) - c.main(0), 1)
    c.aux_transition(c.aux_next(0) - c.aux(0) - c.main(0), 1)
    c.builders[0] = (c.const(0).ref, c.const(1).ref, A.NONE, A.NONE, c.rand(0).ref)
    raw = bytearray(c.to_bytes())
    assert raw[7] == 1                                       # no additive numerator: written as version 1 (3-word builders) ...
    raw[7] = 2
    with pytest.raises(aero_amd.AeroError):                  # ... and as version 2 the record is too short
        aero_amd.Air(bytes(raw))


def test_general_recurrence_loader_rules():
    from aero_amd import air as A

    def base():
        b = A.AirBuilder(1, 2, 1)
        b.transition(b.main_next(0) - b.main(0), 1)
        b.aux_transition(b.aux_next(0) - b.aux(0), 1)
        b.aux_transition(b.aux_next(1) - b.aux(1) * b.aux(1), 2)
        b.aux_builder(0, 1, 1)
        return b
    ok = base()
    ok.aux_builder_general(1, 2, ok.aux(1) * ok.aux(1) + ok.aux(0) * ok.rand(0) + ok.main_next(0))    # own and earlier columns, current row
    prog = ok.to_bytes()
    assert prog[7] == 2
    assert aero_amd.Air(prog).info()["has_aux_builders"] == 1
    later = base()
    later.builders[0] = (later.const(1).ref, (later.aux(1) + 1).ref, A.GENERAL, A.NONE, A.NONE)          # column 0 reading column 1
    later.aux_builder_general(1, 2, later.aux(1))
    with pytest.raises(aero_amd.AeroError):
        aero_amd.Air(later.to_bytes())
    nxt = base()
    nxt.aux_builder_general(1, 2, nxt.aux_next(0) + 1)                                                  # the NEXT row of the auxiliary segment does not exist yet
    with pytest.raises(aero_amd.AeroError):
        aero_amd.Air(nxt.to_bytes())
    v1 = bytearray(prog)
    v1[7] = 1
    with pytest.raises(aero_amd.AeroError):
        aero_amd.Air(bytes(v1))
    affine_reads_aux = base()
    affine_reads_aux.aux_builder(1, 2, affine_reads_aux.aux(0) + 1)                                     # only a general recurrence may read the auxiliary segment
    with pytest.raises(aero_amd.AeroError):
        aero_amd.Air(affine_reads_aux.to_bytes())


def test_cpp_recorder_writes_the_same_programs_as_the_python_builder(tmp_path):
    """include/aero_air_builder.hpp: an AIR written as C++ expressions -> AEROAIR bytes (what a Rust host does with a symbolic field
    element over `Air::evaluate_transition`, constraints_worker.rs:32-43). Same construction order = same bytes as aero_amd/air.py."""
    import shutil
    import subprocess
    from aero_amd import air as A
    from tests import air_examples as ex
    if not shutil.which("g++"):
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "air_builder_demo")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", os.path.join(root, "tests", "c_abi", "air_builder_demo.cpp"), "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    cases = [(["fib", "2", "0", "0", "2"], A.fib_air(2).to_bytes()),
             (["fib", "4", "2", "3", "3"], A.fib_air(4, (2, 3, 3)).to_bytes()),
             (["fib", "72", "9", "16", "8"], A.fib_air(72, (9, 16, 8)).to_bytes()),
             (["v2", "6", "4"], ex.v2_air(6, 4)[0].to_bytes()),                      # version 2: sequences, affine and general builders
             (["v2", "10", "2"], ex.v2_air(10, 2)[0].to_bytes())]
    for args, want in cases:
        got = bytes.fromhex(subprocess.check_output([exe] + args, text=True).strip())
        assert got == want, args
        aero_amd.Air(got)                                                            # and the library loads it
    assert cases[0][1] == aero_amd.fib_program(2)                                    # ... which is also what the library's own emitter writes
