"""The run-time compiled evaluation kernel of an AIR program (aero_amd/csrc/air_jit.hip) needs no GPU to be BUILT: hiprtc
cross-compiles for gfx950 like hipcc does. Checked here: the generated source is what the program says (one definition per node it
needs, literals for constants, the boundary divisors as one fraction per row) and it compiles, for both fields and both output forms."""
import re

import pytest

import aero_amd
from tests import air_examples as ex


@pytest.mark.parametrize("ext,fused", [(1, True), (2, True), (1, False)])
def test_fibair_program_compiles_for_gfx950(ext, fused):
    air = aero_amd.Air(aero_amd.fib_program(4, (2, 3, 2)))
    src = air.jit_source(10, ext, fused)
    assert 'extern "C" __global__' in src and ("typedef gl::FQ F;" if ext == 2 else "typedef gl::FB F;") in src
    assert ("sh_d[q][threadIdx.x] = run;" in src) == fused          # one inversion per thread over its rows, fused form only
    assert ("a.out_cols[" in src) == (not fused)
    air.jit_compile(10, ext, fused)
    air.jit_compile(10, ext, fused)                                 # cached in the handle


def test_vm_shaped_program_source_and_build():
    b, _, _ = ex.synth_vm(6, 2, 3)
    air = aero_amd.Air(b.to_bytes())
    info = air.info()
    src = air.jit_source(6, 1, True)
    body = src[src.index('extern "C" __global__'):]
    nt = info["main_transition"] + info["aux_transition"]
    # one accumulation per transition constraint: base-field values as 160-bit sums (wmac), E-valued ones as field multiplications
    assert len(re.findall(r"^\s+acc = F::add\(acc, F::mulb?\(F::make\(pool\[oT", body, re.M)) + len(re.findall(r"^\s+wmac\(wacc_0, pool\[oT", body, re.M)) == nt
    assert len(re.findall(r"pool\[oB \+ \d+\]", body)) == 2 * (info["main_assertions"] + info["aux_assertions"])
    assert len(re.findall(r"const uint64_t d\d+ = ", body)) == air.num_divisors(6) - 1
    defs = re.findall(r"const (?:uint64_t|T) ([te]\d+) = ", body)
    assert len(defs) == len(set(defs)) <= info["num_nodes"]          # every node at most once
    assert "0x9e3779b97f4a7c15" not in body                           # (no stray literals: constants appear as written in the program)
    air.jit_compile(6, 1, True)


def test_a_program_without_assertion_groups_needs_no_inversion():
    b = aero_amd.air.AirBuilder(1)
    b.transition(b.main_next(0) - 2 * b.main(0), 1)
    b.assert_single(0, 0, b.const(1))
    air = aero_amd.Air(b.to_bytes())
    src = air.jit_source(5, 1, True)
    assert "gl::inv(tot)" in src                                       # one divisor group: one fraction per row
    air.jit_compile(5, 1, True)


def test_code_objects_are_cached_on_disk_when_asked(tmp_path, monkeypatch):
    import time
    monkeypatch.setenv("AERO_AIR_JIT_CACHE", str(tmp_path))
    program = aero_amd.fib_program(6, (2, 3, 3))
    t0 = time.perf_counter()
    aero_amd.Air(program).jit_compile(9, 1, True)
    first = time.perf_counter() - t0
    files = [f for f in tmp_path.iterdir() if f.suffix == ".co"]
    assert len(files) == 1 and files[0].stat().st_size > 1000 and not [f for f in tmp_path.iterdir() if ".tmp" in f.name]
    t0 = time.perf_counter()
    aero_amd.Air(program).jit_compile(9, 1, True)          # a fresh handle: served from the directory
    assert time.perf_counter() - t0 < first / 2
    aero_amd.Air(program).jit_compile(9, 2, True)          # another field: another kernel
    assert len([f for f in tmp_path.iterdir() if f.suffix == ".co"]) == 2


def test_a_corrupt_or_foreign_cache_entry_is_recompiled_not_loaded(tmp_path, monkeypatch):
    monkeypatch.setenv("AERO_AIR_JIT_CACHE", str(tmp_path))
    program = aero_amd.fib_program(4, (1, 1, 2))
    aero_amd.Air(program).jit_compile(8, 1, True)
    (entry,) = [f for f in tmp_path.iterdir() if f.suffix == ".co"]
    good = entry.read_bytes()
    assert good[:8] == b"AEROJIT1" and good[8 + 16 + 64:][:4] == b"\x7fELF"       # header (magic, lengths, two BLAKE2s digests) + code object
    for bad in (good[:-100],                                                     # truncated
                good[:200] + bytes([good[200] ^ 1]) + good[201:],                # one flipped bit in the code object
                good[88:],                                                       # a bare code object under the right name (the old format)
                b"AEROJIT1" + bytes(80) + good[88:]):                            # header that does not match
        entry.write_bytes(bad)
        aero_amd.Air(program).jit_compile(8, 1, True)                            # fresh handle: reads the directory, rejects, recompiles
        assert entry.read_bytes() == good


def test_a_cache_directory_others_can_write_to_is_ignored(tmp_path, monkeypatch):
    import os
    d = tmp_path / "shared"
    d.mkdir()
    os.chmod(d, 0o777)
    monkeypatch.setenv("AERO_AIR_JIT_CACHE", str(d))
    aero_amd.Air(aero_amd.fib_program(4)).jit_compile(8, 1, True)                # still compiles ...
    assert not list(d.iterdir())                                                 # ... but neither reads nor writes there


def test_prepare_builds_the_kernel_a_proof_will_ask_for():
    # aero_air_prepare: exactly the (field, rows per thread) form a proof of that length uses, ahead of the proof; no GPU involved
    import time
    from tests import air_examples as ex
    b, _, _ = ex.v2_air(10)
    air = aero_amd.Air(b.to_bytes())
    opt = aero_amd.ProofOptions(27, 8, 8, 4, 2, 4, 6)
    t0 = time.perf_counter()
    air.prepare(10, opt)
    first = time.perf_counter() - t0
    t0 = time.perf_counter()
    air.prepare(10, opt)                                   # cached in the handle
    air.prepare(10, opt, world=1)
    assert time.perf_counter() - t0 < max(0.05, first / 5)
    air.prepare(10, opt, world=4)                          # a rank's share of the domain: possibly another rows-per-thread form
    with pytest.raises(aero_amd.AeroError):
        air.prepare(10, opt, world=3)                      # ranks are a power of two
    with pytest.raises(aero_amd.AeroError):
        air.prepare(2, opt)                                # trace lengths start at 2^3
