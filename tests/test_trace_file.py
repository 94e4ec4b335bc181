"""The AEROTRC trace hand-over format (include/aero_stark.h: aero_trace_file_*): how a trace produced by another program
(the reference: `processor::execute` in miden-proof-generator/src/main.rs:20-31) reaches this library."""
import json
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

import aero_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_write_then_info_and_layout(tmp_path):
    t = aero_amd.fib_trace(4, 5)
    path = tmp_path / "t.aerotrc"
    aero_amd.trace_file_write(path, t, (3, 2, 5))
    assert aero_amd.trace_file_info(path) == (4, 5, aero_amd.AIR_FIB, (3, 2, 5))
    raw = open(path, "rb").read()
    assert raw[:8] == b"AEROTRC\x01" and struct.unpack_from("<6I", raw, 8) == (4, 5, 0, 3, 2, 5)
    assert np.frombuffer(raw[32:], "<u8").reshape(4, 32).tolist() == t.tolist()       # column-major, little-endian
    with pytest.raises(aero_amd.AeroError):
        aero_amd.trace_file_info(tmp_path / "missing")
    bad = tmp_path / "bad"
    bad.write_bytes(b"NOTATRACE" + raw[9:])
    with pytest.raises(aero_amd.AeroError):
        aero_amd.trace_file_info(bad)


def test_command_line_writes_a_trace_and_reencodes_the_golden_container(tmp_path, golden_dir):
    out = tmp_path / "t.aerotrc"
    r = subprocess.run([sys.executable, "-m", "aero_amd", "trace", "--width", "2", "--log-n", "6", "--out", str(out)], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr
    assert aero_amd.trace_file_info(out)[:3] == (2, 6, 0)
    fib = os.path.join(golden_dir, "fib.bin")
    # = `bin/stark_parser proofs/fib.bin proof` / `... public-inputs` / `... trace-queries <positions>`
    r = subprocess.run([sys.executable, "-m", "aero_amd", "cairo", fib, "proof"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0 and json.loads(r.stdout)[0] == "0x48", r.stderr
    r = subprocess.run([sys.executable, "-m", "aero_amd", "cairo", fib, "public-inputs"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0 and json.loads(r.stdout)[0] == "0x4"
    pos = json.load(open(os.path.join(golden_dir, "fib_kat.json")))["G1"]["positions"]
    r = subprocess.run([sys.executable, "-m", "aero_amd", "cairo", fib, "fri-queries", json.dumps(pos)], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0 and len(json.loads(r.stdout)) > 1000
    pb = tmp_path / "p.pb"
    r = subprocess.run([sys.executable, "-m", "aero_amd", "protobuf", fib, "--out", str(pb)], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0 and pb.stat().st_size > 50303


@pytest.mark.gpu
def test_load_and_prove_from_a_trace_file(tmp_path, oracle):
    ctx = aero_amd.Context(0)
    opt = aero_amd.ProofOptions.with_96_bit_security()
    for (W, log_n, air) in [(2, 10, (0, 0, 2)), (6, 23 - 3, (0, 0, 2)), (4, 9, (3, 2, 5))]:     # 2^20 x 6 = 48 MiB: several 32 MiB chunks
        t = aero_amd.fib_trace(W, log_n)
        path = tmp_path / f"t{W}_{log_n}.aerotrc"
        aero_amd.trace_file_write(path, t, air)
        dev, air_id, got_air = ctx.trace_file_load(path)
        assert air_id == aero_amd.AIR_FIB and got_air == air
        assert dev.shape == (W, 1 << log_n) and (dev.download() == t).all()
        if log_n <= 10:
            proof, pub = ctx.prove_fib_aux(dev, air[0], air[1], opt, aux_degree=air[2])
            want, want_pub, _ = oracle.prove_fib_aux(W, log_n, air[0], air[1], opt.to_list(), D=air[2])
            assert proof == want and pub == want_pub
        dev.free()
    # refused: a trace for an AIR this library does not have, a non-canonical element, a truncated file, trailing bytes
    t = aero_amd.fib_trace(2, 6)
    p = tmp_path / "miden.aerotrc"
    aero_amd.trace_file_write(p, t, air_id=aero_amd.AIR_MIDEN_PROCESSOR)
    with pytest.raises(aero_amd.AeroError) as e:
        ctx.trace_file_load(p)
    assert e.value.code == -5
    t[1, 7] = aero_amd.P
    p = tmp_path / "noncanon.aerotrc"
    aero_amd.trace_file_write(p, t)
    with pytest.raises(aero_amd.AeroError):
        ctx.trace_file_load(p)
    good = tmp_path / "good.aerotrc"
    aero_amd.trace_file_write(good, aero_amd.fib_trace(2, 6))
    raw = open(good, "rb").read()
    for name, data in (("short", raw[:-8]), ("long", raw + b"\0")):
        q = tmp_path / name
        q.write_bytes(data)
        with pytest.raises(aero_amd.AeroError):
            ctx.trace_file_load(q)
    dev, _, _ = ctx.trace_file_load(good)                  # the context is still usable
    assert dev.shape == (2, 64)
    dev.free()
    # command line: prove --trace
    out = tmp_path / "p.bin"
    r = subprocess.run([sys.executable, "-m", "aero_amd", "prove", "--trace", str(tmp_path / "t4_9.aerotrc"), "--out", str(out)], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([sys.executable, "-m", "aero_amd", "verify", str(out), "--aux", "3,2,5", "--log-n", "9"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0 and "accepted" in r.stdout, r.stderr
    ctx.close()


def test_command_line_writes_a_constraint_program_with_its_trace(tmp_path):
    # `python -m aero_amd program`: the VM-shaped program, a trace file with the "AIR travels as a program" id and the public inputs
    pre = tmp_path / "vm"
    r = subprocess.run([sys.executable, "-m", "aero_amd", "program", "--log-n", "7", "--pairs", "2", "--aux", "3", "--rands", "4", "--out", str(pre)],
                       capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr
    assert aero_amd.trace_file_info(str(pre) + ".aerotrc")[:3] == (24, 7, aero_amd.AIR_PROGRAM)
    air = aero_amd.Air(open(str(pre) + ".aeroair", "rb").read())
    assert air.info()["main_width"] == 24 and len(open(str(pre) + ".pub").read().split(",")) == air.info()["num_pub"]


@pytest.mark.gpu
def test_prove_a_foreign_trace_with_its_constraint_program_from_the_command_line(tmp_path, oracle):
    # what a Miden build would do: dump the trace (AIR id 2) and the AIR as a program, prove, verify with the OOD check over the program
    pre = tmp_path / "vm"
    run = lambda *a: subprocess.run([sys.executable, "-m", "aero_amd", *a], capture_output=True, text=True, cwd=ROOT)
    r = run("program", "--log-n", "10", "--pairs", "3", "--aux", "4", "--rands", "4", "--out", str(pre))
    assert r.returncode == 0, r.stderr
    out = tmp_path / "vm.bin"
    r = run("prove", "--trace", str(pre) + ".aerotrc", "--air", str(pre) + ".aeroair", "--pub", str(pre) + ".pub", "--fold", "4", "--out", str(out))
    assert r.returncode == 0, r.stderr
    r = run("verify", str(out), "--air", str(pre) + ".aeroair", "--log-n", "10")
    assert r.returncode == 0 and "accepted" in r.stdout, r.stderr
    # the same bytes as the oracle's proof of that program and trace
    program = open(str(pre) + ".aeroair", "rb").read()
    trace, pub = aero_amd.synth_vm_trace(10, 3)
    opt = aero_amd.ProofOptions.with_96_bit_security()
    opt.fri_folding_factor = 4
    want, _ = oracle.prove_air(program, trace, pub, opt.to_list())
    blob = open(out, "rb").read()
    assert blob.endswith(want)
    # another program does not accept it
    other = tmp_path / "other.aeroair"
    other.write_bytes(aero_amd.synth_vm_program(10, 3, 4, 5))
    r = run("verify", str(out), "--air", str(other), "--log-n", "10")
    assert r.returncode != 0
