"""python -m aero_amd — command-line driver of the backend, the counterpart of the reference's `miden-proof-generator`
(miden-proof-generator/src/main.rs:23-51: build options, prove, `proof.to_bytes()`, bincode ProofData container on disk).

    python -m aero_amd prove  --width 2 --log-n 20 --out proofs/fib_gpu.bin [--aux 9,16,8] [--quadratic] [--fold 8] [--blowup 8]
    python -m aero_amd prove  --trace dump.aerotrc --out proofs/p.bin    # a trace produced elsewhere (AEROTRC file: include/aero_stark.h)
    python -m aero_amd program --log-n 12 --out /tmp/vm       # VM-shaped constraint program + a trace that satisfies it + public inputs (no GPU)
    python -m aero_amd prove  --trace /tmp/vm.aerotrc --air /tmp/vm.aeroair --pub /tmp/vm.pub --fold 4 --out /tmp/vm.bin ; python -m aero_amd verify /tmp/vm.bin --air /tmp/vm.aeroair
    python -m aero_amd trace  --width 2 --log-n 20 --out dump.aerotrc [--aux 9,16,8]   # write the synthetic trace in that format (no GPU)
    python -m aero_amd cairo  proofs/p.bin proof | public-inputs | trace-queries '[5207,...]' | constraint-queries '[..]' | fri-queries '[..]'
                                                                         # = bin/stark_parser <file> <command> (miden-to-cairo-parser/src/main.rs:42-113)
    python -m aero_amd protobuf proofs/p.bin --out p.pb [--public-inputs]  # sdk.StarkProof / sdk.MidenPublicInputs bytes (aero-sdk/proto)
    python -m aero_amd verify proofs/fib_gpu.bin [--aux 9,16,8]        # host only, no GPU
    python -m aero_amd verify /path/to/reference/proofs/fib.bin --miden [--cairo-compat]   # unknown AIR: the Cairo verifier's checks

The container is `u64 len || input_bytes || u64 len || proof_bytes` (miden-proof-generator/src/lib.rs:1-6); for the built-in
AIR input_bytes = the public results as little-endian u64."""
import argparse
import os
import struct
import sys
import time

import aero_amd


def split_container(blob):
    (n,) = struct.unpack_from("<Q", blob, 0)
    inputs = blob[8:8 + n]
    (m,) = struct.unpack_from("<Q", blob, 8 + n)
    proof = blob[16 + n:16 + n + m]
    if 16 + n + m != len(blob):
        raise SystemExit("not a ProofData container")
    return inputs, proof


def miden_pub_elements(inputs):
    """program hash (4 elements) || stack inputs || outputs.stack || overflow addresses, lengths dropped."""
    elems = list(struct.unpack_from("<4Q", inputs, 0))
    off = 32
    for _ in range(3):
        (cnt,) = struct.unpack_from("<Q", inputs, off)
        off += 8
        elems += list(struct.unpack_from(f"<{cnt}Q", inputs, off))
        off += 8 * cnt
    return elems


def main():
    ap = argparse.ArgumentParser(prog="python -m aero_amd")
    sub = ap.add_subparsers(dest="cmd", required=True)
    p = sub.add_parser("prove")
    p.add_argument("--width", type=int, default=2)
    p.add_argument("--log-n", type=int, default=10)
    p.add_argument("--aux", default="0,0,2", help="aux_width,aux_rands,aux_degree")
    p.add_argument("--quadratic", action="store_true")
    p.add_argument("--fold", type=int, default=8)
    p.add_argument("--blowup", type=int, default=8)
    p.add_argument("--device", type=int, default=0)
    p.add_argument("--out", required=True)
    p.add_argument("--trace", default=None, help="AEROTRC file holding the trace (and its AIR parameters) instead of the synthetic one")
    p.add_argument("--air", default=None, help="AEROAIR constraint program (include/aero_air.h) the trace satisfies: proves through aero_prove_air")
    p.add_argument("--pub", default="", help="--air: the program's public-input elements, comma separated")
    p.add_argument("--no-verify", action="store_true", help="skip the verification miden-proof-generator runs on every proof before it writes it "
                   "(main.rs:47); by default the library verifies the proof inside the call (aero_ctx_set_self_verify) and nothing is written on a rejection")
    g = sub.add_parser("program", help="write the VM-shaped synthetic constraint program, a trace that satisfies it and its public inputs (no GPU)")
    g.add_argument("--log-n", type=int, default=10)
    g.add_argument("--pairs", type=int, default=26)
    g.add_argument("--aux", type=int, default=9)
    g.add_argument("--rands", type=int, default=16)
    g.add_argument("--out", required=True, help="prefix: <out>.aeroair, <out>.aerotrc, <out>.pub")
    t = sub.add_parser("trace")
    t.add_argument("--width", type=int, default=2)
    t.add_argument("--log-n", type=int, default=10)
    t.add_argument("--aux", default="0,0,2")
    t.add_argument("--out", required=True)
    c = sub.add_parser("cairo")
    c.add_argument("file")
    c.add_argument("command", choices=sorted(aero_amd.CAIRO_COMMANDS))
    c.add_argument("indexes", nargs="?", default="[]", help="JSON array of query positions (the *-queries commands)")
    b = sub.add_parser("protobuf")
    b.add_argument("file")
    b.add_argument("--out", required=True)
    b.add_argument("--public-inputs", action="store_true", help="encode the container's Miden public inputs instead of the proof")
    v = sub.add_parser("verify")
    v.add_argument("file")
    v.add_argument("--aux", default="0,0,2")
    v.add_argument("--miden", action="store_true", help="proof of an AIR this library does not know (e.g. the reference's proofs/fib.bin): "
                   "everything except the out-of-domain constraint check, which is what the reference's Cairo verifier does")
    v.add_argument("--min-security", type=int, default=96, help="reject proofs whose num_queries * log2(blowup) + grinding is below this")
    v.add_argument("--log-n", type=int, default=0, help="trace length the statement is about (0 = accept the proof's own)")
    v.add_argument("--cairo-compat", action="store_true", help="also require the shape the reference's Cairo verifier hard-codes")
    v.add_argument("--air", default=None, help="AEROAIR constraint program the proof is about (out-of-domain check over the program)")
    args = ap.parse_args()
    if args.cmd == "cairo":
        import json
        inputs, proof = split_container(open(args.file, "rb").read())
        print(aero_amd.cairo_memory(args.command, proof, inputs, json.loads(args.indexes)))
        return
    if args.cmd == "protobuf":
        inputs, proof = split_container(open(args.file, "rb").read())
        data = aero_amd.miden_public_inputs_to_protobuf(inputs) if args.public_inputs else aero_amd.proof_to_protobuf(proof)
        with open(args.out, "wb") as f:
            f.write(data)
        print(f"{len(data)} protobuf bytes -> {args.out}")
        return
    if args.cmd == "program":
        program = aero_amd.synth_vm_program(args.log_n, args.pairs, args.aux, args.rands)
        trace, pub = aero_amd.synth_vm_trace(args.log_n, args.pairs)
        with open(args.out + ".aeroair", "wb") as f:
            f.write(program)
        aero_amd.trace_file_write(args.out + ".aerotrc", trace, (args.aux, args.rands, 2), air_id=aero_amd.AIR_PROGRAM)
        with open(args.out + ".pub", "w") as f:
            f.write(",".join(str(x) for x in pub))
        info = aero_amd.Air(program).info()
        print(f"{info['main_width']} + {info['aux_width']} columns, {info['main_transition'] + info['aux_transition']} transition constraints, "
              f"{info['main_assertions'] + info['aux_assertions']} assertions, 2^{args.log_n} rows -> {args.out}.aeroair / .aerotrc / .pub")
        return
    aux = tuple(int(x) for x in args.aux.split(","))
    if args.cmd == "trace":
        aero_amd.trace_file_write(args.out, aero_amd.fib_trace(args.width, args.log_n), aux)
        print(f"{args.width} x 2^{args.log_n} trace -> {args.out}")
        return
    if args.cmd == "prove":
        opt = aero_amd.ProofOptions.with_96_bit_security()
        opt.fri_folding_factor, opt.blowup_factor = args.fold, args.blowup
        if args.quadratic:
            opt.field_extension = 2
        ctx = aero_amd.Context(args.device)
        ctx.set_self_verify(not args.no_verify)
        if args.trace:
            dev, _, aux = ctx.trace_file_load(args.trace)
            args.width, rows = dev.shape
            args.log_n = rows.bit_length() - 1
        else:
            dev = ctx.trace_upload(aero_amd.fib_trace(args.width, args.log_n))
        t0 = time.perf_counter()
        if args.air:
            air = aero_amd.Air(open(args.air, "rb").read())
            pub = [int(x) for x in (open(args.pub).read() if os.path.exists(args.pub) else args.pub).split(",") if x.strip()]
            proof = ctx.prove_air(air, dev, pub, opt)
        else:
            proof, pub = ctx.prove_fib_aux(dev, aux[0], aux[1], opt, aux_degree=aux[2])
        ms = (time.perf_counter() - t0) * 1e3
        blob = aero_amd.proof_container(b"".join(struct.pack("<Q", int(x)) for x in pub), proof)
        with open(args.out, "wb") as f:
            f.write(blob)
        print(f"proved {args.width} x 2^{args.log_n} in {ms:.2f} ms (first call includes table setup): {len(proof)} proof bytes -> {args.out}")
    else:
        inputs, proof = split_container(open(args.file, "rb").read())
        pol = dict(min_query_security_bits=args.min_security, expected_log_n=args.log_n, cairo_compat=args.cairo_compat)
        if args.air:
            pol.pop("cairo_compat")
            aero_amd.verify_air(proof, list(struct.unpack(f"<{len(inputs) // 8}Q", inputs)), aero_amd.Air(open(args.air, "rb").read()), **pol)
        elif args.miden:
            aero_amd.verify_fib(proof, miden_pub_elements(inputs), None, allow_unknown_air=True, **pol)
        else:
            aero_amd.verify_fib(proof, list(struct.unpack(f"<{len(inputs) // 8}Q", inputs)), aux, **pol)
        q, f = aero_amd.proof_security_bits(proof)
        print(f"proof accepted (query security {q} bits, field-size term {f} bits)")


if __name__ == "__main__":
    try:
        main()
    except aero_amd.AeroError as e:
        print(e, file=sys.stderr)
        sys.exit(1)
