"""python -m aero_amd — command-line driver of the backend, the counterpart of the reference's `miden-proof-generator`
(miden-proof-generator/src/main.rs:23-51: build options, prove, `proof.to_bytes()`, bincode ProofData container on disk).

    python -m aero_amd prove  --width 2 --log-n 20 --out proofs/fib_gpu.bin [--aux 9,16,8] [--quadratic] [--fold 8] [--blowup 8]
    python -m aero_amd verify proofs/fib_gpu.bin [--aux 9,16,8]        # host only, no GPU
    python -m aero_amd verify /path/to/reference/proofs/fib.bin --miden  # unknown AIR: the Cairo verifier's checks

The container is `u64 len || input_bytes || u64 len || proof_bytes` (miden-proof-generator/src/lib.rs:1-6); for the built-in
AIR input_bytes = the public results as little-endian u64."""
import argparse
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
    v = sub.add_parser("verify")
    v.add_argument("file")
    v.add_argument("--aux", default="0,0,2")
    v.add_argument("--miden", action="store_true", help="proof of an AIR this library does not know (e.g. the reference's proofs/fib.bin)")
    args = ap.parse_args()
    aux = tuple(int(x) for x in args.aux.split(","))
    if args.cmd == "prove":
        opt = aero_amd.ProofOptions.with_96_bit_security()
        opt.fri_folding_factor, opt.blowup_factor = args.fold, args.blowup
        if args.quadratic:
            opt.field_extension = 2
        ctx = aero_amd.Context(args.device)
        dev = ctx.trace_upload(aero_amd.fib_trace(args.width, args.log_n))
        t0 = time.perf_counter()
        proof, pub = ctx.prove_fib_aux(dev, aux[0], aux[1], opt, aux_degree=aux[2])
        ms = (time.perf_counter() - t0) * 1e3
        blob = aero_amd.proof_container(b"".join(struct.pack("<Q", int(x)) for x in pub), proof)
        with open(args.out, "wb") as f:
            f.write(blob)
        print(f"proved {args.width} x 2^{args.log_n} in {ms:.2f} ms (first call includes table setup): {len(proof)} proof bytes -> {args.out}")
    else:
        inputs, proof = split_container(open(args.file, "rb").read())
        if args.miden:
            aero_amd.verify_fib(proof, miden_pub_elements(inputs))
        else:
            aero_amd.verify_fib(proof, list(struct.unpack(f"<{len(inputs) // 8}Q", inputs)), aux)
        print("proof accepted")


if __name__ == "__main__":
    try:
        main()
    except aero_amd.AeroError as e:
        print(e, file=sys.stderr)
        sys.exit(1)
