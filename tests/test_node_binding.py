"""The Node binding of the worker seam (bindings/node: N-API addon + worker-thread shims after aero-sdk/src/hashing_worker.ts and
constraints_worker.ts). CPU part: the addon builds against node_api.h, loads the library and produces the same ProverOutput
message as the C ABI called from Python. GPU part: the SDK's pool pattern on Node - batches of rows posted to hashing workers
round robin, answers merged by batch index (pool.rs:84-104, proving_worker.rs:154-159) - with every digest checked against
Node's own BLAKE2s."""
import json
import os
import shutil
import struct
import subprocess

import pytest

import aero_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE_DIR = os.path.join(ROOT, "bindings", "node")
needs_node = pytest.mark.skipif(shutil.which("node") is None or not os.path.exists("/usr/include/node/node_api.h"),
                                reason="node or its N-API headers are not installed")


@pytest.fixture(scope="module")
def addon():
    subprocess.check_call(["make", "-C", NODE_DIR, "-s"])
    path = os.path.join(NODE_DIR, "aero_worker.node")
    assert os.path.exists(path)
    return path


@needs_node
def test_prover_output_through_the_node_addon(addon, golden_dir):
    aero_amd.lib()
    container = os.path.join(golden_dir, "fib.bin")
    out = subprocess.run(["node", os.path.join(NODE_DIR, "prover_output.js"), aero_amd.LIB_PATH, container], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    blob = open(container, "rb").read()
    (n,) = struct.unpack_from("<Q", blob, 0)
    inputs = blob[8:8 + n]
    (m,) = struct.unpack_from("<Q", blob, 8 + n)
    proof = blob[16 + n:16 + n + m]
    want = aero_amd.prover_output(proof, inputs)
    assert bytes.fromhex(res["hex"]) == want
    assert res["proofLen"] == len(aero_amd.proof_to_protobuf(proof)) and res["publicInputsLen"] == len(aero_amd.miden_public_inputs_to_protobuf(inputs))


@needs_node
def test_node_addon_fails_loudly_without_a_library(addon):
    code = "const a = require(process.argv[1]); try { a.open('/nonexistent/libaero_stark.so', 0); console.log('opened'); } catch (e) { console.log('error: ' + e.message); }"
    out = subprocess.run(["node", "-e", code, addon], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and out.stdout.startswith("error: aero_worker: cannot load the library"), out.stdout + out.stderr


@needs_node
@pytest.mark.gpu
@pytest.mark.parametrize("rows,width,chunk,workers", [(4096, 72, 1024, 2), (1000, 2, 300, 3)])
def test_hashing_pool_on_node_worker_threads(addon, rows, width, chunk, workers):
    aero_amd.lib()
    out = subprocess.run(["node", os.path.join(NODE_DIR, "demo_pool.js"), aero_amd.LIB_PATH, str(rows), str(width), str(chunk), str(workers)],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-1000:] + out.stderr[-2000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["ok"] and res["rows"] == rows and res["mismatches"] == 0 and res["batches"] == -(-rows // chunk)
