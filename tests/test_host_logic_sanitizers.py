"""The library's HOST logic under ThreadSanitizer and AddressSanitizer against a stand-in HIP runtime (tools/hipstub: streams are
ordered queues on worker threads, copies move bytes, events order streams, kernels do not run) - no GPU involved: context and
matrix lifetimes from several threads, the pool's worker threads, the thread-rank group's rendezvous + event protocol with every
exchanged byte checked, the abort path, error codes, the stage entry points (staging ring, table caches), and whole proofs of the
built-in AIR and of a constraint program run as far as the prover's own consistency check lets them without kernels (the host
pipeline, the module path of the run-time compiled kernel, the error path that drains the streams). (TSan found the unsynchronised slot binding of
aero_local_group_comm this way.) The reference's counterpart is the worker pool of aero-sdk/miden-wasm/src/proving_worker.rs:276-321."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("sanitizer", ["thread", "address"])
def test_host_logic_is_clean_under(sanitizer):
    if not shutil.which("hipcc") or not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("needs the ROCm clang")
    env = {k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD", "AERO_CRASH_TRACE")}
    r = subprocess.run([os.path.join(ROOT, "tools", "hipstub", "run.sh"), sanitizer], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=1200, env=env)
    tail = r.stdout[-4000:]
    if r.returncode != 0 and any(k in r.stdout for k in ("unexpected memory mapping", "ReExec", "personality", "Shadow memory range interleaves")):
        pytest.skip("the sanitizer runtime cannot set up its shadow memory in this environment")
    assert r.returncode == 0, tail
    assert "host logic ok" in r.stdout, tail
    assert "ThreadSanitizer" not in r.stdout and "AddressSanitizer" not in r.stdout, tail
