"""Host placement next to the GPU (aero_amd/csrc/numa.hip; include/aero_stark.h: aero_numa_*): the sysfs parsing against a synthetic tree, on
a box without a GPU. The reference's pool has no placement (aero-sdk/miden-wasm/src/pool.rs:28-45 sizes it by hardwareConcurrency); on a
two-socket 8-GPU node a rank's workers and pinned buffers belong on its GPU's socket."""
import ctypes as C

import numpy as np
import pytest

import aero_amd


def cpulist(text):
    n = C.c_uint32(0)
    rc = aero_amd.lib().aero_numa_parse_cpulist(text.encode(), None, C.c_uint32(0), C.byref(n))
    if rc != 0:
        return None
    out = (C.c_int32 * max(1, n.value))()
    assert aero_amd.lib().aero_numa_parse_cpulist(text.encode(), out, C.c_uint32(n.value), C.byref(n)) == 0
    return list(out[:n.value])


def test_cpulist_format():
    assert cpulist("0-3") == [0, 1, 2, 3]
    assert cpulist("0-1,8-9,64\n") == [0, 1, 8, 9, 64]
    assert cpulist("7") == [7]
    assert cpulist("") == [] and cpulist("\n") == []
    assert cpulist(" 2 - 3") is None            # the kernel writes no blanks inside a range
    assert cpulist("3-1") is None and cpulist("a") is None and cpulist("1,,2") is None and cpulist("1-") is None
    assert cpulist("0-127,256-383") == list(range(128)) + list(range(256, 384))


def tree(tmp_path, bdf, node, cpus_by_node):
    d = tmp_path / "bus" / "pci" / "devices" / bdf
    d.mkdir(parents=True)
    (d / "numa_node").write_text(f"{node}\n")
    for k, text in cpus_by_node.items():
        nd = tmp_path / "devices" / "system" / "node" / f"node{k}"
        nd.mkdir(parents=True)
        (nd / "cpulist").write_text(text + "\n")
    return str(tmp_path).encode()


def query(root, bdf):
    node, n = C.c_int32(-7), C.c_uint32(0)
    cpus = (C.c_int32 * 1024)()
    rc = aero_amd.lib().aero_numa_query(root, bdf.encode(), C.byref(node), cpus, C.c_uint32(1024), C.byref(n))
    return rc, node.value, list(cpus[:n.value])


def test_device_node_and_its_cpus_from_sysfs(tmp_path):
    root = tree(tmp_path, "0000:c1:00.0", 1, {0: "0-63,128-191", 1: "64-127,192-255"})
    rc, node, cpus = query(root, "0000:C1:00.0")                 # hipDeviceGetPCIBusId prints upper-case hex
    assert rc == 0 and node == 1 and cpus == list(range(64, 128)) + list(range(192, 256))
    rc, node, cpus = query(root, "0000:05:00.0")                 # no such device: unknown, nothing bound
    assert rc == 0 and node == -1 and cpus == []


def test_unknown_node_and_broken_tree(tmp_path):
    root = tree(tmp_path, "0000:03:00.0", -1, {0: "0-7"})        # single-socket boxes and most containers report -1
    assert query(root, "0000:03:00.0") == (0, -1, [])
    root2 = tree(tmp_path / "b", "0000:03:00.0", 2, {0: "0-7"})  # node 2 named but absent
    rc, node, _ = query(root2, "0000:03:00.0")
    assert rc != 0 and node == 2
    root3 = tree(tmp_path / "c", "0000:03:00.0", 0, {0: "0-7,x"})
    assert query(root3, "0000:03:00.0")[0] != 0


def test_null_arguments_are_refused():
    n = C.c_uint32(0)
    assert aero_amd.lib().aero_numa_parse_cpulist(None, None, C.c_uint32(0), C.byref(n)) != 0
    assert aero_amd.lib().aero_numa_query(None, b"x", None, None, C.c_uint32(0), None) != 0
    assert aero_amd.lib().aero_host_alloc_near(C.c_size_t(0), C.c_int32(0), None) != 0


def test_pinned_views_keep_their_buffer_alive():
    """PinnedTrace.array is a view over library-owned pinned memory: a view the caller still holds must outlive release() (a freed buffer
    behind a live numpy view was a use-after-free in the host process). Runs where hipHostMalloc works without a GPU, else is skipped."""
    t = np.arange(32, dtype=np.uint64).reshape(2, 16)
    try:
        p = aero_amd.PinnedTrace(t)
    except aero_amd.AeroError:
        pytest.skip("no HIP runtime that can pin host memory on this box")
    view, row = p.array, p.array[1]
    p.release()
    assert p.array is None
    assert (view == t).all() and (row == t[1]).all()             # still backed by live memory
    del p, view
    assert row[3] == t[1][3]
