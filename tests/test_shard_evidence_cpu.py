"""The evidence a sharded test run leaves (aero_amd/shard.py: chunk_digests, TorchComm evidence) and what the harness makes of it
(tests/test_gpu_sharded.py: exchange_mismatches, diagnose) - exercised here on synthetic exchanges, because on the GPU box this code only
runs to the end when something has already gone wrong (profiles/r5_sharded_anomaly.md had no such evidence to look at)."""
import copy
import importlib.util
import os

import pytest

torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


shard = _load("aero_shard_for_test", os.path.join(ROOT, "aero_amd", "shard.py"))
harness = _load("sharded_harness_for_test", os.path.join(ROOT, "tests", "test_gpu_sharded.py"))


def test_fingerprints_tell_a_bit_flip_a_swap_and_a_shift():
    g = torch.Generator().manual_seed(3)
    t = torch.randint(0, 256, (4 * 4096,), dtype=torch.uint8, generator=g)
    base = shard.chunk_digests(torch, t, 4)
    assert len(base) == 4 and all(len(d) == 64 for d in base) and len(set(base)) == 4
    flipped = t.clone(); flipped[4096 + 17] ^= 1
    d = shard.chunk_digests(torch, flipped, 4)
    assert [a == b for a, b in zip(base, d)] == [True, False, True, True]
    swapped = t.clone(); swapped[8:16], swapped[16:24] = t[16:24].clone(), t[8:16].clone()       # two words exchanged: the plain sum cannot see it
    assert shard.chunk_digests(torch, swapped, 4)[0] != base[0]
    rolled = t.clone(); rolled[:4096] = torch.roll(t[:4096], 8)
    assert shard.chunk_digests(torch, rolled, 4)[0] != base[0]
    assert shard.chunk_digests(torch, t.clone(), 4) == base


def _world(world=4, per=512):
    """Evidence of one commitment as `world` consistent ranks would leave it: an all-to-all of `per` bytes per peer, the root all-gather."""
    g = torch.Generator().manual_seed(11)
    send = [torch.randint(0, 256, (world * per,), dtype=torch.uint8, generator=g) for _ in range(world)]
    recv = [torch.cat([send[q][r * per:(r + 1) * per] for q in range(world)]) for r in range(world)]
    roots = [torch.randint(0, 256, (32,), dtype=torch.uint8, generator=g) for _ in range(world)]
    top = torch.cat(roots)
    ev = []
    for r in range(world):
        ev.append([{"op": "all_to_all", "bytes": per, "send": shard.chunk_digests(torch, send[r], world), "recv": shard.chunk_digests(torch, recv[r], world)},
                   {"op": "all_gather", "bytes": 32, "send": shard.chunk_digests(torch, roots[r], 1), "recv": shard.chunk_digests(torch, top, world)}])
    return ev


def test_consistent_exchanges_raise_no_flag():
    assert harness.exchange_mismatches(_world()) == []


def test_a_damaged_piece_is_named_by_receiver_and_sender():
    ev = _world()
    bad = copy.deepcopy(ev)
    bad[2][0]["recv"][1] = "00" * 32                       # what rank 2 holds of rank 1's rows is not what rank 1 sent
    m = harness.exchange_mismatches(bad)
    assert m == ["call 0 (all_to_all, 512 B per peer): rank 2 did not receive what rank 1 sent it"]
    bad = copy.deepcopy(ev)
    bad[0][1]["recv"][3] = "ff" * 32                       # rank 0's view of subtree root 3
    assert harness.exchange_mismatches(bad) == ["call 1 (all_gather, 32 B per rank): rank 0 holds a piece of rank 3 that 3 did not send"]
    short = copy.deepcopy(ev); short[1].pop()
    assert "different numbers of exchange calls" in harness.exchange_mismatches(short)[0]


def test_diagnosis_of_a_wrong_proof_names_ranks_place_and_cause(tmp_path, monkeypatch):
    monkeypatch.setenv("GRAFT_REPO_ROOT", str(tmp_path))
    want = bytes(range(200)) * 3
    wrong = bytearray(want); wrong[24 + 32 + 5] ^= 0x40      # second commitment of the proof
    per_rank = [want, bytes(wrong), want, bytes(wrong)]
    info = {"status": [{"code": 0, "msg": ""}] * 4, "evidence": _world()}
    assert harness.diagnose(want, [want] * 4, info) == ""
    text = harness.diagnose(want, per_rank, info, tmp_path, "unit test")
    assert "ranks [1, 3] of 4 differ" in text and "[[0, 2], [1, 3]]" in text
    assert "first difference at offset 61: byte 5 of commitment #1" in text
    assert "every piece arrived as its sender fingerprinted it" in text and "subtree-root fingerprints per rank" in text
    assert os.path.exists(tmp_path / "gpurun_out" / "shard_evidence" / "unit_test" / "diagnosis.txt")
    info["evidence"][3][0]["recv"][0] = "11" * 32
    text = harness.diagnose(want, per_rank, info, None, "unit test")
    assert "rank 3 did not receive what rank 0 sent it" in text
    info["evidence"] = [[], [], [], []]                    # native communicator: no fingerprints
    assert "no exchange evidence" in harness.diagnose(want, per_rank, info, None, "unit test")
