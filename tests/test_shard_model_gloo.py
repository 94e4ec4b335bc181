"""The sharding identities and the exchange pattern of the multi-GPU proof (aero_amd/csrc/prover.hip),
checked on CPU with world_size-2 and -4 `gloo` processes and the oracle's primitives:
  * rank k's LDE shard (rows j = k mod G) is the plain LDE, with blowup B/G, of the coefficients scaled by (w_N^k)^i;
  * all-to-all of leaf digests + arrival-order interleave + one subtree per rank + all-gather of the subtree roots + log2 G
    host levels reproduce the root of the tree over ALL rows;
  * FRI fold groups stay inside a coset: folding a shard with the coset's offset gives the shard of the folded layer;
  * a coset of a constraint domain larger than one shard is picked out of the all-gathered shards by global LDE row index.
The product code cannot run without a GPU (tests/test_gpu_sharded.py covers it there); this pins the algebra it relies on and
the collectives' data layout under a real process group."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import json, os, sys
    sys.path.insert(0, %(root)r)
    import numpy as np
    import torch, torch.distributed as dist
    from tests import oracle_lib

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    orc = oracle_lib.load()
    orc.set_threads(1)
    P = 0xFFFFFFFF00000001
    G, B, log_n, W, fold = world, 8, 8, 3, 8
    n, N = 1 << log_n, B << log_n
    M = N // G
    rng = np.random.default_rng(2024)                      # same stream on every rank: everybody knows the polynomials
    coeffs = (rng.integers(0, P, size=(W, n), dtype=np.uint64))
    wN = orc.root_of_unity(log_n + 3)

    # ---- 1. coset LDE: scale coefficient i by (w_N^rank)^i, extend with blowup B / G
    s = orc.pow(wN, rank)
    scale = np.array([orc.pow(s, i) for i in range(n)], dtype=np.uint64)
    local = np.stack([orc.lde(np.array([orc.mul(int(c), int(f)) for c, f in zip(coeffs[col], scale)], dtype=np.uint64), B // G)
                      for col in range(W)])
    full = np.stack([orc.lde(coeffs[col], B) for col in range(W)])
    ok_lde = bool((local == full[:, rank::G]).all())

    # ---- 2. commitment: local row digests -> all-to-all -> subtree -> all-gather of roots -> top levels
    dig = torch.from_numpy(orc.hash_rows(local).copy())     # (M, 32) uint8, local leaf t = global leaf t*G + rank
    recv = torch.empty_like(dig)
    dist.all_to_all_single(recv, dig)                       # chunk r of `dig` -> rank r; chunk k of `recv` came from rank k
    leaves = recv.numpy().reshape(G, M // G, 32).transpose(1, 0, 2).reshape(M, 32)   # leaf u'*G + k <- piece k, entry u'
    sub = orc.merkle_nodes(np.ascontiguousarray(leaves))
    roots = [torch.empty(32, dtype=torch.uint8) for _ in range(G)]
    dist.all_gather(roots, torch.from_numpy(sub[1].copy()))
    top = [None] * (2 * G)
    for r in range(G):
        top[G + r] = roots[r].numpy().tobytes()
    for i in range(G - 1, 0, -1):
        top[i] = orc.blake2s(top[2 * i] + top[2 * i + 1])
    ref_nodes = orc.merkle_nodes(orc.hash_rows(full))
    ok_root = top[1] == ref_nodes[1].tobytes()
    ok_sub = sub[1].tobytes() == ref_nodes[G + rank].tobytes()          # my subtree = node G + rank of the full tree

    # ---- 3. FRI fold locality: shard of the folded layer = fold of the shard with the coset offset 7 * w_N^rank
    layer = rng.integers(0, P, size=N, dtype=np.uint64)
    alpha = int(rng.integers(0, P, dtype=np.uint64))
    folded = orc.fri_fold(layer, fold, alpha)
    mine = orc.fri_fold(np.ascontiguousarray(layer[rank::G]), fold, orc.mul(alpha, orc.inv(orc.pow(wN, rank))))
    ok_fold = bool((mine == folded[rank::G]).all())

    # ---- 4. constraint domain larger than a shard (ce_n = 2 M): every rank evaluates H on its own coset, the shards are
    #         all-gathered and point t of the coset h_r <w_ce> (h_r = 7 w_N^rank) is global LDE row J = rank + t N / ce_n, found
    #         at [J mod G][J div G] of the gathered block
    ceN = 2 * M
    hc = rng.integers(0, P, size=ceN, dtype=np.uint64)                      # H, degree < ce_n
    h_all = orc.lde(hc, N // ceN)                                           # H over the whole LDE domain 7 <w_N>
    mine_h = torch.from_numpy(np.ascontiguousarray(h_all[rank::G]).view(np.int64).copy())
    shards = [torch.empty_like(mine_h) for _ in range(G)]
    dist.all_gather(shards, mine_h)
    shards = [t.numpy().view(np.uint64) for t in shards]
    w_ce = orc.root_of_unity(ceN.bit_length() - 1)
    h_r = orc.mul(7, orc.pow(wN, rank))
    ok_sel = True
    for t in [0, 1, 2, 3, ceN // 2 - 1, ceN // 2, ceN - 2, ceN - 1] + [int(v) for v in rng.integers(0, ceN, size=8)]:
        J = (rank + t * (N // ceN)) %% N
        x = orc.mul(h_r, orc.pow(w_ce, t))
        acc = 0
        for c in hc[::-1]:
            acc = (acc * x + int(c)) %% P
        ok_sel = ok_sel and int(shards[J %% G][J // G]) == acc
    res = [None] * world
    dist.all_gather_object(res, (ok_lde, ok_root, ok_sub, ok_fold, ok_sel))
    if rank == 0:
        print(json.dumps({"world": world, "checks": res}))
    dist.barrier()
    dist.destroy_process_group()
''')


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 4])
def test_shard_identities_under_gloo(tmp_path, world):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                          "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script)],
                         capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["world"] == world
    for rank, checks in enumerate(r["checks"]):
        assert all(checks), f"rank {rank}: (coset LDE, root, subtree, fold locality, constraint-coset selection) = {checks}"


# ---- RcclComm's failure agreement (host logic only; the library is replaced by a stand-in: there is no GPU and no second rank here) -----
class _FakeRcclLib:
    def __init__(self, preflight_rc=0, create_rc=0):
        self.preflight_rc, self.create_rc = preflight_rc, create_rc
        self.created = self.destroyed = 0

        class _Fn:
            restype = argtypes = None

            def __call__(_self, *_a):
                return b"stand-in error text"
        self.aero_rccl_last_error = _Fn()

    def aero_rccl_unique_id(self, buf):
        return self.preflight_rc

    def aero_rccl_available(self):
        return self.preflight_rc

    def aero_rccl_create(self, *a):
        self.created += 1
        return self.create_rc

    def aero_rccl_destroy(self, h):
        self.destroyed += 1

    def aero_rccl_comm(self, *a):
        return 0

    def aero_rccl_stats(self, h, out):
        return 0


@pytest.mark.parametrize("mine,peer,when", [(0, -4, "preflight"), (-4, 0, "preflight"), (0, -4, "create"), (-5, 0, "create"), (0, 0, None)])
def test_rccl_comm_raises_on_every_rank_when_one_rank_cannot_join(monkeypatch, mine, peer, when):
    """A rank that cannot take part (no librccl: status before the collective initialisation; refused ncclCommInitRank: after it)
    must make its PEERS raise too, instead of leaving them inside RCCL's bootstrap."""
    import aero_amd
    from aero_amd import shard

    class Ctx:
        h = None
    fake = _FakeRcclLib(preflight_rc=mine if when == "preflight" else 0, create_rc=mine if when == "create" else 0)
    monkeypatch.setattr(aero_amd, "lib", lambda: fake)
    calls = []

    def agree(status):            # this rank's status and the peer's: the worst of the two, as an all-gather would give
        calls.append(status)
        stage = "preflight" if len(calls) == 1 else "create"
        return min(status, peer if stage == when else 0)
    share = lambda uid: uid if uid is not None else bytes(128)
    if when is None:
        c = shard.RcclComm(Ctx(), 0, 2, share_id=share, agree=agree)
        assert calls == [0, 0] and fake.created == 1 and c.h is not None
        return
    with pytest.raises(aero_amd.AeroError) as e:
        shard.RcclComm(Ctx(), 0, 2, share_id=share, agree=agree)
    assert "not created" in str(e.value)
    if when == "preflight":
        assert fake.created == 0 and calls == [mine]              # nobody entered the collective initialisation
    else:
        assert fake.created == 1 and fake.destroyed == (1 if mine == 0 else 0)    # a communicator that did come up is released
