"""CPU, world_size 2 (gloo): the N > 1 logic of the prover -- column shards, the one all-gather
of 32-byte leaf digests, and the Merkle root every rank then derives -- against the single-rank
result.  The per-column device work itself is rank-local and covered by the GPU parity tests."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard(n, rank, world):
    return n * rank // world, n * (rank + 1) // world


def _worker(rank, world, port, S, digests, queries, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    # interleaved, uneven ownership like the transform's final-pass groups give
    my_cols = np.array([c for c in range(S) if (c // 3) % world == rank], dtype=np.uint32)
    full = bench.all_gather_digests(dist, digests[my_cols].copy(), my_cols, S, world)
    from oracle.loader import Oracle
    o = Oracle()
    _, root = o.merkle(full)
    own = [int(my_cols[p]) for p in bench.owned_queries(queries.astype(np.uint32), my_cols)]
    cnt = torch.tensor([len(own)])
    dist.all_reduce(cnt)
    out.put((rank, root, own, int(cnt.item())))
    dist.destroy_process_group()


def test_sharded_commit_matches_single_rank(oracle):
    S, world = 64, 2
    rng = np.random.default_rng(0)
    digests = rng.integers(0, 256, size=(S, 32), dtype=np.uint8)
    queries = rng.integers(0, S, size=20)
    _, want_root = oracle.merkle(digests)
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, S, digests, queries, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = [out.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    owned = []
    for rank, root, own, total in res:
        assert root == want_root            # every rank derives the same commitment
        assert total == len(queries)        # every query index has exactly one owner
        owned += own
    assert sorted(owned) == sorted(int(q) for q in queries)


def test_shards_partition_columns():
    for n in (4096, 8192, 1000):
        for world in (1, 2, 4, 8, 3):
            spans = [_shard(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))


# ---- lane-sharded Encode: the routing of the two all-to-alls, world 2 and 4 on CPU
class _FakeCtx:
    def sync(self):
        pass


class _FakeSet:
    """what bench.all_to_all_sets needs of a DeviceSet, backed by numpy"""

    def __init__(self, arr):
        self.a = np.ascontiguousarray(arr, dtype=np.uint64)
        self.ctx = _FakeCtx()

    count = property(lambda self: self.a.shape[0])
    shape = property(lambda self: self.a.shape)
    nbytes = property(lambda self: self.a.nbytes)

    def download(self):
        return self.a.copy()

    def upload(self, host):
        self.a[...] = host


def split_np(cols, W):
    """lumen_lanes_split on the host: [n][2][nl][N] -> [W*n][2][nl][N/W], block g = lanes of rank g"""
    n, _, nl, N = cols.shape
    return np.concatenate([cols[..., g * (N // W):(g + 1) * (N // W)] for g in range(W)])


def assemble_np(lanes, W):
    """lumen_lanes_assemble on the host: [W*n][2][nl][N/W] -> [n][2][nl][N]"""
    n = lanes.shape[0] // W
    return np.concatenate([lanes[g * n:(g + 1) * n] for g in range(W)], axis=3)


def fake_encode(lanes, rho):
    """any lane-independent map cols -> rho*cols ciphertexts (the real one is tested on the GPU)"""
    acc = np.cumsum(lanes, axis=0, dtype=np.uint64)
    return np.concatenate([lanes * np.uint64(3) + np.uint64(1), acc ^ np.uint64(0x5555)])


def _lane_worker(rank, world, port, matrix, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    cols = matrix.shape[0]
    c = cols // world
    own = matrix[rank * c:(rank + 1) * c]                    # the only input this rank ever holds
    blocks = _FakeSet(split_np(own, world))
    lanes = _FakeSet(np.zeros((cols,) + blocks.shape[1:], dtype=np.uint64))
    bench.all_to_all_sets(dist, blocks, lanes, world)
    enc = _FakeSet(fake_encode(lanes.a, 2))
    recv = _FakeSet(np.zeros_like(enc.a))
    bench.all_to_all_sets(dist, enc, recv, world)
    out.put((rank, lanes.a.copy(), assemble_np(recv.a, world)))
    dist.destroy_process_group()


def _run_lane_world(world):
    rng = np.random.default_rng(world)
    cols, nl, N = 8, 2, 64
    matrix = rng.integers(0, 2**63, size=(cols, 2, nl, N), dtype=np.uint64)
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_lane_worker, args=(r, world, port, matrix, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, lanes, mine = out.get(timeout=180)
        res[r] = (lanes, mine)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = fake_encode(matrix, 2)                            # single-rank result, full width
    Nw, Sw = N // world, full.shape[0] // world
    for r in range(world):
        lanes, mine = res[r]
        assert np.array_equal(lanes, matrix[..., r * Nw:(r + 1) * Nw])   # lane shard of ALL input columns
        assert np.array_equal(mine, full[r * Sw:(r + 1) * Sw])           # whole ciphertexts of ITS encoded columns


def test_lane_sharded_exchange_world2():
    _run_lane_world(2)


def test_lane_sharded_exchange_world4():
    _run_lane_world(4)


def test_lane_sharded_exchange_world8():
    """the world the 8-GPU node runs: 8 ranks, one input column and 8 lanes each"""
    _run_lane_world(8)


def _attach_worker(rank, world, port, mode, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import argparse
    import time
    import types
    import bench
    import lumenos_amd.hip as hip

    class FakeGroup:  # the library's group as bench.join_ranks uses it; what fails, and where, is the test's `mode`
        transport = "rccl"
        transport_note = "fake"
        closed = False
        join_calls = 0

        @staticmethod
        def unique_id():
            if mode == "no_librccl_on_1" and rank == 1:
                raise hip.LumenError("dlopen(librccl.so.1) failed")
            return np.arange(128, dtype=np.uint8) + (0 if rank == 0 else 1)  # only rank 0's id may be used

        @classmethod
        def join(cls, ctx, r, w, uid):
            cls.join_calls += 1
            assert np.array_equal(uid, np.arange(128, dtype=np.uint8)), "the id rank 0 drew reaches every rank"
            if mode == "join_fails_on_1" and r == 1:
                raise hip.LumenError(f"ncclCommInitRank failed on rank {r}")
            if mode == "rank1_fails_rank0_blocks":  # the asymmetric case: a healthy rank waits inside ncclCommInitRank
                if r == 1:
                    raise hip.LumenError("ncclCommInitRank failed on rank 1")
                time.sleep(3600)
            return cls()

        def close(self):
            FakeGroup.closed = True

    hip.Group = FakeGroup

    class J:
        pass

    job = J()
    job.rank, job.world, job.group = rank, world, None
    job.ctx = types.SimpleNamespace(device=0 if mode == "shared_device" else rank)
    args = argparse.Namespace(transport="rccl", share_gpu=False)
    text = bench.attach_group(job, args, dist, new_nccl_group=lambda: "nccl-subgroup")
    out.put((rank, text, job.group is not None, getattr(job, "nccl_pg", None), FakeGroup.closed, FakeGroup.join_calls))
    dist.destroy_process_group()


def _run_attach(mode, env=None):
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        procs = [ctx.Process(target=_attach_worker, args=(r, world, port, mode, out)) for r in range(world)]
        for p in procs:
            p.start()
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    return procs, out


@pytest.mark.parametrize("mode", ["all_join", "join_fails_on_1", "no_librccl_on_1", "shared_device"])
def test_ranks_agree_on_the_transport(mode):
    """bench.attach_group / join_ranks under gloo, world 2.  What can be checked locally is agreed BEFORE anybody
    enters ncclCommInitRank: a rank without a usable librccl, or two ranks on one device, and NO rank joins.  Then the
    id of rank 0 travels over the control plane; when every rank joins, all run the in-library path; when one rank's
    join fails, ALL fall back to the torch.distributed path together (a rank left alone in the other path would wait
    in a collective for ever) and config.transport says why."""
    procs, out = _run_attach(mode)
    res = sorted(out.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, text, has_group, pg, closed, join_calls in res:
        if mode == "all_join":
            assert text.startswith("lumen_group: rccl") and has_group and pg is None and join_calls == 1
            continue
        assert "FALLBACK" in text and not has_group and pg == "nccl-subgroup", text
        if mode == "join_fails_on_1":
            assert "ncclCommInitRank failed on rank 1" in text and join_calls == 1
            assert closed == (rank != 1)  # the rank that had joined gave its group back
        elif mode == "no_librccl_on_1":
            assert "rank 1: dlopen(librccl.so.1) failed" in text and join_calls == 0
        else:
            assert "ranks [0, 1] share one device" in text and join_calls == 0


def test_a_rank_left_alone_in_the_join_ends_the_job():
    """The asymmetric failure: rank 1's ncclCommInitRank fails, rank 0's waits for a peer that will never come.  Rank 0
    must not sit there until the control plane times out: after LUMEN_BENCH_JOIN_TIMEOUT it exits non-zero (exit code
    3), which makes torch.distributed.run end the whole job."""
    procs, _ = _run_attach("rank1_fails_rank0_blocks", env={"LUMEN_BENCH_JOIN_TIMEOUT": "3"})
    procs[0].join(timeout=120)
    assert procs[0].exitcode == 3
    procs[1].join(timeout=5)  # rank 1 waits for rank 0 in the control-plane gather: the launcher would kill it
    if procs[1].is_alive():
        procs[1].terminate()
        procs[1].join(timeout=30)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` as typed: the parent starts one rank per GPU under torch.distributed.run as a
    child process, relays its output and exits with its code.  Without a GPU here every rank stops at "needs a
    HIP device" -- which proves the ranks were started with RANK / WORLD_SIZE set and that the parent reports
    their failure instead of the old "must be launched with torch.distributed.run"."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the rehearsal in profiles/ covers the real run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--config", "2048x1024", "--dist-backend", "gloo", "--share-gpu", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode != 0
    assert "needs a HIP device" in res.stderr, res.stderr[-1500:]
    assert "must be launched with" not in res.stderr
    # the parent itself never imports torch.cuda / HIP: it is a plain launcher
    src = open(os.path.join(root, "bench_lib", "multi.py")).read()
    body = src[src.index("def launch_ranks"):]
    code = body[body.index('"""', body.index('"""') + 3) + 3:]  # behind the docstring
    assert "torch.cuda" not in code and "lumenos_amd" not in code and "import torch" not in code
