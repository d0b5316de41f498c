"""CPU, world_size 2 (gloo): the N > 1 logic of the prover -- column shards, the one all-gather
of 32-byte leaf digests, and the Merkle root every rank then derives -- against the single-rank
result.  The per-column device work itself is rank-local and covered by the GPU parity tests."""
import os
import socket

import numpy as np
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
