"""One process per GPU (lumen_group_create_rank) with W > 1 on ONE GPU: W host threads play the W processes.

NOT collected by a plain `pytest tests` (the file name does not match test_*.py): these cases need an RCCL that lets
several ranks share the one device, i.e. the test double tests/cpp/fake_rccl.cpp, which can only be loaded into a fresh
process.  tests/test_group_rccl.py::test_group_suite_through_the_rccl_branch runs this file, together with
tests/test_group.py, in that child process."""
import threading

import numpy as np
import pytest

from tests.helpers import T_REF, random_cts
from tests.test_group import FAKE, FAKE_VERSION, ranks_of, small  # noqa: F401 -- `small` is the module fixture

pytestmark = pytest.mark.gpu
assert FAKE, "run through tests/test_group_rccl.py (LUMEN_TEST_GROUP_TRANSPORT=rccl + the test double on LD_LIBRARY_PATH)"


def run_ranks(world, body):
    """body(rank) on `world` threads (ctypes releases the GIL inside the library); re-raises the first failure"""
    errs = [None] * world

    def wrap(r):
        try:
            body(r)
        except BaseException as e:  # noqa: BLE001 -- reported below, per rank
            errs[r] = e

    ts = [threading.Thread(target=wrap, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(300)
    assert not any(t.is_alive() for t in ts), "a rank is stuck inside a collective"
    return errs


@pytest.mark.parametrize("world", [2, 4, 8])
def test_per_rank_groups_commit_matches_single_context(oracle, small, world):
    """The sequence of test_group_commit_matches_single_context with every rank in its OWN lumen_group (n = 1 local
    context of W): ncclCommInitRank from a shared id, all-to-alls whose peers are other threads, the all-gather, the
    fingerprint exchange in front of the query gather, sends from the owners to rank 0."""
    from lumenos_amd.hip import Group
    P, ctx = small
    cols, rho, nl = 64, 2, 3
    S, c, Sw = cols * rho, cols // world, cols * rho // world
    roots = oracle.field_roots(T_REF, S)
    ctx.field_set(roots)
    m = random_cts(P, cols, nl, seed=191)
    zero = random_cts(P, 1, nl, seed=192)[0]
    full = ctx.encode(ctx.upload(m), zero, rho)
    lvl1 = ctx.rescale(full, 2)
    dig = ctx.leaf_digests(lvl1)
    _, root = ctx.merkle_build(dig)
    idx = np.array([5, S - 1, 0, 5, Sw % S, 77 % S, (2 * Sw - 1) % S], dtype=np.uint32)
    opened = ctx.gather(lvl1, idx).download()
    want_enc = full.download()
    ctx.sync()

    ctxs = ranks_of(ctx, world)
    uid = Group.unique_id()
    out = [None] * world

    def body(r):
        cx = ctxs[r]
        g = Group.join(cx, r, world, uid)
        assert g.transport == "rccl" and g.rccl_ranks == world and FAKE_VERSION in g.transport_note
        assert "ncclCommInitRank %d of %d" % (r, world) in g.transport_note
        enc = g.encode([cx.upload(m[r * c:(r + 1) * c])], zero, rho)[0]
        assert np.array_equal(enc.download(), want_enc[r * Sw:(r + 1) * Sw]), r
        l1 = cx.rescale(enc, 2)
        cx.leaf_digests_begin(l1)
        g.all_gather_digests()
        assert np.array_equal(g.digests(S), dig), r  # every rank holds all W * n digests
        if r == 0:
            assert g.merkle_root() == root
        q = g.gather([l1], idx)
        if r == 0:
            assert q.count == len(idx) and np.array_equal(q.download(), opened)
        else:
            assert q is None
        assert g.stats("all_to_all_1")[2] == 1 and g.stats("all_to_all_2")[2] == 1
        assert g.stats("all_gather")[2] == 1 and g.stats("gather_to_root")[2] == 1
        out[r] = g

    errs = run_ranks(world, body)
    assert errs == [None] * world, errs
    for g in out:
        g.close()
    for cx in ctxs[1:]:
        cx.close()


def test_per_rank_gather_refuses_ranks_that_disagree_on_the_queries(small):
    """lumen_group_gather in the per-rank form: a rank that was handed other indices is told so on EVERY rank, before
    any data send is posted (real RCCL would wait for the unmatched receive for good)."""
    from lumenos_amd.hip import Group, LumenError
    P, ctx = small
    world = 2
    ctxs = ranks_of(ctx, world)
    uid = Group.unique_id()
    sets = [cx.upload(random_cts(P, 4, 2, seed=40 + r)) for r, cx in enumerate(ctxs)]
    ok = [None] * world

    def body(r):
        g = Group.join(ctxs[r], r, world, uid)
        idx = np.array([1, 6, 2] if r == 0 else [1, 6, 3], dtype=np.uint32)
        with pytest.raises(LumenError, match="other query indices"):
            g.gather([sets[r]], idx)
        good = np.array([7, 0, 7], dtype=np.uint32)  # the group is still usable afterwards
        q = g.gather([sets[r]], good)
        if r == 0:
            want = np.stack([sets[i // 4].download()[i % 4] for i in good])
            assert np.array_equal(q.download(), want)
        g.sync()
        ok[r] = g

    errs = run_ranks(world, body)
    assert errs == [None] * world, errs
    for g in ok:
        g.close()
    ctxs[1].close()




def test_per_rank_gather_bad_arguments_on_one_rank_fail_on_every_rank(small):
    """A rank whose OWN arguments are bad (an index out of range here) must not return before the agreement exchange:
    its peers are already inside the all-gather and real RCCL would keep them there for good.  Every rank fails -- the
    offender with what is wrong, the others with who rejected -- and the group stays usable."""
    from lumenos_amd.hip import Group, LumenError
    P, ctx = small
    world = 2
    ctxs = ranks_of(ctx, world)
    uid = Group.unique_id()
    sets = [cx.upload(random_cts(P, 4, 2, seed=60 + r)) for r, cx in enumerate(ctxs)]
    ok = [None] * world

    def body(r):
        g = Group.join(ctxs[r], r, world, uid)
        idx = np.array([1, 6, 2] if r == 0 else [1, 6, 99], dtype=np.uint32)  # 99: beyond 2 x 4 columns, on rank 1 only
        with pytest.raises(LumenError, match="rejected its arguments" if r == 0 else "out of range"):
            g.gather([sets[r]], idx)
        good = np.array([7, 0, 7], dtype=np.uint32)
        q = g.gather([sets[r]], good)
        if r == 0:
            assert np.array_equal(q.download(), np.stack([sets[i // 4].download()[i % 4] for i in good]))
        g.sync()
        ok[r] = g

    errs = run_ranks(world, body)
    assert errs == [None] * world, errs
    for g in ok:
        g.close()
    ctxs[1].close()
