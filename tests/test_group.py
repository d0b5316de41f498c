"""The multi-GPU group API (include/lumenos_hip.h "lumen_group") on ONE GPU: W contexts on device 0
play the W ranks of a single host process -- the reference's topology, one process that owns the request
(cmd/server/main.go:187-266) -- and the exchange steps run inside the library (copy transport: RCCL refuses two
ranks on one device).  Everything is held to the single-context result, itself bit-exact against the oracle.

The same file runs a second time, in a child process, through the library's RCCL branch with W = 2, 4, 8
(tests/test_group_rccl.py: LUMEN_TEST_GROUP_TRANSPORT=rccl, the test double tests/cpp/fake_rccl.cpp first on
LD_LIBRARY_PATH, the test hook lumen_test_allow_shared_device_rccl): every assertion below then holds for the grouped
ncclSend / ncclRecv, ncclAllGather and gather-to-root call sequences too."""
import os

import numpy as np
import pytest

from tests.helpers import T_REF, make_context, make_params, random_cts

pytestmark = pytest.mark.gpu

TRANSPORT = os.environ.get("LUMEN_TEST_GROUP_TRANSPORT", "copy")
FAKE = TRANSPORT == "rccl"  # the child process of tests/test_group_rccl.py
FAKE_VERSION = "99999"      # what tests/cpp/fake_rccl.cpp answers to ncclGetVersion


@pytest.fixture(scope="module")
def small(oracle):
    P = make_params(oracle, 10, 3)
    ctx = make_context(P)
    if FAKE:
        ctx.test_allow_shared_device_rccl(True)  # clones inherit it
    yield P, ctx
    ctx.close()


def check_transport(g, world):
    if FAKE:
        assert g.transport == "rccl" and g.rccl_ranks == world and FAKE_VERSION in g.transport_note, g.transport_note
    else:
        assert g.transport == "copy" and g.rccl_ranks == 0 and "one device" in g.transport_note


def ranks_of(ctx, world):
    """rank 0 is the context itself, the others its clones (same device: they share tables and keys)"""
    return [ctx] + [ctx.clone() for _ in range(world - 1)]


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_group_all_to_all_routes_blocks(small, world):
    from lumenos_amd.hip import Group
    P, ctx = small
    ctxs = ranks_of(ctx, world)
    assert all(c.device == 0 for c in ctxs)  # bench.py's join_ranks tells ranks apart by it
    g = Group(ctxs, transport=TRANSPORT)
    check_transport(g, world)
    assert g.world == world
    n, nl = 3 * world, 2
    host = [random_cts(P, n, nl, seed=300 + r) for r in range(world)]
    send = [c.upload(h) for c, h in zip(ctxs, host)]
    recv = [c.new_set(n, nl) for c in ctxs]
    g.all_to_all(send, recv)
    b = n // world
    for r in range(world):
        want = np.concatenate([host[s][r * b:(r + 1) * b] for s in range(world)])
        assert np.array_equal(recv[r].download(), want), r
    ms, sent, calls = g.stats("all_to_all")
    assert calls == 1 and sent == (world - 1) * b * 2 * nl * P.N * 8 and (ms > 0 or world == 1)
    g.close()
    for c in ctxs[1:]:
        c.close()


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_group_commit_matches_single_context(oracle, small, world):
    """Encode -> rescale -> leaf digests -> all-gather -> Merkle root -> query gather over W ranks: the same
    encoded columns, the same root, the same opened columns as one context."""
    from lumenos_amd.hip import Group
    P, ctx = small
    cols, rho, nl = 64, 2, 3
    S, c, Sw = cols * rho, cols // world, cols * rho // world
    roots = oracle.field_roots(T_REF, S)
    ctx.field_set(roots)
    m = random_cts(P, cols, nl, seed=191)
    zero = random_cts(P, 1, nl, seed=192)[0]
    full = ctx.encode(ctx.upload(m), zero, rho)
    assert np.array_equal(full.download(), P.ct_encode(m, rho, zero, roots))
    lvl1 = ctx.rescale(full, 2)
    dig = ctx.leaf_digests(lvl1)
    _, root = ctx.merkle_build(dig)
    idx = np.array([5, S - 1, 0, 5, Sw % S, 77 % S, (2 * Sw - 1) % S], dtype=np.uint32)
    opened = ctx.gather(lvl1, idx).download()

    ctxs = ranks_of(ctx, world)
    g = Group(ctxs, transport=TRANSPORT)
    check_transport(g, world)
    own = [cx.upload(m[r * c:(r + 1) * c]) for r, cx in enumerate(ctxs)]
    enc = g.encode(own, zero, rho)
    for r in range(world):
        assert enc[r].count == Sw and enc[r].log_world == 0
        assert np.array_equal(enc[r].download(), full.download(r * Sw, Sw)), r
    l1 = [cx.rescale(e, 2) for cx, e in zip(ctxs, enc)]
    for cx, s in zip(ctxs, l1):
        cx.leaf_digests_begin(s)
    g.all_gather_digests()
    assert g.merkle_root() == root
    assert np.array_equal(g.digests(S), dig)
    q = g.gather(l1, idx)
    assert q.count == len(idx) and np.array_equal(q.download(), opened)
    if world > 1:
        assert g.stats("all_to_all_1")[2] == 1 and g.stats("all_to_all_2")[2] == 1
        assert g.stats("all_gather")[2] == 1 and g.stats("gather_to_root")[2] == 1
    g.close()
    for cx in ctxs[1:]:
        cx.close()


def test_group_upload_download(small):
    """lumen_group_upload / _download: every rank's block between host and device in one call (page-locked and
    pageable buffers), equal to the per-context transfers."""
    from lumenos_amd.hip import Group, pinned_empty, pinned_free
    P, ctx = small
    ctxs = ranks_of(ctx, 4)
    g = Group(ctxs, transport=TRANSPORT)
    host = [random_cts(P, 5, 2, seed=700 + r) for r in range(4)]
    pinned = []
    for r in (0, 2):  # two ranks from page-locked memory, two from ordinary arrays
        a = pinned_empty(host[r].shape)
        a[:] = host[r]
        pinned.append(a)
        host[r] = a
    sets = [c.new_set(5, 2) for c in ctxs]
    g.upload(sets, host)
    for r in range(4):
        assert np.array_equal(sets[r].download(), host[r]), r
    back = [np.zeros_like(np.asarray(h)) for h in host]
    back[1] = pinned_empty(host[1].shape)
    g.download(sets, back)
    for r in range(4):
        assert np.array_equal(back[r], host[r]), r
    pinned_free(back[1])
    for a in pinned:
        pinned_free(a)
    g.close()
    for c in ctxs[1:]:
        c.close()


def test_group_refuses_what_it_cannot_serve(oracle, small):
    from lumenos_amd.hip import Group, LumenError
    P, ctx = small
    twin = ctx.clone()
    if FAKE:
        ctx.test_allow_shared_device_rccl(False)
    with pytest.raises(LumenError, match="RCCL needs every rank on its own device"):
        Group([ctx, twin], transport="rccl")
    with pytest.raises(LumenError, match="same context"):
        Group([ctx, ctx], transport="copy")
    g = Group([ctx, twin], transport="auto")  # two ranks on one device: copies
    assert g.transport == "copy" and g.transport_note == "stream-ordered copies on one device"
    if FAKE:
        ctx.test_allow_shared_device_rccl(True)
    a, b = ctx.new_set(4, 2), twin.new_set(6, 2)
    with pytest.raises(LumenError, match="differ in size"):
        g.all_to_all([a, b], [ctx.new_set(4, 2), twin.new_set(6, 2)])
    with pytest.raises(LumenError, match="no lumen_leaf_digests_begin job"):
        g.all_gather_digests()
    with pytest.raises(LumenError, match="no gathered digests"):
        g.merkle_root()
    g.close()
    other = make_context(make_params(oracle, 10, 2))
    with pytest.raises(LumenError, match="other parameters"):
        Group([ctx, other], transport="copy")
    other.close()
    twin.close()


def test_group_rccl_world_of_one(small):
    """RCCL in-process on the one GPU there is: the library loads librccl at run time, ncclCommInitAll over one
    device, the all-to-all is a send/recv to self and the all-gather a copy -- what a one-GPU box can check of the
    transport the W-device group uses (the W > 1 data path is the same calls with more peers)."""
    from lumenos_amd.hip import Group
    P, ctx = small
    g = Group([ctx], transport="rccl")
    assert g.transport == "rccl" and g.rccl_ranks == 1 and "librccl version" in g.transport_note
    assert (FAKE_VERSION in g.transport_note) == FAKE, g.transport_note
    host = random_cts(P, 5, 2, seed=7)
    send, recv = ctx.upload(host), ctx.new_set(5, 2)
    g.all_to_all([send], [recv])
    assert np.array_equal(recv.download(), host)
    dig = ctx.leaf_digests(send)
    ctx.leaf_digests_begin(send)
    g.all_gather_digests()
    assert np.array_equal(g.digests(5), dig) and g.merkle_root() == ctx.merkle_build(dig)[1]
    g.close()
    # one process per GPU: the unique id travels through the host, every rank joins with its own context
    uid = Group.unique_id()
    g1 = Group.join(ctx, 0, 1, uid)
    assert g1.transport == "rccl" and g1.rccl_ranks == 1
    g1.all_to_all([send], [recv])
    g1.sync()
    g1.close()


def test_group_temporaries_return_in_stream_order(small):
    """lumen_group_encode hands its four temporaries per rank back to the contexts' pools behind an event, without
    waiting for the device; the next call that takes such a block waits for the event on its stream.  Back-to-back
    encodes (which recycle the blocks while the previous call may still be running) give the same bytes as the
    first."""
    from lumenos_amd.hip import Group
    P, ctx = small
    world, cols, rho, nl = 4, 32, 2, 3
    ctx.field_set(np.asarray(ctx_roots(P, cols * rho)))
    m = random_cts(P, cols, nl, seed=77)
    zero = random_cts(P, 1, nl, seed=78)[0]
    ctxs = ranks_of(ctx, world)
    g = Group(ctxs, transport=TRANSPORT)
    own = [cx.upload(m[r * (cols // world):(r + 1) * (cols // world)]) for r, cx in enumerate(ctxs)]
    first = [e.download() for e in g.encode(own, zero, rho)]
    for _ in range(3):
        enc = g.encode(own, zero, rho)  # no sync in between: the pool hands out blocks with events attached
    for r in range(world):
        assert np.array_equal(enc[r].download(), first[r]), r
    g.close()
    for cx in ctxs[1:]:
        cx.close()


def ctx_roots(P, S):
    from oracle.loader import Oracle
    return Oracle().field_roots(T_REF, S)


def test_failed_collective_poisons_the_group(small):
    """RCCL branch only (the test double injects the fault): an ncclSend that fails BETWEEN ncclGroupStart and ncclGroupEnd
    makes the call return with the RCCL group closed -- which launches half a collective.  The group must then refuse every
    later collective (naming the first error) instead of queueing work behind streams that may never drain, and
    lumen_group_destroy must return without waiting for them.  The contexts stay usable."""
    if not FAKE:
        pytest.skip("needs the RCCL test double (tests/test_group_rccl.py runs this file through it)")
    import ctypes
    from lumenos_amd.hip import Group, LumenError
    P, ctx = small
    world = 2
    ctxs = ranks_of(ctx, world)
    g = Group(ctxs, transport=TRANSPORT)
    n, nl = 2 * world, 2
    send = [c.upload(random_cts(P, n, nl, seed=700 + r)) for r, c in enumerate(ctxs)]
    recv = [c.new_set(n, nl) for c in ctxs]
    g.all_to_all(send, recv)  # a healthy call first
    g.sync()
    ctypes.CDLL("librccl.so.1").fake_rccl_fail_send_after(1)  # the second send of the next collective
    with pytest.raises(LumenError, match="ncclSend|Send"):
        g.all_to_all(send, recv)
    for call in (lambda: g.all_to_all(send, recv), g.sync):
        with pytest.raises(LumenError, match="failed half-posted"):
            call()
    g.close()  # returns: no wait for the poisoned streams
    # the contexts themselves are fine (the double's unmatched sends sit in a mailbox, nothing blocks a stream)
    g2 = Group(ctxs, transport=TRANSPORT)
    g2.all_to_all(send, recv)
    g2.sync()
    want = send[1].download()[:n // world]
    assert np.array_equal(recv[0].download()[n // world:], want)
    g2.close()
    ctxs[1].close()
