"""-m gpu: HIP path vs CPU oracle through the C ABI, bit-exact (integer work)."""
import numpy as np
import pytest

from helpers import T_REF, gen_primes, make_context, make_params, random_cts

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small(oracle):
    P = make_params(oracle, 10, 4)
    ctx = make_context(P)
    yield P, ctx
    ctx.close()


@pytest.mark.parametrize("log_n,num_q", [(10, 3), (11, 2), (12, 4), (13, 2), (14, 2)])
def test_limb_ntt_matches_oracle(oracle, log_n, num_q):
    P = make_params(oracle, log_n, num_q)
    ctx = make_context(P)
    cts = random_cts(P, 3, num_q, seed=log_n)
    s = ctx.upload(cts)
    ctx.set_ntt(s, inverse=False)
    got = s.download()
    for c in range(3):
        for k in range(2):
            for l in range(num_q):
                assert np.array_equal(got[c, k, l], P.limb_ntt(cts[c, k, l], l)), (c, k, l)
    ctx.set_ntt(s, inverse=True)
    assert np.array_equal(s.download(), cts)  # INTT(NTT(x)) == x
    # INTT alone against the oracle
    s2 = ctx.upload(cts)
    ctx.set_ntt(s2, inverse=True)
    got = s2.download()
    for l in range(num_q):
        assert np.array_equal(got[1, 0, l], P.limb_intt(cts[1, 0, l], l))
    ctx.close()


@pytest.mark.parametrize("S", [2, 4, 8, 16, 32, 64, 128, 256, 512])
def test_ct_ntt_matches_oracle(oracle, small, S):
    P, ctx = small
    roots = oracle.field_roots(T_REF, max(S, 16))
    ctx.field_set(roots)
    cts = random_cts(P, S, 3, seed=S)
    s = ctx.upload(cts)
    ctx.ct_ntt(s, S)
    assert np.array_equal(s.download(), P.ct_ntt(cts, S, roots))


def test_ct_ntt_three_passes(oracle):
    """size = 32768 > 128 * 128: the schedule needs a third pass, whose middle one runs IN PLACE on the
    tile-major buffer between the passes.  T_REF - 1 is divisible by 2^14 only, so this runs over another
    plaintext prime (T = 1 mod 2^16); N = 256, one limb: 134 MB of ciphertexts."""
    S = 32768
    T = gen_primes(oracle, 40, 1, 2 * S, exclude=())[0]
    assert T % (2 * S) == 1
    P = make_params(oracle, 8, 1, num_p=1, T=T)
    ctx = make_context(P)
    try:
        roots = oracle.field_roots(T, S)
        ctx.field_set(roots)
        cts = random_cts(P, S, 1, seed=77)
        s = ctx.upload(cts)
        ctx.ct_ntt(s, S)
        got = s.download()
        s.free()
    finally:
        ctx.close()
    assert np.array_equal(got, P.ct_ntt(cts, S, roots))


def test_ct_ntt_multi_chunk(oracle, small):
    """len(values) > size: `step` leaks from chunk to chunk (ntt.go:249,263)."""
    P, ctx = small
    roots = oracle.field_roots(T_REF, 128)
    ctx.field_set(roots)
    cts = random_cts(P, 128, 2, seed=5)
    for size in (16, 32, 64):
        s = ctx.upload(cts)
        ctx.ct_ntt(s, size)
        assert np.array_equal(s.download(), P.ct_ntt(cts, size, roots)), size


def test_encode_matches_oracle(oracle, small):
    P, ctx = small
    cols, rho = 64, 2
    roots = oracle.field_roots(T_REF, cols * rho)
    ctx.field_set(roots)
    m = random_cts(P, cols, 4, seed=11)
    zero = random_cts(P, 1, 4, seed=12)[0]
    before = ctx.mul_counter()
    enc = ctx.encode(ctx.upload(m), zero, rho)
    assert np.array_equal(enc.download(), P.ct_encode(m, rho, zero, roots))
    assert ctx.mul_counter() - before == len(oracle.twiddle_trace(cols * rho, cols * rho))


def test_rescale_matches_oracle(oracle, small):
    P, ctx = small
    cts = random_cts(P, 5, 4, seed=21)
    s = ctx.upload(cts)
    for target in (3, 2, 1):
        got = ctx.rescale(s, target).download()
        for c in range(5):
            ref = cts[c]
            while ref.shape[1] > target:
                ref = P.rescale(ref)
            assert np.array_equal(got[c], ref), (target, c)


def test_leaf_digests_async_matches_sync(small):
    """lumen_leaf_digests_begin/_end (side stream, overlapped with later work) == lumen_leaf_digests."""
    P, ctx = small
    s = ctx.new_set(37, 2).fill_random(11)
    want = ctx.leaf_digests(s)
    ctx.leaf_digests_begin(s)
    other = ctx.new_set(8, 2).fill_random(12)  # unrelated work on the main stream meanwhile
    ctx.set_ntt(other, False)
    got = ctx.leaf_digests_end()
    assert np.array_equal(got, want)
    other.free()
    s.free()


def test_leaf_digests_and_merkle(oracle, small):
    P, ctx = small
    cts = random_cts(P, 37, 2, seed=31)  # odd count: unpaired node duplicated (tree.go:127-131)
    s = ctx.upload(cts)
    dig = ctx.leaf_digests(s)
    for c in range(37):
        assert dig[c].tobytes() == oracle.sha256(P.ct_serialize(cts[c])), c
    nodes, root = ctx.merkle_build(dig)
    onodes, oroot = oracle.merkle(dig)
    assert root == oroot and np.array_equal(nodes, onodes)


@pytest.mark.parametrize("lens", [(0, 0, 0), (8, 8, 8), (233, 8, 8), (1, 3, 5), (63, 64, 7), (1024, 2, 0), (130, 0, 61)])
def test_leaf_format_any_byte_alignment(oracle, small, lens):
    """The serialisation layout is a parameter (lumen_leaf_format_set): head | per polynomial poly_head |
    per limb limb_head | raw limbs.  Lattigo's MetaData block has no reason to be a multiple of 4 or 8
    bytes, so the limb data may sit at any byte offset of SHA-256's blocks: digests and serialised bytes
    must equal the oracle's for every alignment, for one and two limbs, and the default framing returns
    when the format is cleared."""
    P, ctx = small
    rng = np.random.default_rng(sum(lens) + 1)
    head, poly, limb = (bytes(rng.integers(0, 256, size=n, dtype=np.uint8)) for n in lens)
    try:
        ctx.leaf_format_set(head, poly, limb)
        for nl in (2, 1):
            cts = random_cts(P, 67, nl, seed=61 + nl)  # more than one wave, not a multiple of 64
            s = ctx.upload(cts)
            dig = ctx.leaf_digests(s)
            blob = ctx.ct_serialize(s, 3, 2)
            want = [P.ct_serialize(cts[c], (head, poly, limb)) for c in range(67)]
            assert blob == want[3] + want[4]
            # the wire image of the whole slice -- what EncryptedProof.WriteTo emits for MatR (ligero.go:664-671) --
            # is assembled on the device: every ciphertext starts where the previous one ends, at any alignment
            assert ctx.ct_serialize(s) == b"".join(want)
            # and back (ct.ReadFrom for a whole slice): the image taken apart on the device, framing checked
            back = ctx.ct_deserialize(b"".join(want), 67, nl)
            assert np.array_equal(back.download(), cts)
            if sum(lens):
                from lumenos_amd.hip import LumenError
                bad = bytearray(b"".join(want))
                pos = 5 * len(want[0]) + (0 if lens[0] else len(want[0]) - 8 * P.N - 1)  # a header byte of ciphertext 5
                bad[pos] ^= 0x01
                with pytest.raises(LumenError, match="differ from the serialisation format"):
                    ctx.ct_deserialize(bytes(bad), 67, nl)
            with pytest.raises(Exception, match="bytes in the current format"):
                ctx.ct_deserialize(b"".join(want)[:-1], 67, nl)
            # ... and the asynchronous form into page-locked memory at an odd offset, on a clone's stream
            from lumenos_amd.hip import pinned_bytes
            each = ctx.ct_serialized_size(nl)
            assert each == len(want[0])
            buf = pinned_bytes(11 + 67 * each)
            buf[:] = 0xEE
            twin = ctx.clone()
            ctx.sync()
            twin.ct_serialize_into(s, buf, offset=11, wait=False)
            twin.sync()
            assert buf[:11].tobytes() == b"\xee" * 11 and buf[11:].tobytes() == b"".join(want)
            twin.close()
            assert len(want[0]) == sum(lens[:1]) + 2 * (lens[1] + nl * (lens[2] + 8 * P.N))
            for c in range(67):
                assert dig[c].tobytes() == oracle.sha256(want[c]), (nl, c)
            ctx.leaf_digests_begin(s)
            assert np.array_equal(ctx.leaf_digests_end(), dig)
    finally:
        ctx.leaf_format_set()
    cts = random_cts(P, 2, 2, seed=7)
    assert ctx.leaf_digests(ctx.upload(cts))[1].tobytes() == oracle.sha256(P.ct_serialize(cts[1]))
    with pytest.raises(Exception):
        ctx.leaf_format_set(b"x" * 1025, b"", b"")


def test_set_destroyed_while_its_leaves_are_hashed(small):
    """A set freed between lumen_leaf_digests_begin and _end: the destroy waits for the side stream's job
    (it reads that storage); the digests are the synchronous ones and a set created right after may reuse
    the storage safely."""
    P, ctx = small
    cts = random_cts(P, 300, 2, seed=71)
    s = ctx.upload(cts)
    want = ctx.leaf_digests(s)
    ctx.leaf_digests_begin(s)
    s.free()
    t = ctx.new_set(300, 2).fill_random(5)  # same size: takes the block the pool just got back
    got = ctx.leaf_digests_end()
    assert np.array_equal(got, want)
    t.free()


def test_gather(oracle, small):
    P, ctx = small
    cts = random_cts(P, 9, 2, seed=41)
    idx = np.array([3, 3, 0, 8, 5], dtype=np.uint32)
    assert np.array_equal(ctx.gather(ctx.upload(cts), idx).download(), cts[idx])


@pytest.fixture(scope="module")
def keyed(oracle):
    P = make_params(oracle, 10, 5)
    P.seed(99)
    sk = P.keygen_secret()
    ctx = make_context(P)
    yield P, ctx, sk
    ctx.close()


def test_mul_plain_matches_oracle(oracle, keyed):
    P, ctx, sk = keyed
    cts = random_cts(P, 3, 5, seed=51)
    pt = P.encode(np.arange(1, P.N + 1, dtype=np.uint64))
    got = ctx.mul_plain(ctx.upload(cts), pt).download()
    for c in range(3):
        assert np.array_equal(got[c], P.mul_plain(cts[c], pt))


@pytest.mark.parametrize("n", [8, 512, 1024])
def test_inner_sum_matches_oracle(oracle, keyed, n):
    P, ctx, sk = keyed
    gl = P.inner_sum_galois_elements(n)
    assert ctx.inner_sum_galois_elements(n) == gl
    evks = [P.keygen_galois(sk, g) for g in gl]
    for g, e in zip(gl, evks):
        ctx.load_galois_key(g, e)
    cts = random_cts(P, 3, 5, seed=n)
    got = ctx.inner_sum(ctx.upload(cts), n).download()
    for c in range(3):
        assert np.array_equal(got[c], P.inner_sum(cts[c], n, evks)), c


def test_galois_keys_in_lattigo_montgomery_form(oracle, keyed):
    """lumen_load_galois_key_ex(LUMEN_KEY_MONTGOMERY): the key words as Lattigo stores them (x * 2^64 mod q_i,
    GadgetCiphertext.Value copied as it is) give the same InnerSum as the standard-form key -- the conversion runs on
    the device either way -- and a residue >= q_i is still refused, with its digit and limb."""
    from lumenos_amd.hip import LumenError
    P, ctx, sk = keyed
    n = 16
    gl = P.inner_sum_galois_elements(n)
    evks = [P.keygen_galois(sk, g) for g in gl]
    cts = random_cts(P, 2, 5, seed=1234)
    for g, e in zip(gl, evks):
        ctx.load_galois_key(g, e)
    want = ctx.inner_sum(ctx.upload(cts), n).download()
    assert np.array_equal(want[0], P.inner_sum(cts[0], n, evks))
    for g, e in zip(gl, evks):
        m = np.empty_like(e)
        for t, q in enumerate(P.moduli):
            m[:, :, t, :] = ((e[:, :, t, :].astype(object) << 64) % int(q)).astype(np.uint64)
        ctx.load_galois_key(g, m, montgomery=True)
    assert np.array_equal(ctx.inner_sum(ctx.upload(cts), n).download(), want)
    bad = evks[0].copy()
    bad[1, 0, 2, 7] = int(P.moduli[2])
    with pytest.raises(LumenError, match=r"key residue out of range \(digit 1 limb 2\)"):
        ctx.load_galois_key(gl[0], bad)
    for g, e in zip(gl, evks):  # the fixture's keys as the other tests expect them
        ctx.load_galois_key(g, e)
    assert np.array_equal(ctx.inner_sum(ctx.upload(cts), n).download(), want)


@pytest.mark.parametrize("nf", [0, 1, 2])
def test_inner_sum_fused_digit_packing(oracle, keyed, nf):
    """k_intt_pack: the (hi, lo) packing of the first nf two-limb digits is fused into the c1 inverse
    transform, the others go through k_pack_v.  The library picks nf from the batch size (4 of 6 digits at
    64 columns); here every split of L = 5 limbs -- digits (0,1) (2,3) (4) -- is forced in turn."""
    P, ctx, sk = keyed
    n = 64
    gl = P.inner_sum_galois_elements(n)
    evks = [P.keygen_galois(sk, g) for g in gl]
    for g, e in zip(gl, evks):
        ctx.load_galois_key(g, e)
    cts = random_cts(P, 3, 5, seed=640 + nf)
    try:
        ctx.set_tuning("LUMEN_KS_FUSED_DIGITS", nf)
        got = ctx.inner_sum(ctx.upload(cts), n).download()
    finally:
        ctx.set_tuning("LUMEN_KS_FUSED_DIGITS", -1)
    for c in range(3):
        assert np.array_equal(got[c], P.inner_sum(cts[c], n, evks)), c


def test_matrix_inner_sum_matches_oracle_and_decrypts(oracle, keyed):
    """matrixInnerSumEval (ligero.go:299-370): bit-exact vs oracle, and slot 0 decrypts to sum_i pt_i*M[i][j]."""
    P, ctx, sk = keyed
    rows, cols = 512, 6
    pk = P.keygen_public(sk)
    W = oracle.witness(rows, cols, T_REF)
    cts = np.stack([P.encrypt(pk, P.encode(W[:, j])) for j in range(cols)])
    r = np.random.default_rng(7).integers(0, 2**63, size=rows, dtype=np.uint64)
    pt = P.encode(r)
    gl = P.inner_sum_galois_elements(rows)
    evks = [P.keygen_galois(sk, g) for g in gl]
    for g, e in zip(gl, evks):
        ctx.load_galois_key(g, e)
    got = ctx.matrix_inner_sum(ctx.upload(cts), pt, rows).download()
    ref = P.matrix_inner_sum(cts, pt, rows, evks)
    assert np.array_equal(got, ref)
    scale = P.rescale_scale(P.L, 2)
    for j in range(cols):
        want = int(np.sum(W[:, j].astype(object) * (r.astype(object) % T_REF)) % T_REF)
        assert int(P.decrypt(sk, got[j], 1, scale)[0]) == want, j


# ------------------------------------------------------------------ golden fixtures
import os  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _ctx_from_golden(g):
    from lumenos_amd.hip import Context
    return Context(int(g["log_n"]), [int(x) for x in g["q"]], [int(x) for x in g["p"]],
                   [int(x) for x in g["psi"]], int(g["T"]))


@pytest.mark.parametrize("S", [16, 32, 64])
def test_golden_encode_gpu(S):
    g = np.load(os.path.join(GOLD, f"encode_S{S}.npz"))
    ctx = _ctx_from_golden(g)
    ctx.field_set(g["roots"])
    assert np.array_equal(ctx.encode(ctx.upload(g["matrix"]), g["zero"], 2).download(), g["encoded"])
    ctx.close()


def test_golden_evaluator_gpu():
    g = np.load(os.path.join(GOLD, "evaluator.npz"))
    ctx = _ctx_from_golden(g)
    s = ctx.upload(g["cts"])
    lvl1 = ctx.rescale(s, 2)
    assert np.array_equal(lvl1.download(), g["level1"])
    assert np.array_equal(ctx.leaf_digests(lvl1), g["digests"])
    for ge, evk in zip(g["gal_els"], g["evks"]):
        ctx.load_galois_key(int(ge), evk)
    assert np.array_equal(ctx.matrix_inner_sum(s, g["pt"], int(g["rows"])).download(), g["matrix_inner_sum"])
    ctx.close()


# ------------------------------------------------------------------ BASELINE.json full sizes
@pytest.mark.parametrize("S", [2048, 4096, 8192])
def test_encode_full_ciphertext_count(oracle, S):
    """The README shapes' transform sizes (S = 2*cols) on a narrow ring: every lane replays the
    same DAG, so a small N exercises the full schedule (two HBM passes, leaking `step`)."""
    P = make_params(oracle, 8, 2, num_p=0)
    ctx = make_context(P)
    roots = oracle.field_roots(T_REF, S)
    ctx.field_set(roots)
    m = random_cts(P, S // 2, 2, seed=S)
    zero = random_cts(P, 1, 2, seed=S + 1)[0]
    got = ctx.encode(ctx.upload(m), zero, 2).download()
    assert np.array_equal(got, P.ct_encode(m, 2, zero, roots))
    ctx.close()


@pytest.fixture(scope="module")
def full_d(oracle):
    """16384x4096 / LogN=14 parameters (12 Q limbs, 2 P limbs) exactly as bench.py builds them."""
    from lumenos_amd import params as lp
    from oracle.loader import Params
    B = lp.generate_bgv_params_for_ntt(4096, 14)
    P = Params.from_moduli(oracle, 14, B.q, B.p, B.T)
    assert P.psi == B.psi
    P.seed(14)
    ctx = make_context(P)
    yield P, ctx
    ctx.close()


def test_full_size_ntt_roundtrip_and_linearity(full_d):
    P, ctx = full_d
    a = random_cts(P, 2, P.L, seed=1)
    b = random_cts(P, 2, P.L, seed=2)
    q = np.array(P.moduli[:P.L], dtype=np.uint64)[None, None, :, None]
    ab = (a + b) % q  # values < 2^59: no 64-bit wrap
    sa, sb, sab = ctx.upload(a), ctx.upload(b), ctx.upload(ab)
    for s in (sa, sb, sab):
        ctx.set_ntt(s, inverse=False)
    A, Bv, AB = sa.download(), sb.download(), sab.download()
    assert np.array_equal((A + Bv) % q, AB)                       # linearity
    assert np.array_equal(A[0, 0, 3], P.limb_ntt(a[0, 0, 3], 3))  # one limb against the oracle
    ctx.set_ntt(sa, inverse=True)
    assert np.array_equal(sa.download(), a)                       # INTT(NTT(x)) == x


def test_full_size_rescale_and_digest(oracle, full_d):
    P, ctx = full_d
    cts = random_cts(P, 2, P.L, seed=3)
    lvl1 = ctx.rescale(ctx.upload(cts), 2)
    ref_l1, ref_dig = P.commit_leaves(cts)
    assert np.array_equal(lvl1.download(), ref_l1)
    assert np.array_equal(ctx.leaf_digests(lvl1), ref_dig)


def test_full_size_matrix_inner_sum(oracle, full_d):
    """rows = N = 16384: 13 column rotations + the row swap, real keys."""
    P, ctx = full_d
    rows = 16384
    sk = P.keygen_secret()
    pk = P.keygen_public(sk)
    gl = P.inner_sum_galois_elements(rows)
    assert len(gl) == 14 and gl[-1] == 2 * P.N - 1
    evks = [P.keygen_galois(sk, g) for g in gl]
    for g, e in zip(gl, evks):
        ctx.load_galois_key(g, e)
    rng = np.random.default_rng(4)
    col = rng.integers(0, T_REF, size=rows, dtype=np.uint64)
    r = rng.integers(0, 2**63, size=rows, dtype=np.uint64)
    cts = P.encrypt(pk, P.encode(col))[None]
    pt = P.encode(r)
    got = ctx.matrix_inner_sum(ctx.upload(cts), pt, rows).download()
    assert np.array_equal(got, P.matrix_inner_sum(cts, pt, rows, evks))
    want = int(np.sum(col.astype(object) * (r.astype(object) % T_REF)) % T_REF)
    assert int(P.decrypt(sk, got[0], 1, P.rescale_scale(P.L, 2))[0]) == want


def test_full_size_inner_sum_batch_of_64_fused_packing(oracle, full_d):
    """The headline geometry of step 1 of a rotation: 64 columns at N = 2^14, L = 12 -- the batch size at
    which k_intt_pack fuses the packing of 4 of the 6 digits into the c1 transform (256 two-transform
    workgroups + 256 single ones).  Same 64 random ciphertexts through the fused launch (the library's
    choice) and with the fusion forced off; two of the columns against the oracle."""
    P, ctx = full_d
    n = 16384
    sk = P.keygen_secret()
    gl = P.inner_sum_galois_elements(n)
    evks = [P.keygen_galois(sk, g) for g in gl]
    for g, e in zip(gl, evks):
        ctx.load_galois_key(g, e)
    cts = random_cts(P, 64, P.L, seed=6464)
    ctx.set_tuning("LUMEN_KS_FUSED_DIGITS", -1)
    fused = ctx.inner_sum(ctx.upload(cts), n).download()
    ctx.set_tuning("LUMEN_KS_FUSED_DIGITS", 0)
    plain = ctx.inner_sum(ctx.upload(cts), n).download()
    ctx.set_tuning("LUMEN_KS_FUSED_DIGITS", -1)
    assert np.array_equal(fused, plain)
    for c in (0, 63):
        assert np.array_equal(fused[c], P.inner_sum(cts[c], n, evks)), c


@pytest.mark.parametrize("world", [2, 3, 8])
def test_encode_shards_cover_full_encode(oracle, small, world):
    """Multi-GPU Commit: the union of the per-rank shards (run one after the other on this one GPU)
    is exactly fhe.Encode's output, each column owned once."""
    P, ctx = small
    cols, rho = 128, 2
    S = cols * rho
    roots = oracle.field_roots(T_REF, S)
    ctx.field_set(roots)
    m = random_cts(P, cols, 2, seed=77)
    zero = random_cts(P, 1, 2, seed=78)[0]
    dm = ctx.upload(m)
    full = ctx.encode(dm, zero, rho).download()
    assert np.array_equal(full, P.ct_encode(m, rho, zero, roots))
    seen = np.zeros(S, dtype=int)
    for rank in range(world):
        shard, idx = ctx.encode_shard(dm, zero, rho, rank, world)
        assert np.all(np.diff(idx.astype(np.int64)) > 0)
        assert np.array_equal(shard.download(), full[idx])
        seen[idx] += 1
    assert np.all(seen == 1)

def test_encode_and_in_place_transform_three_limbs_256_slots(oracle, small):
    """fhe.Encode of 64 columns (S = 128) on three limbs and fhe.NTT in place on 256 two-limb ciphertexts against the
    oracle (the shapes on which rounds 2-5 also ran the op-by-op interpreter the register-blocked kernel replaced; the
    interpreter left the library in round 6 and this comparison is what its A/B duplicated)."""
    P, ctx = small
    cols, rho, nl = 64, 2, 3
    roots = oracle.field_roots(T_REF, 256)
    ctx.field_set(roots)
    m = random_cts(P, cols, nl, seed=191)
    zero = random_cts(P, 1, nl, seed=192)[0]
    assert np.array_equal(ctx.encode(ctx.upload(m), zero, rho).download(), P.ct_encode(m, rho, zero, roots))
    cts = random_cts(P, 256, 2, seed=193)
    s = ctx.upload(cts)
    ctx.ct_ntt(s, 256)
    assert np.array_equal(s.download(), P.ct_ntt(cts, 256, roots))


@pytest.mark.parametrize("world", [2, 4, 8])
def test_encode_lane_shard_alltoall_matches_full(oracle, small, world):
    """SURVEY 8e, every rank played in turn on this one GPU: rank r uploads only ITS input columns, cuts
    them into lane blocks (lumen_lanes_split), an all-to-all gives every rank the lane shard of ALL
    columns, lumen_encode runs on the shard (the transform never mixes lanes), a second all-to-all and
    lumen_lanes_assemble give every rank whole ciphertexts of its block of encoded columns.  Each step is
    compared with the single-GPU fhe.Encode (itself bit-exact against the oracle): nothing is replicated
    and a rank holds 1/W of the input."""
    P, ctx = small
    cols, rho, nl = 64, 2, 2
    S, logw = cols * rho, world.bit_length() - 1
    Nw, c, Sw = P.N >> logw, cols // world, S // world
    roots = oracle.field_roots(T_REF, S)
    ctx.field_set(roots)
    m = random_cts(P, cols, nl, seed=91)
    zero = random_cts(P, 1, nl, seed=92)[0]
    full = ctx.encode(ctx.upload(m), zero, rho).download()
    assert np.array_equal(full, P.ct_encode(m, rho, zero, roots))
    # all-to-all #1: source r sends block g of its split columns to rank g
    split = [ctx.lanes_split(ctx.upload(m[r * c:(r + 1) * c]), logw) for r in range(world)]
    assert all(sp.log_world == logw and sp.shape == (world * c, 2, nl, Nw) for sp in split)
    split = [sp.download() for sp in split]
    enc_lanes = []
    for g in range(world):
        lanes_in = np.concatenate([split[r][g * c:(g + 1) * c] for r in range(world)])  # ct-major: all columns
        assert np.array_equal(lanes_in, m[..., g * Nw:(g + 1) * Nw])
        e = ctx.encode(ctx.upload_lanes(lanes_in, logw), np.ascontiguousarray(zero[..., g * Nw:(g + 1) * Nw]), rho)
        assert e.log_world == logw and e.count == S
        enc_lanes.append(e.download())
        assert np.array_equal(enc_lanes[g], full[..., g * Nw:(g + 1) * Nw]), g
    # all-to-all #2: rank h receives block h of every rank's encoded lane shard (a contiguous slice)
    for h in range(world):
        recv = np.concatenate([enc_lanes[g][h * Sw:(h + 1) * Sw] for g in range(world)])
        got = ctx.lanes_assemble(ctx.upload_lanes(recv, logw))
        assert got.log_world == 0 and np.array_equal(got.download(), full[h * Sw:(h + 1) * Sw]), h
    # full-width entry points refuse a lane set instead of mis-reading it
    with pytest.raises(Exception, match="lane-sharded"):
        ctx.rescale(ctx.upload_lanes(recv, logw), 1)


def test_device_merkle_root_and_device_digests(oracle, small):
    """The multi-GPU tail: digests stay in HBM (lumen_leaf_digests_end_device: what the all-gather reads),
    the Merkle root is built on the device from them; equal to the host tree for even, odd and
    non-power-of-two leaf counts."""
    import ctypes as C
    P, ctx = small
    for count in (1, 2, 37, 64, 300):
        cts = random_cts(P, count, 2, seed=100 + count)
        s = ctx.upload(cts)
        dig = ctx.leaf_digests(s)
        _, want = ctx.merkle_build(dig)
        assert want == oracle.merkle(dig)[1]
        ctx.leaf_digests_begin(s)
        ptr, n = ctx.leaf_digests_end_device()
        assert n == count
        assert ctx.merkle_root_device(ptr, count) == want, count


@pytest.mark.parametrize("log_n,logn_small,num_p", [(10, 10, 2), (12, 10, 2), (12, 8, 2), (14, 10, 2),
                                                    (12, 10, 1), (13, 11, 1), (12, 10, 0), (11, 8, 0)])
def test_ring_switch_matches_oracle(oracle, log_n, logn_small, num_p):
    """RingSwitchNew (fhe/ring_switch.go:106-113): bit-exact vs the oracle on the three gadget paths Lattigo
    takes by the key's LevelP (2 special primes: one hybrid RNS digit, the reference's configurations;
    1: base-2^13 digits + ModDown; 0: base-2^13 digits, no ModDown), and the sub-ring contract: the
    small-ring ciphertext decrypts (under skNew) to the coefficients X^(i*N/n) of the input's plaintext.
    T as in TestRingSwitch (ring_switch_test.go:17): with T ~ 2^57 a single 58-bit limb leaves no room for
    noise.  The library is handed the WHOLE key the client posts and reads RNS digit 0 of it."""
    T = 0x3EE0001
    P = make_params(oracle, log_n, 3, num_p=num_p, T=T)
    P.seed(log_n * 100 + logn_small)
    sk, ctx = P.keygen_secret(), make_context(P)
    pk = P.keygen_public(sk)
    rng = np.random.default_rng(9)
    cts = np.stack([P.rescale_to_level1(P.encrypt(pk, P.encode(rng.integers(0, T, size=P.N, dtype=np.uint64))))
                    for _ in range(3)])
    sk_small = P.keygen_secret_small(logn_small)
    key = P.keygen_ringswitch(sk, sk_small, logn_small)
    assert ctx.ringswitch_key_shape() == key.shape == (*P.rs_key_shape(), 2, P.L + P.K, P.N)
    assert ctx.lib.lumen_ringswitch_digits(ctx.h, 13) == P.rs_num_digits() == (1 if num_p == 2 else 5)
    ctx.load_ringswitch_key(logn_small, key)
    got = ctx.ring_switch(ctx.upload(cts))
    gap = P.N >> logn_small
    for c in range(3):
        assert np.array_equal(got[c], P.ring_switch(cts[c], key, logn_small)), c
        assert np.array_equal(P.decrypt_small_coeffs(sk_small, logn_small, got[c]),
                              P.decrypt_big_coeffs_l0(sk, cts[c])[::gap])
    # RNS digit 0 alone is enough (ApplyEvaluationKey works at level 0)
    ctx.load_ringswitch_key(logn_small, key[0])
    assert np.array_equal(ctx.ring_switch(ctx.upload(cts[:1]))[0], got[0])
    ctx.close()


def test_ring_switch_reference_test_twin(oracle):
    """TestRingSwitch (fhe/ring_switch_test.go:13-77) through the C ABI: LogN = 12 -> 12, LogQ = [58], no
    special prime, T = 0x3ee0001; m = [1, 1] encrypted under pk at level 0; RingSwitch; decrypt under skNew
    and decode: mCheck == m.  (Keys, encryption and decryption are the oracle's -- the client side of the
    test; the switch is the device's and bit-equal to the oracle's.)"""
    T = 0x3EE0001
    P = make_params(oracle, 12, 1, num_p=0, T=T)
    P.seed(13)
    sk, ctx = P.keygen_secret(), make_context(P)
    pk = P.keygen_public(sk)
    m = np.array([1, 1], dtype=np.uint64)
    ct = P.encrypt(pk, P.encode(m))
    sk_new = P.keygen_secret_small(12)
    key = P.keygen_ringswitch(sk, sk_new, 12)
    assert ctx.ringswitch_key_shape() == key.shape == (1, 5, 2, 1, P.N)
    ctx.load_ringswitch_key(12, key)
    ct2 = ctx.ring_switch(ctx.upload(ct[None]))[0]
    assert np.array_equal(ct2, P.ring_switch(ct, key, 12))
    assert np.array_equal(P.decode_coeffs(P.decrypt_small_coeffs(sk_new, 12, ct2), 1, 2), m)
    ctx.close()


# ------------------------------------------------------------------ lazy-arithmetic headroom
def _ntt_primes_near(limit, two_n, count, down=True):
    """`count` primes == 1 mod 2N just below (or from) `limit`."""
    from lumenos_amd import params as lp
    p = limit - ((limit - 1) % two_n) if down else limit + ((1 - limit) % two_n)
    out = []
    while len(out) < count:
        if lp.is_prime(p):
            out.append(p)
        p += -two_n if down else two_n
    return out


def _adversarial_cts(P, nl, seed):
    """Rows: all q-1, all 0, alternating q-1/0, single spike, uniform random."""
    cts = random_cts(P, 3, nl, seed=seed)
    for l in range(nl):
        q = P.moduli[l]
        cts[0, 0, l, :] = q - 1
        cts[0, 1, l, :] = 0
        cts[1, 0, l, ::2], cts[1, 0, l, 1::2] = q - 1, 0
        cts[1, 1, l, :] = 0
        cts[1, 1, l, P.N - 1] = q - 1
    return cts


@pytest.mark.parametrize("log_n", [10, 12, 13, 14])
def test_limb_ntt_largest_moduli_adversarial_inputs(oracle, log_n):
    """Moduli right under the context's bound (3*logN+8)*q < 2^64 and a 21-bit one, inputs that
    maximise every lazy intermediate (all q-1): forward/inverse transforms stay bit-exact."""
    from oracle.loader import Params
    two_n = 2 << log_n
    qmax = (2**64 - 1) // (3 * log_n + 8)
    q = _ntt_primes_near(qmax, two_n, 2) + _ntt_primes_near(1 << 20, two_n, 1, down=False)
    P = Params.from_moduli(oracle, log_n, q, [], T_REF)
    ctx = make_context(P)
    cts = _adversarial_cts(P, 3, seed=log_n)
    s = ctx.upload(cts)
    ctx.set_ntt(s, inverse=False)
    got = s.download()
    for c in range(3):
        for k in range(2):
            for l in range(3):
                assert np.array_equal(got[c, k, l], P.limb_ntt(cts[c, k, l], l)), (c, k, l)
    ctx.set_ntt(s, inverse=True)
    assert np.array_equal(s.download(), cts)
    s2 = ctx.upload(cts)
    ctx.set_ntt(s2, inverse=True)
    got = s2.download()
    for c in range(2):
        for l in range(3):
            assert np.array_equal(got[c, 0, l], P.limb_intt(cts[c, 0, l], l)), (c, l)
    ctx.close()


def test_key_switch_and_rescale_largest_moduli(oracle):
    """The whole evaluator chain (MulNew, InnerSum with its key switches, Rescale to level 1) on a
    modulus chain right under the bound, with the adversarial rows among the inputs."""
    from oracle.loader import Params
    log_n, two_n = 10, 2 << 10
    qmax = (2**64 - 1) // (3 * log_n + 8)
    pr = _ntt_primes_near(qmax, two_n, 7)
    P = Params.from_moduli(oracle, log_n, pr[:5], pr[5:], T_REF)
    P.seed(5)
    sk = P.keygen_secret()
    ctx = make_context(P)
    n = 16
    gl = P.inner_sum_galois_elements(n)
    evks = [P.keygen_galois(sk, g) for g in gl]
    for g, e in zip(gl, evks):
        ctx.load_galois_key(g, e)
    cts = _adversarial_cts(P, 5, seed=3)
    pt = P.encode(np.arange(1, P.N + 1, dtype=np.uint64))
    got = ctx.matrix_inner_sum(ctx.upload(cts), pt, n).download()
    assert np.array_equal(got, P.matrix_inner_sum(cts, pt, n, evks))
    lvl1 = ctx.rescale(ctx.upload(cts), 1).download()
    for c in range(3):
        ref = cts[c]
        while ref.shape[1] > 1:
            ref = P.rescale(ref)
        assert np.array_equal(lvl1[c], ref), c
    ctx.close()


@pytest.mark.parametrize("nq,npr", [(3, 2), (2, 2), (1, 1), (3, 1)])
def test_lazy_accumulator_at_the_modulus_bound(oracle, nq, npr):
    """The InnerSum accumulator is lazy in [0, 2q) across the rotations (k_moddown_ntt; 8q < 2^64 is part of the modulus
    bound lumen_ctx_create enforces).  Its three ways out, with moduli right under that bound and all-(q-1) rows among
    the inputs: L = 3 (one limb dropped: the single-limb rescale kernels read the lazy words), L <= 2 (nothing to drop:
    k_acc_canon), and lumen_inner_sum (k_acc_canon on the way out)."""
    from oracle.loader import Params
    log_n, two_n = 10, 2 << 10
    qmax = (2**64 - 1) // (3 * log_n + 8)
    pr = _ntt_primes_near(qmax, two_n, nq + npr)
    P = Params.from_moduli(oracle, log_n, pr[:nq], pr[nq:], T_REF)
    P.seed(11)
    sk = P.keygen_secret()
    ctx = make_context(P)
    n = 32
    gl = P.inner_sum_galois_elements(n)
    evks = [P.keygen_galois(sk, g) for g in gl]
    for g, e in zip(gl, evks):
        ctx.load_galois_key(g, e)
    cts = _adversarial_cts(P, nq, seed=13)
    pt = P.encode(np.arange(1, P.N + 1, dtype=np.uint64))
    d = ctx.upload(cts)
    assert np.array_equal(ctx.matrix_inner_sum(d, pt, n).download(), P.matrix_inner_sum(cts, pt, n, evks))
    got = ctx.inner_sum(d, n).download()
    assert np.array_equal(got, np.stack([P.inner_sum(c, n, evks) for c in cts]))
    assert all(int(got[:, :, l].max()) < P.moduli[l] for l in range(nq))  # canonical on the way out
    ctx.close()


def test_scratch_placement_is_chosen_by_measurement_and_changes_no_residue(oracle):
    """Round 6: a context's first key switch draws LUMEN_KS_PLACEMENT candidates per scratch buffer and keeps the blocks
    under which rotations run fastest (lm_keyswitch.hip, get_scratch).  The choice must not show in the results: the same
    matrixInnerSumEval with the selection off, on, and on again after lumen_ctx_trim gives the oracle's residues; the
    chosen blocks are where lumen_ctx_scratch_info says, large enough, and stay put from one call to the next; the
    diagnostic probe runs on them."""
    P = make_params(oracle, 12, 8)            # beta = 4 digits x 10 limbs x 32 KB x 64 columns: ext = 84 MB, above the 64 MB threshold
    P.seed(21)
    sk = P.keygen_secret()
    ctx = make_context(P)
    n = 8
    gl = P.inner_sum_galois_elements(n)
    evks = [P.keygen_galois(sk, g) for g in gl]
    for g, e in zip(gl, evks):
        ctx.load_galois_key(g, e)
    cts = random_cts(P, 64, P.L, seed=22)
    pt = P.encode(np.arange(1, P.N + 1, dtype=np.uint64))
    want = P.matrix_inner_sum(cts, pt, n, evks)
    d = ctx.upload(cts)
    for k in (0, 6, 6):
        ctx.set_tuning("LUMEN_KS_PLACEMENT", k)
        ctx.trim()  # the buffers are drawn again at the next key switch
        assert np.array_equal(ctx.matrix_inner_sum(d, pt, n).download(), want), k
        blocks = {name: ctx.scratch_info(name) for name in ("ks_ext", "ks_u", "ks_coef", "ks_acc2", "ks_acc")}
        beta, LK = (P.L + 2 - 1) // 2, P.L + 2
        assert blocks["ks_ext"][0] and blocks["ks_ext"][1] >= 64 * beta * LK * P.N * 8, blocks
        assert blocks["ks_u"][0] and blocks["ks_u"][1] >= 64 * 2 * LK * P.N * 8, blocks
        assert np.array_equal(ctx.matrix_inner_sum(d, pt, n).download(), want), k
        assert blocks == {name: ctx.scratch_info(name) for name in blocks}, "the chosen blocks moved between two calls"
    assert ctx.ks_mac_probe(64, reps=3) > 0
    assert ctx.scratch_info("no_such_buffer") == (None, 0)
    ctx.close()


def test_tuning_switch_errors(small):
    """lumen_ctx_set_tuning: an unknown name and a value out of a switch's range are errors and change nothing (a typo in an
    A/B tool must not silently measure the default); the RCCL shared-device switch is a named test hook, not a name here."""
    from lumenos_amd.hip import LumenError
    _, ctx = small
    for name, bad in (("LUMEN_MODUP_TGROUP", 0), ("LUMEN_MODDOWN_TGROUP", 32), ("LUMEN_KS_BATCH", 0), ("LUMEN_KS_LANES", 3),
                      ("LUMEN_KS_PLACEMENT", 33)):
        with pytest.raises(LumenError, match="out of range"):
            ctx.set_tuning(name, bad)
    for name in ("LUMEN_NO_SUCH_SWITCH", "LUMEN_RCCL_SHARED_DEVICE", "LUMEN_CT_BLOCKS"):
        with pytest.raises(LumenError, match="unknown tuning switch"):
            ctx.set_tuning(name, 1)
    ctx.set_tuning("LUMEN_KS_LANES", 0)          # the documented ways back to the derived defaults
    ctx.set_tuning("LUMEN_KS_FUSED_DIGITS", -1)


def test_rescale_mixed_size_moduli_falls_back(oracle):
    """Moduli more than 16x apart have no coefficient-form tables: the per-step path answers."""
    from oracle.loader import Params
    log_n, two_n = 10, 2 << 10
    q = _ntt_primes_near(1 << 58, two_n, 2) + _ntt_primes_near(1 << 30, two_n, 2, down=False)
    P = Params.from_moduli(oracle, log_n, q, [], T_REF)
    ctx = make_context(P)
    cts = _adversarial_cts(P, 4, seed=8)
    got = ctx.rescale(ctx.upload(cts), 1).download()
    for c in range(3):
        ref = cts[c]
        while ref.shape[1] > 1:
            ref = P.rescale(ref)
        assert np.array_equal(got[c], ref), c
    ctx.close()


# ------------------------------------------------------------------ witness encryption
@pytest.mark.parametrize("log_n,num_q", [(10, 3), (12, 2), (14, 2)])
def test_encrypt_pk_matches_oracle_and_decrypts(oracle, log_n, num_q):
    """lumen_encrypt_pk == the oracle's deterministic encryption bit for bit (same ChaCha20-derived
    u, e0, e1), decrypts to the plaintexts, and does not depend on how the columns are batched."""
    P = make_params(oracle, log_n, num_q)
    P.seed(log_n)
    sk = P.keygen_secret()
    pk = P.keygen_public(sk)
    ctx = make_context(P)
    ctx.load_public_key(pk)
    seed = np.frombuffer(bytes(range(7, 39)), dtype=np.uint8)
    rng = np.random.default_rng(log_n)
    count, first = 5, 2**33 + 11
    vals = rng.integers(0, T_REF, size=(count, P.N), dtype=np.uint64)
    pts = np.stack([P.encode(v) for v in vals])
    got = ctx.encrypt_pk(pts, count, seed, first).download()
    for i in range(count):
        assert np.array_equal(got[i], P.encrypt_det(pk, pts[i], seed, first + i)), i
        assert np.array_equal(P.decrypt(sk, got[i], P.N), vals[i]), i
    # encryptions of zero (fhe/code.go:21-25 pads the matrix with one)
    z = ctx.encrypt_pk(None, 2, seed, 77).download()
    assert np.array_equal(z[1], P.encrypt_det(pk, None, seed, 78))
    assert not P.decrypt(sk, z[0], P.N).any()
    # a shard that starts in the middle produces the same ciphertexts
    tail = ctx.encrypt_pk(pts[3:], 2, seed, first + 3).download()
    assert np.array_equal(tail, got[3:])
    ctx.close()


def test_encrypted_witness_runs_through_the_prover(oracle):
    """Encrypt a witness on the GPU, run matrixInnerSumEval on it, decrypt: sum_i r_i * W[i][j]."""
    P = make_params(oracle, 10, 5)
    P.seed(4)
    sk = P.keygen_secret()
    pk = P.keygen_public(sk)
    ctx = make_context(P)
    ctx.load_public_key(pk)
    rows, cols = 256, 4
    W = oracle.witness(rows, cols, T_REF)
    pts = np.stack([P.encode(W[:, j]) for j in range(cols)])
    seed = np.zeros(32, dtype=np.uint8)
    m = ctx.encrypt_pk(pts, cols, seed, 0)
    gl = P.inner_sum_galois_elements(rows)
    for g in gl:
        ctx.load_galois_key(g, P.keygen_galois(sk, g))
    r = np.random.default_rng(9).integers(0, 2**63, size=rows, dtype=np.uint64)
    got = ctx.matrix_inner_sum(m, P.encode(r), rows).download()
    scale = P.rescale_scale(P.L, 2)
    for j in range(cols):
        want = int(np.sum(W[:, j].astype(object) * (r.astype(object) % T_REF)) % T_REF)
        assert int(P.decrypt(sk, got[j], 1, scale)[0]) == want, j
    ctx.close()


def test_golden_encrypt_det_gpu():
    """The committed fixture of the deterministic encryption, replayed through lumen_encrypt_pk."""
    g = np.load(os.path.join(GOLD, "encrypt_det.npz"))
    ctx = _ctx_from_golden(g)
    ctx.load_public_key(g["pk"])
    got = ctx.encrypt_pk(g["plaintexts"], 3, g["seed"], int(g["first"])).download()
    assert np.array_equal(got, g["ciphertexts"])
    ctx.close()


@pytest.mark.parametrize("log_n,num_q,rows", [(10, 3, 1024), (10, 2, 300), (12, 2, 4096), (14, 2, 16384)])
def test_encrypt_values_matches_oracle_encode_then_encrypt(oracle, log_n, num_q, rows):
    """lumen_encrypt_values (Encoder.Encode on the device, fused into the encryption's transforms) ==
    oracle Encode followed by the deterministic encryption, bit for bit; decrypts to the slot values."""
    from lumenos_amd import params as lp
    P = make_params(oracle, log_n, num_q)
    P.seed(100 + log_n)
    sk = P.keygen_secret()
    pk = P.keygen_public(sk)
    ctx = make_context(P)
    ctx.load_public_key(pk)
    ctx.encoder_set(lp.encoder_psi(T_REF, log_n))
    seed = np.frombuffer(bytes(range(50, 82)), dtype=np.uint8)
    rng = np.random.default_rng(rows)
    count, first = 3, 12345
    vals = rng.integers(0, 2**64, size=(count, rows), dtype=np.uint64)  # unreduced, as Prove's r (ligero.go:202)
    got = ctx.encrypt_values(vals, seed, first).download()
    for i in range(count):
        assert np.array_equal(got[i], P.encrypt_det(pk, P.encode(vals[i]), seed, first + i)), i
        assert np.array_equal(P.decrypt(sk, got[i], rows), vals[i] % np.uint64(T_REF)), i
    ctx.close()


# ------------------------------------------------------------------ client-side decryption
@pytest.mark.parametrize("log_n,num_q", [(10, 2), (10, 1), (12, 2), (14, 2)])
def test_decrypt_matches_oracle(oracle, log_n, num_q):
    """lumen_decrypt == the oracle's Decrypt + Decode on real encryptions (one and two limbs),
    including a non-trivial scale."""
    from lumenos_amd import params as lp
    P = make_params(oracle, log_n, num_q)
    P.seed(3 * log_n + num_q)
    sk = P.keygen_secret()
    pk = P.keygen_public(sk)
    ctx = make_context(P)
    ctx.encoder_set(lp.encoder_psi(T_REF, log_n))
    ctx.load_secret_key(sk)
    rng = np.random.default_rng(log_n)
    vals = rng.integers(0, T_REF, size=(3, P.N), dtype=np.uint64)
    cts = np.stack([P.encrypt(pk, P.encode(v)) for v in vals])
    s = ctx.upload(cts)
    got = ctx.decrypt(s, P.N)
    if num_q > 1:  # one 58-bit limb cannot hold T * noise for the 57-bit T: nothing decrypts there,
        assert np.array_equal(got, vals)  # the device must then still agree with the oracle's output
    for scale in (1, 12345678901234567):
        got = ctx.decrypt(s, 17, scale)
        for i in range(3):
            assert np.array_equal(got[i], P.decrypt(sk, cts[i], 17, scale)), (scale, i)
    ctx.close()


def test_gpu_only_round_trip_encrypt_prove_decrypt(oracle):
    """Witness -> lumen_encrypt_values -> matrixInnerSumEval -> lumen_decrypt, no oracle arithmetic in
    between: slot 0 of column j is sum_i r_i * W[i][j] mod T (the check Verify performs on MatR)."""
    from lumenos_amd import params as lp
    P = make_params(oracle, 10, 5)
    P.seed(21)
    sk = P.keygen_secret()
    pk = P.keygen_public(sk)
    ctx = make_context(P)
    ctx.encoder_set(lp.encoder_psi(T_REF, 10))
    ctx.load_public_key(pk)
    ctx.load_secret_key(sk)
    rows, cols = 512, 5
    W = oracle.witness(rows, cols, T_REF)
    m = ctx.encrypt_values(np.ascontiguousarray(W.T), np.arange(32, dtype=np.uint8), 0)
    for g in P.inner_sum_galois_elements(rows):
        ctx.load_galois_key(g, P.keygen_galois(sk, g))
    r = np.random.default_rng(2).integers(0, 2**63, size=rows, dtype=np.uint64)
    out = ctx.matrix_inner_sum(m, P.encode(r), rows)
    got = ctx.decrypt(out, 1, P.rescale_scale(P.L, 2))[:, 0]
    want = [int(np.sum(W[:, j].astype(object) * (r.astype(object) % T_REF)) % T_REF) for j in range(cols)]
    assert [int(x) for x in got] == want
    ctx.close()


# ------------------------------------------------------------------ error convention
@pytest.mark.parametrize("nl", [3, 5, 6])
def test_decrypt_at_any_level_matches_oracle(oracle, nl):
    """lumen_decrypt deeper than level 1 (TestEncode decrypts ciphertexts that were never rescaled): Garner's
    mixed-radix CRT on the device == the oracle's, on real encryptions and on uniformly random ciphertexts
    (phases on both sides of Q/2)."""
    from lumenos_amd import params as lp
    P = make_params(oracle, 10, 6)
    P.seed(30 + nl)
    sk = P.keygen_secret()
    pk = P.keygen_public(sk)
    ctx = make_context(P)
    ctx.load_secret_key(sk)
    ctx.encoder_set(lp.encoder_psi(T_REF, P.logN))
    rng = np.random.default_rng(nl)
    vals = rng.integers(0, T_REF, size=(2, P.N), dtype=np.uint64)
    cts = np.stack([P.encrypt(pk, P.encode(vals[i], nl), nl) for i in range(2)] + list(random_cts(P, 2, nl, seed=5)))
    got = ctx.decrypt(ctx.upload(cts), P.N, 7)
    assert np.array_equal(got, P.decrypt_batch(sk, cts, P.N, 7))
    assert np.array_equal(ctx.decrypt(ctx.upload(cts[:2]), P.N), vals)
    ctx.close()


def test_error_paths_report_status_and_message(oracle, small):
    """Every entry point returns non-zero and leaves a message (the cgo convention of the reference,
    vdec/prover.go:121-232): misuse must never reach a kernel with operands it does not expect."""
    from lumenos_amd.hip import LumenError
    from lumenos_amd import params as lp
    P, ctx = small
    s4 = ctx.new_set(2, 4).fill_random(1)
    s2 = ctx.new_set(2, 2).fill_random(2)
    seed = np.zeros(32, dtype=np.uint8)

    def fails(fn, text):
        with pytest.raises(LumenError) as e:
            fn()
        assert text in str(e.value), str(e.value)

    fails(lambda: ctx.new_set(1, 99), "num_limbs")
    fails(lambda: ctx.rescale(s2, 3), "target_limbs")
    fails(lambda: ctx.gather(s2, np.array([5], dtype=np.uint32)), "out of range")
    fails(lambda: ctx.inner_sum(s2, 8), "top level")
    fails(lambda: ctx.inner_sum(s4, 6), "power of two")
    pt = np.zeros((4, P.N), dtype=np.uint64)
    fails(lambda: ctx.matrix_inner_sum(s4, pt, 1 << 30), "power of two")
    fails(lambda: ctx.load_galois_key(4, np.zeros(P.evk_shape(), dtype=np.uint64)), "odd residue")
    # the encryption / decryption entry points without their keys or tables
    from helpers import make_context
    fresh = make_context(P)
    fails(lambda: fresh.encrypt_pk(None, 1, seed, 0), "no public key")
    fresh.load_public_key(np.zeros((2, P.L + P.K, P.N), dtype=np.uint64))
    fails(lambda: fresh.encrypt_values(np.zeros((1, 8), dtype=np.uint64), seed, 0), "no encoder tables")
    fails(lambda: fresh.encoder_set(12345), "primitive 2N-th root")
    fresh.encoder_set(lp.encoder_psi(T_REF, P.logN))
    fails(lambda: fresh.decrypt(fresh.new_set(1, 2), 1), "no secret key")
    fresh.load_secret_key(np.zeros((P.L, P.N), dtype=np.uint64))
    fails(lambda: fresh.decrypt(fresh.upload_lanes(np.zeros((1, 2, 2, P.N // 2), dtype=np.uint64), 1), 1), "lane-sharded")
    bad = np.full((2, P.L + P.K, P.N), 2**63, dtype=np.uint64)
    fails(lambda: fresh.load_public_key(bad), "out of range")
    # one asynchronous leaf job at a time
    fresh.leaf_digests_begin(fresh.new_set(3, 2).fill_random(4))
    fails(lambda: fresh.leaf_digests_begin(fresh.new_set(3, 2).fill_random(5)), "already in flight")
    assert fresh.leaf_digests_end().shape == (3, 32)
    # round-3 entry points: the asynchronous serialiser wants page-locked memory and a large enough buffer,
    # tuning switches are a closed list, the ring switch needs its key and a supported target degree
    import ctypes as C
    from lumenos_amd.hip import pinned_bytes
    u8p = C.POINTER(C.c_uint8)
    each = fresh.ct_serialized_size(2)
    fails(lambda: fresh.ct_serialize_into(s2, np.zeros(2 * each, dtype=np.uint8), wait=False), "page-locked")
    buf = pinned_bytes(3 * each)
    fails(lambda: fresh._ck(fresh.lib.lumen_ct_serialize(fresh.h, s2.h, 0, 2, buf.ctypes.data_as(u8p), each)), "too small")
    fails(lambda: fresh._ck(fresh.lib.lumen_ct_serialize(fresh.h, s2.h, 1, 2, buf.ctypes.data_as(u8p), 3 * each)), "exceeds set")
    fails(lambda: fresh.set_tuning("LUMEN_NO_SUCH_SWITCH", 1), "unknown tuning switch")
    out = np.zeros(2 * 2 * 256, dtype=np.uint64)
    fails(lambda: fresh._ck(fresh.lib.lumen_ring_switch(fresh.h, s2.h, out.ctypes.data_as(C.POINTER(C.c_uint64)))),
          "no ring-switch key")
    key = np.zeros(fresh.ringswitch_key_shape(), dtype=np.uint64)
    fails(lambda: fresh.load_ringswitch_key(9, key), "not supported")
    fails(lambda: fresh.load_ringswitch_key(P.logN + 1, key), "not supported")
    key[0, 0, 0, 0, 0] = 2**63
    fails(lambda: fresh.load_ringswitch_key(8, key), "out of range")
    # round 4: the key's length crosses the ABI -- an older (1+K)-limb layout or a truncated block is refused
    # before a word of it is read
    u64p = C.POINTER(C.c_uint64)
    short = np.zeros(key.size // 2 + 3, dtype=np.uint64)
    fails(lambda: fresh._ck(fresh.lib.lumen_load_ringswitch_key(fresh.h, 8, 13, short.ctypes.data_as(u64p), short.size)),
          "ring-switch key of")
    # ... lumen_ctx_trim hands pooled storage and scratch back but not under a leaf job, and the context works on
    gone = fresh.new_set(4, 2).fill_random(9)
    want = fresh.leaf_digests(gone)
    fresh.leaf_digests_begin(gone)
    fails(fresh.trim, "in flight")
    assert np.array_equal(fresh.leaf_digests_end(), want)
    gone.free()
    fresh.trim()
    again = fresh.new_set(4, 2).fill_random(9)
    assert np.array_equal(fresh.leaf_digests(again), want)
    fresh.wait_for(fresh)  # waiting for oneself is a no-op, not a deadlock
    fresh.close()


def test_full_size_encrypt_rescale_decrypt_round_trip(full_d):
    """At the headline parameters (N = 2^14, 12 + 2 limbs): witness columns -> lumen_encrypt_values ->
    Rescale to level 1 -> lumen_decrypt returns the columns (a size-independent property: no oracle
    arithmetic on the path), and one ciphertext is spot-checked against the oracle's encryption."""
    from lumenos_amd import params as lp
    P, ctx = full_d
    sk = P.keygen_secret()
    pk = P.keygen_public(sk)
    ctx.load_public_key(pk)
    ctx.load_secret_key(sk)
    ctx.encoder_set(lp.encoder_psi(T_REF, 14))
    rng = np.random.default_rng(140)
    vals = rng.integers(0, T_REF, size=(6, P.N), dtype=np.uint64)
    seed = np.frombuffer(bytes(range(1, 33)), dtype=np.uint8)
    cts = ctx.encrypt_values(vals, seed, 1000)
    assert np.array_equal(cts.download(2, 1)[0], P.encrypt_det(pk, P.encode(vals[2]), seed, 1002))
    lvl1 = ctx.rescale(cts, 2)
    got = ctx.decrypt(lvl1, P.N, P.rescale_scale(P.L, 2))
    assert np.array_equal(got, vals)


@pytest.mark.parametrize("cols,rho", [(2, 1), (1, 2), (2, 2), (8, 4), (16, 1), (4, 8), (64, 4)])
def test_encode_other_rates_and_smallest_shapes(oracle, small, cols, rho):
    """fhe.Encode for rhoInv other than the reference's constant 2 (cmd/server/main.go:23) and for the
    smallest matrices the control flow of nttInner distinguishes (sizes 2, 4, 8: the hard-coded base cases,
    ntt.go:24-244; 16, 32: the first six-step splits; rho = 1: no padding column at all)."""
    P, ctx = small
    S = cols * rho
    m = random_cts(P, cols, 2, seed=1000 + 17 * cols + rho)
    zero = random_cts(P, 1, 2, seed=2000 + cols)[0]
    # the base cases of 4 and 8 values read RootForward(4) / (8) whatever the table's size: with the field the
    # reference would build for such a matrix (core.NewPrimeField(T, cols * rhoInv) of 4 or 8 roots) it panics
    # with an index out of range; the library refuses the call, the oracle's wrapper raises
    if S in (4, 8):
        from lumenos_amd.hip import LumenError
        tight = oracle.field_roots(T_REF, S)
        ctx.field_set(tight)
        with pytest.raises(LumenError, match="the reference panics"):
            ctx.encode(ctx.upload(m), zero, rho)
        with pytest.raises(IndexError, match="the reference panics"):
            P.ct_encode(m, rho, zero, tight)
    roots = oracle.field_roots(T_REF, max(S, 16))
    ctx.field_set(roots)
    got = ctx.encode(ctx.upload(m), zero, rho)
    assert got.count == S
    assert np.array_equal(got.download(), P.ct_encode(m, rho, zero, roots)), (cols, rho)


def test_empty_sets_through_the_batch_entry_points(oracle, small):
    """[]*rlwe.Ciphertext of length zero is legal Go everywhere on the path (a rank that owns no queried
    column gathers nothing; a proof slice may be empty): every batch entry point accepts a set of zero
    ciphertexts, launches nothing and returns an empty result."""
    from lumenos_amd.hip import pinned_bytes
    P, ctx = small
    e4, e2 = ctx.new_set(0, P.L), ctx.new_set(0, 2)
    assert e4.count == 0 and e4.download().shape == (0, 2, P.L, P.N)
    assert ctx.rescale(e4, 2).count == 0
    assert ctx.gather(ctx.upload(random_cts(P, 3, 2, seed=1)), np.zeros(0, dtype=np.uint32)).count == 0
    assert ctx.ct_serialize(e2) == b""
    assert ctx.ct_serialize_into(e2, pinned_bytes(8), wait=False) == 0
    assert ctx.ct_deserialize(b"", 0, 2).count == 0
    assert ctx.leaf_digests(e2).shape == (0, 32)
    pt = np.zeros((P.L, P.N), dtype=np.uint64)
    sk = P.keygen_secret()
    for g in P.inner_sum_galois_elements(8):
        ctx.load_galois_key(g, P.keygen_galois(sk, g))
    assert ctx.mul_plain(e4, pt).count == 0
    assert ctx.inner_sum(e4, 8).count == 0
    assert ctx.matrix_inner_sum(e4, pt, 8).count == 0
    key = P.keygen_ringswitch(sk, P.keygen_secret_small(8), 8)
    ctx.load_ringswitch_key(8, key)
    assert ctx.ring_switch(e2).shape == (0, 2, 256)
    ctx.set_ntt(e4, False)
    ctx.sync()


@pytest.mark.parametrize("num_q,num_p", [(22, 2), (9, 1), (15, 2)])
def test_maximum_limb_count_and_wide_gadget(oracle, num_q, num_p):
    """The ABI's limits: L + K = LUMEN_MAX_LIMBS = 24 limbs, and gadget products of more than 7 digits (beta = 11,
    9, 8: the accumulated sum then takes the Barrett step of the reduction instead of conditional subtractions --
    the reference's own chains stop at beta = 6).  Limb transforms, rescale from the top to level 1 and to a middle
    level, ct x pt, InnerSum, matrixInnerSumEval, witness encryption and decryption at depth, bit for bit."""
    P = make_params(oracle, 10, num_q, num_p=num_p)
    assert P.L + P.K <= 24 and P.beta() == -(-num_q // num_p)
    P.seed(2200 + num_q)
    ctx = make_context(P)
    sk = P.keygen_secret()
    pk = P.keygen_public(sk)
    cts = random_cts(P, 3, P.L, seed=num_q)
    s = ctx.upload(cts)
    ctx.set_ntt(s, True)
    got = s.download()
    for l in (0, P.L // 2, P.L - 1):
        assert np.array_equal(got[1, 1, l], P.limb_intt(cts[1, 1, l], l)), l
    ctx.set_ntt(s, False)
    assert np.array_equal(s.download(), cts)
    for target in (2, P.L // 2):
        ref = cts[2]
        while ref.shape[1] > target:
            ref = P.rescale(ref)
        assert np.array_equal(ctx.rescale(ctx.upload(cts[2:]), target).download()[0], ref), target
    n = 8
    gl = P.inner_sum_galois_elements(n)
    evks = [P.keygen_galois(sk, g) for g in gl]
    for g, e in zip(gl, evks):
        ctx.load_galois_key(g, e)
    inner = ctx.inner_sum(ctx.upload(cts), n).download()
    for c in range(3):
        assert np.array_equal(inner[c], P.inner_sum(cts[c], n, evks)), c
    rng = np.random.default_rng(num_q)
    col = rng.integers(0, T_REF, size=n, dtype=np.uint64)
    r = rng.integers(0, 2**63, size=n, dtype=np.uint64)
    enc = P.encrypt(pk, P.encode(col))[None]
    pt = P.encode(r)
    out = ctx.matrix_inner_sum(ctx.upload(enc), pt, n).download()
    assert np.array_equal(out, P.matrix_inner_sum(enc, pt, n, evks))
    want = int(np.sum(col.astype(object) * (r.astype(object) % T_REF)) % T_REF)
    assert int(P.decrypt(sk, out[0], 1, P.rescale_scale(P.L, 2))[0]) == want
    # the input side and the client side at this depth
    from lumenos_amd import params as lp
    ctx.load_public_key(pk)
    ctx.load_secret_key(sk)
    ctx.encoder_set(lp.encoder_psi(T_REF, P.logN))
    seed = np.arange(32, dtype=np.uint8)
    vals = rng.integers(0, T_REF, size=(2, P.N), dtype=np.uint64)
    e = ctx.encrypt_values(vals, seed, 3)
    assert np.array_equal(e.download()[1], P.encrypt_det(pk, P.encode(vals[1]), seed, 4))
    assert np.array_equal(ctx.decrypt(e, P.N), vals)  # all L limbs: mixed-radix CRT
    ctx.close()


def test_ragged_batches(oracle, keyed):
    """Counts that are no multiple of any internal batch: 67 columns through the key switch (batches of 64 + 3),
    300 ciphertexts through the ring switch (256 + 44), 131 through the rescale and the leaf digests."""
    P, ctx, sk = keyed
    n = 4
    gl = P.inner_sum_galois_elements(n)
    evks = [P.keygen_galois(sk, g) for g in gl]
    for g, e in zip(gl, evks):
        ctx.load_galois_key(g, e)
    cts = random_cts(P, 67, P.L, seed=6700)
    pt = P.encode(np.arange(n, dtype=np.uint64) + 5)
    got = ctx.matrix_inner_sum(ctx.upload(cts), pt, n).download()
    want = P.matrix_inner_sum(cts, pt, n, evks)
    assert np.array_equal(got, want)
    l1 = random_cts(P, 300, 2, seed=300)
    sk_small = P.keygen_secret_small(8)
    key = P.keygen_ringswitch(sk, sk_small, 8)
    ctx.load_ringswitch_key(8, key)
    rs = ctx.ring_switch(ctx.upload(l1))
    for c in (0, 255, 256, 299):
        assert np.array_equal(rs[c], P.ring_switch(l1[c], key, 8)), c
    big = random_cts(P, 131, P.L, seed=131)
    lvl1 = ctx.rescale(ctx.upload(big), 2)
    ref_l1, ref_dig = P.commit_leaves(big)
    assert np.array_equal(lvl1.download(), ref_l1)
    assert np.array_equal(ctx.leaf_digests(lvl1), ref_dig)
