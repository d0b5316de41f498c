"""CPU: the decrypted-value equalities the reference's own tests assert, re-run on the oracle
(fhe/code_test.go:87-116 TestEncode, fhe/ligero_test.go:128-174 TestLigeroE2E), plus golden
regression vectors."""
import os

import numpy as np
import pytest

from helpers import T_REF, make_params

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def bgv(oracle):
    P = make_params(oracle, 10, 6)  # LogQ = [58, 56 x 5], LogP = [55, 55]
    P.seed(3)
    sk = P.keygen_secret()
    pk = P.keygen_public(sk)
    return P, sk, pk


def test_encrypt_decrypt_rescale(bgv):
    P, sk, pk = bgv
    vals = np.random.default_rng(1).integers(0, T_REF, size=P.N, dtype=np.uint64)
    ct = P.encrypt(pk, P.encode(vals))
    assert np.array_equal(P.decrypt(sk, ct, P.N), vals)
    assert np.array_equal(P.decrypt(sk, P.rescale(ct), P.N, P.rescale_scale(P.L, P.L - 1)), vals)
    assert np.array_equal(P.decrypt(sk, P.rescale_to_level1(ct), P.N, P.rescale_scale(P.L, 2)), vals)


def test_mul_plain_is_slotwise_product(bgv):
    P, sk, pk = bgv
    rng = np.random.default_rng(2)
    a = rng.integers(0, T_REF, size=P.N, dtype=np.uint64)
    b = rng.integers(0, T_REF, size=P.N, dtype=np.uint64)
    got = P.decrypt(sk, P.mul_plain(P.encrypt(pk, P.encode(a)), P.encode(b)), P.N)
    assert np.array_equal(got, np.array([int(x) * int(y) % T_REF for x, y in zip(a, b)], dtype=np.uint64))


def test_rotation_and_row_swap(bgv):
    P, sk, pk = bgv
    vals = np.random.default_rng(3).integers(0, T_REF, size=P.N, dtype=np.uint64)
    ct = P.encrypt(pk, P.encode(vals))
    half = P.N // 2
    g = P.galois_element(1)
    d = P.decrypt(sk, P.automorphism(ct, g, P.keygen_galois(sk, g)), P.N)
    assert np.array_equal(d[:half], np.roll(vals[:half], -1)) and np.array_equal(d[half:], np.roll(vals[half:], -1))
    g = 2 * P.N - 1
    d = P.decrypt(sk, P.automorphism(ct, g, P.keygen_galois(sk, g)), P.N)
    assert np.array_equal(d[:half], vals[half:]) and np.array_equal(d[half:], vals[:half])


@pytest.mark.parametrize("n", [64, 512, 1024])
def test_inner_sum_contract(bgv, n):
    """InnerSum(ct, 1, n): slot 0 = sum of the first n slots, including n == N."""
    P, sk, pk = bgv
    vals = np.random.default_rng(n).integers(0, T_REF, size=P.N, dtype=np.uint64)
    ct = P.encrypt(pk, P.encode(vals))
    gl = P.inner_sum_galois_elements(n)
    assert len(gl) == n.bit_length() - 1
    out = P.inner_sum(ct, n, [P.keygen_galois(sk, g) for g in gl])
    assert int(P.decrypt(sk, out, 1)[0]) == int(np.sum(vals[:n].astype(object)) % T_REF)


def test_fhe_encode_decrypts_to_plain_encode(oracle, bgv):
    """TestEncode (fhe/code_test.go:14-123): Dec(fhe.Encode(Enc(M))) == core.Encode(M) row by row."""
    P, sk, pk = bgv
    rows, cols, rho = P.N // 2, 16, 2
    W = oracle.witness(rows, cols, T_REF)
    roots = oracle.field_roots(T_REF, cols * rho)
    cts = np.stack([P.encrypt(pk, P.encode(W[:, j])) for j in range(cols)])
    zero = P.encrypt(pk, P.encode(np.zeros(rows, dtype=np.uint64)))
    enc = P.ct_encode(cts, rho, zero, roots)
    dec = np.stack([P.decrypt(sk, enc[j], rows) for j in range(cols * rho)], axis=1)
    ref = np.stack([oracle.plain_encode(W[i], rho, T_REF, roots) for i in range(rows)])
    assert np.array_equal(dec, ref)


def test_ligero_e2e_on_oracle(oracle, bgv):
    """TestLigeroE2E (fhe/ligero_test.go:70-176) on a small shape: Commit + Prove on ciphertexts,
    decrypt, then the checks of Proof.Verify (ligero.go:517-574) and the MatR/MatZ comparison
    against the plain prover (ligero.go:799-953)."""
    from oracle.loader import Transcript
    P, sk, pk = bgv
    rows, cols, rho, queries = P.N // 2, 16, 2, 12
    S = cols * rho
    W = oracle.witness(rows, cols, T_REF)
    roots = oracle.field_roots(T_REF, S)
    cts = np.stack([P.encrypt(pk, P.encode(W[:, j])) for j in range(cols)])
    zero = P.encrypt(pk, P.encode(np.zeros(rows, dtype=np.uint64)))
    # Commit
    enc = P.ct_encode(cts, rho, zero, roots)
    lvl1, dig = P.commit_leaves(enc)
    nodes, root = oracle.merkle(dig)
    # Prove
    tr = Transcript(oracle, "test")
    r = np.array([tr.sample_u64("r") for _ in range(rows)], dtype=np.uint64)  # raw u64 (ligero.go:202-203)
    z = 1
    b = np.array([pow(pow(z, cols, T_REF), i, T_REF) for i in range(rows)], dtype=np.uint64)
    gl = P.inner_sum_galois_elements(rows)
    evks = [P.keygen_galois(sk, g) for g in gl]
    mat_r = P.matrix_inner_sum(cts, P.encode(r), rows, evks)
    mat_z = P.matrix_inner_sum(cts, P.encode(b), rows, evks)
    tr.append("point", int(z).to_bytes(8, "little"))
    qidx = [tr.sample_u64("query") % S for _ in range(queries)]
    # client: decrypt
    sc = P.rescale_scale(P.L, 2)
    R = np.array([int(P.decrypt(sk, mat_r[j], 1, sc)[0]) for j in range(cols)], dtype=np.uint64)
    Z = np.array([int(P.decrypt(sk, mat_z[j], 1, sc)[0]) for j in range(cols)], dtype=np.uint64)
    # plain prover equality (ligero_test.go:164-174)
    rT = r.astype(object) % T_REF
    Wo = W.astype(object)
    assert [int(x) for x in R] == [int(np.sum(Wo[:, j] * rT) % T_REF) for j in range(cols)]
    assert [int(x) for x in Z] == [int(np.sum(Wo[:, j] * b.astype(object)) % T_REF) for j in range(cols)]
    # Verify (ligero.go:554-571)
    encR = oracle.plain_encode(R, rho, T_REF, roots)
    encZ = oracle.plain_encode(Z, rho, T_REF, roots)
    for qi in qidx:
        col = P.decrypt(sk, lvl1[qi], rows, sc).astype(object)
        assert oracle.merkle_verify(dig[qi].tobytes(), oracle.merkle_path(nodes, S, qi), root, qi)
        assert int(np.sum(col * rT) % T_REF) == int(encR[qi])
        assert int(np.sum(col * b.astype(object)) % T_REF) == int(encZ[qi])
    a = [pow(z, i, T_REF) for i in range(cols)]
    value = int(np.sum(Wo.reshape(-1))) % T_REF  # P(1): Horner at z = 1 (core/poly.go:21-30)
    assert sum(int(Z[j]) * a[j] for j in range(cols)) % T_REF == value


@pytest.mark.parametrize("S", [16, 32, 64])
def test_golden_encode(oracle, S):
    from oracle.loader import Params
    g = np.load(os.path.join(GOLD, f"encode_S{S}.npz"))
    P = Params.from_moduli(oracle, int(g["log_n"]), [int(x) for x in g["q"]], [int(x) for x in g["p"]], int(g["T"]))
    assert P.psi == [int(x) for x in g["psi"]]
    assert np.array_equal(P.ct_encode(g["matrix"], 2, g["zero"], g["roots"]), g["encoded"])


def test_golden_evaluator(oracle):
    from oracle.loader import Params
    g = np.load(os.path.join(GOLD, "evaluator.npz"))
    P = Params.from_moduli(oracle, int(g["log_n"]), [int(x) for x in g["q"]], [int(x) for x in g["p"]], int(g["T"]))
    lvl1, dig = P.commit_leaves(g["cts"])
    assert np.array_equal(lvl1, g["level1"]) and np.array_equal(dig, g["digests"])
    assert [int(x) for x in g["gal_els"]] == P.inner_sum_galois_elements(int(g["rows"]))
    got = P.matrix_inner_sum(g["cts"], g["pt"], int(g["rows"]), list(g["evks"]))
    assert np.array_equal(got, g["matrix_inner_sum"])


@pytest.mark.parametrize("num_p,logn_small", [(2, 10), (2, 8), (1, 10), (1, 8), (0, 10), (0, 7)])
def test_ring_switch_on_oracle(oracle, num_p, logn_small):
    """RingSwitchNew (fhe/ring_switch.go:106-113) on the oracle's BGV, on all three gadget paths Lattigo
    takes by the key's LevelP (two special primes: RNS digit only, the reference's configurations; one:
    base-2^13 digits + ModDown; none: base-2^13 digits, TestRingSwitch's parameters): the same-degree switch
    decrypts to the same slots; a smaller degree keeps the coefficients of X^(i*N/n)
    (SwitchCiphertextRingDegreeNTT)."""
    T = 0x3EE0001  # ring_switch_test.go:17
    P = make_params(oracle, 10, 3, num_p=num_p, T=T)
    P.seed(77)
    sk = P.keygen_secret()
    pk = P.keygen_public(sk)
    vals = np.zeros(P.N, dtype=np.uint64)
    vals[:2] = 1  # m := []uint64{1, 1}
    ct = P.rescale_to_level1(P.encrypt(pk, P.encode(vals)))
    sk_small = P.keygen_secret_small(logn_small)
    rns, pw2 = P.rs_key_shape()
    assert (rns, pw2) == ({2: 2, 1: 3, 0: 3}[num_p], 1 if num_p == 2 else 5)
    key = P.keygen_ringswitch(sk, sk_small, logn_small)
    assert key.shape == (rns, pw2, 2, P.L + P.K, P.N)
    small = P.ring_switch(ct, key, logn_small)
    m = P.decrypt_small_coeffs(sk_small, logn_small, small)
    assert np.array_equal(m, P.decrypt_big_coeffs_l0(sk, ct)[::P.N >> logn_small])
    if logn_small == 10:
        assert np.array_equal(P.decode_coeffs(m, P.rescale_scale(P.L, 2), 2), vals[:2])
    # level 0 reads RNS digit 0 only: the other digits of the key do not matter
    key2 = key.copy()
    key2[1:] = 0
    assert np.array_equal(P.ring_switch(ct, key2, logn_small), small)


def test_ring_switch_reference_test_parameters(oracle):
    """TestRingSwitch itself (fhe/ring_switch_test.go:13-77): LogN = 12 -> 12, LogQ = [58], no special
    prime, T = 0x3ee0001, m = [1, 1] encrypted under pk at level 0 (no rescale), RingSwitch, decrypt under
    skNew, decode: mCheck == m.  With LevelP = -1 the key has ceil(58/13) = 5 power-of-two entries for its
    one RNS digit and the switch has no ModDown."""
    T = 0x3EE0001
    P = make_params(oracle, 12, 1, num_p=0, T=T)
    P.seed(13)
    sk = P.keygen_secret()
    pk = P.keygen_public(sk)
    m = np.array([1, 1], dtype=np.uint64)
    ct = P.encrypt(pk, P.encode(m))
    assert ct.shape == (2, 1, P.N)
    sk_new = P.keygen_secret_small(12)
    assert P.rs_key_shape() == (1, 5)
    key = P.keygen_ringswitch(sk, sk_new, 12)
    ct2 = P.ring_switch(ct, key, 12)
    coeffs = P.decrypt_small_coeffs(sk_new, 12, ct2)
    assert np.array_equal(P.decode_coeffs(coeffs, 1, 2), m)
    # and the whole plaintext polynomial survives, not just two slots
    assert np.array_equal(coeffs, P.decrypt_big_coeffs_l0(sk, ct))


# ------------------------------------------------------------------ deterministic pk encryption (lo_encdet.c)
def test_det_encrypt_decrypts_and_noise_is_small(oracle):
    """The checker of lumen_encrypt_pk: Dec(Enc_det(pt)) = pt at the top level, encryptions of zero
    decrypt to zero, and the phase noise stays near |u*e_pk + e0 + e1*s| <~ 2*19*N (far below q/2T)."""
    P = make_params(oracle, 10, 3)
    P.seed(2)
    sk = P.keygen_secret()
    pk = P.keygen_public(sk)
    seed = np.arange(32, dtype=np.uint8)
    vals = np.random.default_rng(1).integers(0, T_REF, size=P.N, dtype=np.uint64)
    for idx in (0, 1, 2**40 + 5):
        ct = P.encrypt_det(pk, P.encode(vals), seed, idx)
        assert np.array_equal(P.decrypt(sk, ct, P.N), vals)
    z = P.encrypt_det(pk, None, seed, 9)
    assert not P.decrypt(sk, z, P.N).any()
    # different indices / seeds give different ciphertexts; the same pair repeats exactly
    a, b = P.encrypt_det(pk, None, seed, 3), P.encrypt_det(pk, None, seed, 4)
    assert not np.array_equal(a, b) and np.array_equal(a, P.encrypt_det(pk, None, seed, 3))
    seed2 = seed.copy()
    seed2[31] ^= 1
    assert not np.array_equal(a, P.encrypt_det(pk, None, seed2, 3))


def test_det_sampler_distribution(oracle):
    """Ternary u: each of -1, 0, 1 with probability 1/3; e: sigma 3.2, |e| <= 19, mean 0
    ([LATTIGO-RECALL] DefaultXs / DefaultXe)."""
    P = make_params(oracle, 12, 1, num_p=0)
    seed = np.frombuffer(bytes(range(100, 132)), dtype=np.uint8)
    u = np.concatenate([P.det_small(seed, i, 0) for i in range(16)]).astype(np.int64)
    n = u.size
    for v in (-1, 0, 1):
        assert abs((u == v).sum() / n - 1 / 3) < 4 * np.sqrt(2 / 9 / n)
    e = np.concatenate([P.det_small(seed, i, s) for i in range(16) for s in (1, 2)]).astype(np.int64)
    assert np.abs(e).max() <= 19
    assert abs(e.mean()) < 4 * 3.2 / np.sqrt(e.size)
    assert abs(e.var() - 10.24) < 0.25
    assert abs((e == 0).mean() - 0.12467) < 0.005
    # streams are independent of each other
    assert not np.array_equal(P.det_small(seed, 0, 1), P.det_small(seed, 0, 2))


def test_det_sampler_chacha_layout(oracle):
    """Coefficient k of stream 0 is word k of ChaCha20(seed, nonce = LE64(index) || LE32(0)):
    checked against the RFC 8439 keystream the oracle's ChaCha20 produces."""
    import ctypes as C
    P = make_params(oracle, 10, 1, num_p=0)
    seed = np.arange(32, dtype=np.uint8)
    idx = 0x0102030405060708
    nonce = np.frombuffer(idx.to_bytes(8, "little") + (0).to_bytes(4, "little"), dtype=np.uint8).copy()
    ks = np.zeros(4 * P.N, dtype=np.uint8)
    oracle.lib.lo_chacha20_xor(seed.ctypes.data_as(C.POINTER(C.c_uint8)), nonce.ctypes.data_as(C.POINTER(C.c_uint8)),
                               0, ks.ctypes.data_as(C.POINTER(C.c_uint8)), ks.size)
    w = ks.view("<u4").astype(np.uint64)
    assert np.array_equal(P.det_small(seed, idx, 0).astype(np.int64), ((w * 3) >> 32).astype(np.int64) - 1)


def test_golden_encrypt_det(oracle):
    """tests/golden/encrypt_det.npz: the sampler's small polynomials and the ciphertexts, replayed."""
    from oracle.loader import Params
    g = np.load(os.path.join(GOLD, "encrypt_det.npz"))
    P = Params.from_moduli(oracle, int(g["log_n"]), [int(x) for x in g["q"]], [int(x) for x in g["p"]], int(g["T"]))
    first = int(g["first"])
    for i in range(3):
        for s in range(3):
            assert np.array_equal(P.det_small(g["seed"], first + i, s), g["small"][i, s]), (i, s)
        ct = P.encrypt_det(g["pk"], g["plaintexts"][i], g["seed"], first + i)
        assert np.array_equal(ct, g["ciphertexts"][i]), i
        assert np.array_equal(P.decrypt(g["sk"], ct, P.N), g["values"][i]), i
