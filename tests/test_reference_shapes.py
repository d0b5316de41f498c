"""GPU: the reference's own test shapes and every BASELINE.json configuration AT THEIR OWN PARAMETERS
(the Q chain is the one fhe.GenerateBGVParamsForNTT derives, fhe/bfv.go:121-188 -- no extra limb).

* TestEncode (fhe/code_test.go:14-123): 2048 x 1024, LogN = 13, L = 10; every one of the 2048 encoded
  columns is decrypted at the top level and compared with core.Encode of the plain rows.
* Commit at TestLigeroE2E's shape (fhe/ligero_test.go:24, 2048 x 1024, LogN = 12, L = 10): every leaf
  ciphertext (level 1) decrypts to the plain encoding.
* BASELINE configs A (1024, 12), B (2048, 12), C (4096, 13), D (4096, 14): matrixInnerSumEval with
  rows = N on real keys, rescale + leaf digests, witness encryption / decryption and the ring switch to
  LogN = 10, bit for bit against the oracle.

Fresh noise matters here: fhe.Encode never rescales, and the 1024-column shapes only fit the
heuristic's chain because rlwe.Encryptor divides its encryptions of zero by P
(tools/noise_budget.py).
"""
import numpy as np
import pytest

from helpers import T_REF, make_context, random_cts

pytestmark = pytest.mark.gpu

CONFIGS = {  # name -> (cols, LogN, expected Q-chain length: results/baseline/server/bench_*.txt:16)
    "A_2048x1024": (1024, 12, 10),
    "B_4096x2048": (2048, 12, 11),
    "C_8192x4096": (4096, 13, 12),
    "D_16384x4096": (4096, 14, 12),
}


def reference_params(oracle, cols, log_n):
    from lumenos_amd import params as lp
    from oracle.loader import Params
    B = lp.generate_bgv_params_for_ntt(cols, log_n)
    P = Params.from_moduli(oracle, log_n, B.q, B.p, B.T)
    assert P.psi == B.psi
    return P


def psi_T(log_n):
    from lumenos_amd import params as lp
    return pow(lp.primitive_root(T_REF), (T_REF - 1) // (2 << log_n), T_REF)


def encrypt_witness(oracle, P, ctx, rows, cols, seed_byte):
    """RandomMatrixRowMajor (core/utils.go:46-82) + Encoder.Encode + EncryptNew of every column, on the device."""
    sk = P.keygen_secret()
    pk = P.keygen_public(sk)
    ctx.load_public_key(pk)
    ctx.encoder_set(psi_T(P.logN))
    matrix = oracle.witness(rows, cols, T_REF)  # [rows][cols]
    columns = np.ascontiguousarray(matrix.T)    # column j = batched slots of ciphertext j
    seed = np.full(32, seed_byte, dtype=np.uint8)
    cts = ctx.encrypt_values(columns, seed, 0)
    zero = ctx.encrypt_pk(None, 1, seed, cols).download()[0]  # the one Enc(0) of fhe/code.go:15-22
    return sk, matrix, cts, zero


def plain_encode_rows(oracle, matrix, rho, roots):
    """core.Encode of every row (core/code.go:3-23): [rows][cols*rho]."""
    return np.stack([oracle.plain_encode(matrix[i], rho, T_REF, roots) for i in range(matrix.shape[0])])


def test_encode_reference_shape_every_column_decrypts(oracle):
    """TestEncode at its real shape: 2048 x 1024, LogN = 13, the heuristic's L = 10, all 2048 columns."""
    rows, cols, rho, log_n = 2048, 1024, 2, 13
    P = reference_params(oracle, cols, log_n)
    assert (P.L, P.K) == (10, 2)
    P.seed(1313)
    ctx = make_context(P)
    sk, matrix, cts, zero = encrypt_witness(oracle, P, ctx, rows, cols, 0x13)
    S = cols * rho
    roots = oracle.field_roots(T_REF, S)
    ctx.field_set(roots)
    enc = ctx.encode(cts, zero, rho)
    assert enc.count == S and enc.nl == P.L  # Encode does not rescale
    assert ctx.mul_counter() == 9217         # ct x scalar multiplications of S = 2048
    want = plain_encode_rows(oracle, matrix, rho, roots)  # [rows][S]
    ctx.load_secret_key(sk)
    bad = []
    for first in range(0, S, 256):
        got = P.decrypt_batch(sk, enc.download(first, 256), rows)  # Decryptor + Encoder.Decode at level 9
        dev = ctx.decrypt(enc.slice(first, 256), rows)             # the same on the device (mixed-radix CRT)
        assert np.array_equal(dev, got), first
        for j in range(256):
            if not np.array_equal(got[j], want[:, first + j]):
                bad.append(first + j)
    assert not bad, f"{len(bad)} of {S} encoded columns do not decrypt to core.Encode: {bad[:16]}"
    ctx.close()


def test_commit_reference_shape_every_leaf_decrypts(oracle):
    """TestLigeroE2E's shape (2048 x 1024, LogN = 12) with the heuristic's L = 10: Encode, rescale to level 1
    (processLeafParallel, fhe/ligero.go:145-155) and decrypt ALL 2048 leaf ciphertexts -- the verifier only
    ever opens 309 of them.  GPU decrypt (lumen_decrypt) and the oracle's agree."""
    rows, cols, rho, log_n = 2048, 1024, 2, 12
    P = reference_params(oracle, cols, log_n)
    assert (P.L, P.K) == (10, 2)
    P.seed(1212)
    ctx = make_context(P)
    sk, matrix, cts, zero = encrypt_witness(oracle, P, ctx, rows, cols, 0x12)
    S = cols * rho
    roots = oracle.field_roots(T_REF, S)
    ctx.field_set(roots)
    lvl1 = ctx.rescale(ctx.encode(cts, zero, rho), 2)
    scale = P.rescale_scale(P.L, 2)
    ctx.load_secret_key(sk)
    got = ctx.decrypt(lvl1, rows, scale)  # [S][rows]
    want = plain_encode_rows(oracle, matrix, rho, roots)
    bad = [j for j in range(S) if not np.array_equal(got[j], want[:, j])]
    assert not bad, f"{len(bad)} of {S} leaves do not decrypt to core.Encode: {bad[:16]}"
    sample = np.array([0, 1, 777, 1023, 1024, 2047])
    assert np.array_equal(P.decrypt_batch(sk, lvl1.download()[sample], rows, scale), got[sample])
    ctx.close()


@pytest.fixture(scope="module", params=list(CONFIGS))
def config(request, oracle):
    cols, log_n, chain = CONFIGS[request.param]
    P = reference_params(oracle, cols, log_n)
    assert (P.L, P.K) == (chain, 2), (request.param, P.L)
    P.seed(1000 + log_n * 10 + chain)
    ctx = make_context(P)
    sk = P.keygen_secret()
    yield request.param, P, ctx, sk
    ctx.close()


def test_config_rescale_and_leaf_digests(oracle, config):
    _, P, ctx, _ = config
    cts = random_cts(P, 3, P.L, seed=3)
    lvl1 = ctx.rescale(ctx.upload(cts), 2)
    ref_l1, ref_dig = P.commit_leaves(cts)
    assert np.array_equal(lvl1.download(), ref_l1)
    assert np.array_equal(ctx.leaf_digests(lvl1), ref_dig)
    for target in (1, P.L - 1):  # one step and all the way down
        ref = cts[0]
        while ref.shape[1] > target:
            ref = P.rescale(ref)
        assert np.array_equal(ctx.rescale(ctx.upload(cts[:1]), target).download()[0], ref), target


def test_config_encrypt_values_and_decrypt(oracle, config):
    """lumen_encrypt_values == the oracle's deterministic encryptor (QP + division by P), bit for bit,
    at the configuration's ring degree and chain; the ciphertexts decrypt to the columns."""
    _, P, ctx, sk = config
    pk = P.keygen_public(sk)
    ctx.load_public_key(pk)
    ctx.encoder_set(psi_T(P.logN))
    rng = np.random.default_rng(P.logN)
    rows = P.N
    vals = rng.integers(0, T_REF, size=(2, rows), dtype=np.uint64)
    seed = rng.integers(0, 256, size=32, dtype=np.uint8)
    got = ctx.encrypt_values(vals, seed, 5).download()
    for i in range(2):
        assert np.array_equal(got[i], P.encrypt_det(pk, P.encode(vals[i]), seed, 5 + i)), i
        assert np.array_equal(P.decrypt(sk, got[i], rows), vals[i])
    # fresh noise of an encryption over QP divided by P: delta0 + delta1*s, a few units
    ph = P.decrypt_phase(sk, ctx.encrypt_pk(None, 1, seed, 99).download()[0][:, :1])  # limb 0 is enough
    q0 = P.moduli[0]
    e = ph[0].astype(object) * pow(T_REF % q0, -1, q0) % q0  # undo the *T of lo_decrypt_phase
    e = np.array([int(x) if x < q0 // 2 else int(x) - q0 for x in e], dtype=np.float64)
    assert np.abs(e).max() < 8 * np.sqrt((1 + 2 * P.N / 3) / 12) + 8, np.abs(e).max()


def test_config_matrix_inner_sum_rows_eq_slots(oracle, config):
    """matrixInnerSumEval with rows = N (configs B, C, D; A has rows = N/2): log2(N/2) column rotations and
    the row swap on real keys, then Rescale to level 1 -- bit-exact, and slot 0 decrypts to <r, column>."""
    name, P, ctx, sk = config
    rows = P.N // 2 if name.startswith("A_") else P.N
    pk = P.keygen_public(sk)
    gl = P.inner_sum_galois_elements(rows)
    assert len(gl) == P.logN - (1 if rows < P.N else 0)
    evks = [P.keygen_galois(sk, g) for g in gl]
    for g, e in zip(gl, evks):
        ctx.load_galois_key(g, e)
    rng = np.random.default_rng(4)
    col = rng.integers(0, T_REF, size=rows, dtype=np.uint64)
    r = rng.integers(0, 2**63, size=rows, dtype=np.uint64)  # raw u64, as Prove samples it (ligero.go:202-203)
    cts = P.encrypt(pk, P.encode(col))[None]
    pt = P.encode(r)
    out = ctx.matrix_inner_sum(ctx.upload(cts), pt, rows)
    got = out.download()
    assert np.array_equal(got, P.matrix_inner_sum(cts, pt, rows, evks))
    want = int(np.sum(col.astype(object) * (r.astype(object) % T_REF)) % T_REF)
    scale = P.rescale_scale(P.L, 2)
    assert int(P.decrypt(sk, got[0], 1, scale)[0]) == want
    ctx.load_secret_key(sk)
    ctx.encoder_set(psi_T(P.logN))
    assert int(ctx.decrypt(out, 1, scale)[0, 0]) == want


def test_config_ring_switch_to_logn10(oracle, config):
    """BASELINE config 5's tail at every configuration's own parameters (two special primes: Lattigo's hybrid
    key switch with ONE RNS digit at level 0, no power-of-two digits -- tests/test_oracle_kat.py shows the
    reference's key-size logs say the same): RingSwitchNew into LogN = 10 on level-1 ciphertexts, from the
    full [beta][1][2][L+K][N] key a client posts, bit-exact vs the oracle."""
    _, P, ctx, sk = config
    pk = P.keygen_public(sk)
    rng = np.random.default_rng(9)
    cts = np.stack([P.rescale_to_level1(P.encrypt(pk, P.encode(rng.integers(0, T_REF, size=P.N, dtype=np.uint64))))
                    for _ in range(2)])
    sk_small = P.keygen_secret_small(10)
    key = P.keygen_ringswitch(sk, sk_small, 10)
    assert key.shape == ctx.ringswitch_key_shape() == ((P.L + 1) // 2, 1, 2, P.L + 2, P.N)
    assert key.size == oracle.lib.lo_evk_words(P.h)  # one Galois key's size: the "+ 5 / 7 / 15 / 29 MB" of the logs
    ctx.load_ringswitch_key(10, key)
    got = ctx.ring_switch(ctx.upload(cts))
    for c in range(2):
        assert np.array_equal(got[c], P.ring_switch(cts[c], key, 10)), c
    # No decryption check at these parameters: T ~ 2^57 under the single 58-bit modulus of level 0 leaves one
    # bit for noise -- the reference's own client prints "Ring switch is unstable, proof verification will fail"
    # (results/experimental/client/bench_*.txt).  The sub-ring decryption contract is tested with
    # TestRingSwitch's small T in tests/test_gpu_parity.py::test_ring_switch_matches_oracle.


@pytest.mark.parametrize("rows,cols", [(2048, 64), (2048, 1024)])
def test_plain_prover_on_the_device(oracle, rows, cols):
    """LigeroProveReference (fhe/ligero.go:799-953), the plain prover the client checks a decrypted proof
    against, on the same kernels: a context whose one modulus is T, column j of the plain matrix = one
    2 x 1 x (rows/2) "ciphertext".  core.Encode of every row, the byte-for-byte leaves
    (binary.Write of the column, ligero.go:866-872), the Merkle root, MatR / MatZ and the queried columns
    against the oracle's plain field code.  (2048 x 1024 is TestLigeroE2E's shape.)"""
    from lumenos_amd import params as lp
    from lumenos_amd.hip import Context
    from oracle.loader import Transcript
    rho, log_n = 2, (rows // 2).bit_length() - 1
    S = cols * rho
    psi = pow(lp.primitive_root(T_REF), (T_REF - 1) // (2 << log_n), T_REF)
    ctx = Context(log_n, [T_REF], [], [psi], T_REF)
    matrix = oracle.witness(rows, cols, T_REF)                                   # [rows][cols]
    columns = np.ascontiguousarray(matrix.T).reshape(cols, 2, 1, rows // 2)      # column j as lanes
    roots = oracle.field_roots(T_REF, S)
    ctx.field_set(roots)
    m = ctx.upload(columns)
    enc = ctx.encode(m, np.zeros((2, 1, rows // 2), dtype=np.uint64), rho)       # zero padding of core.Encode
    got = enc.download().reshape(S, rows)
    want = np.stack([oracle.plain_encode(matrix[i], rho, T_REF, roots) for i in range(rows)])  # [rows][S]
    assert np.array_equal(got, want.T), "core.Encode of the rows"
    ctx.leaf_format_set(b"", b"", b"")                                           # leaf = the column's bytes
    dig = ctx.leaf_digests(enc)
    for j in (0, 1, S // 2, S - 1):
        assert dig[j].tobytes() == oracle.sha256(want[:, j].astype("<u8").tobytes()), j
    _, root = ctx.merkle_build(dig)
    assert root == oracle.merkle(np.stack([np.frombuffer(oracle.sha256(want[:, j].astype("<u8").tobytes()), dtype=np.uint8)
                                           for j in range(S)]))[1]
    t = Transcript(oracle, "test")
    r = np.array([t.sample_u64("r") for _ in range(rows)], dtype=np.uint64)       # raw u64 words (SampleFields)
    mat_r = ctx.plain_inner_products(m, r)
    M, R = matrix.astype(object), r.astype(object) % T_REF
    assert [int(x) for x in mat_r[:64]] == [int(sum(M[:, j] * R) % T_REF) for j in range(min(cols, 64))], "MatR"
    b = np.array([pow(1, i, T_REF) for i in range(rows)], dtype=np.uint64)       # z = 1: b_i = (z^cols)^i = 1
    mat_z = ctx.plain_inner_products(m, b)
    assert [int(x) for x in mat_z[:16]] == [int(sum(M[:, j]) % T_REF) for j in range(16)], "MatZ"
    assert int(sum(int(x) for x in mat_z) % T_REF) == int(M.sum() % T_REF)       # P(1) of the witness polynomial
    if (rows, cols) == (2048, 1024):
        assert int(M.sum() % T_REF) == 59828798142202325                          # results/baseline/client/bench_2048x1024_12.txt:22
    idx = np.array([3, S - 1, 3, 0], dtype=np.uint32)
    assert np.array_equal(ctx.gather(enc, idx).download().reshape(4, rows), want.T[idx])
    ctx.close()


def test_wire_image_of_more_than_one_chunk_at_headline_size(oracle, config):
    """lumen_ct_serialize assembles the wire image in device chunks of 512 MiB: 1100 level-1 ciphertexts at
    N = 2^14 (577 MB, two chunks) in a format of odd lengths, straight into page-locked memory and through the
    pageable path, against bytes assembled here from the downloaded residues.  (The proof of the headline
    configuration is 4.46 GB: nine chunks for MatR alone.)"""
    from lumenos_amd.hip import pinned_bytes
    name, P, ctx, _ = config
    if not name.startswith("D_"):
        pytest.skip("one large case is enough")
    count, nl = 1100, 2
    head, poly, limb = b"H" * 281 + (2).to_bytes(8, "little"), b"p" * 7, b"l" * 3
    s = ctx.new_set(count, nl).fill_random(77)
    try:
        ctx.leaf_format_set(head, poly, limb)
        each = ctx.ct_serialized_size(nl)
        assert each == len(head) + 2 * (len(poly) + nl * (len(limb) + 8 * P.N)) and count * each > (512 << 20)
        host = s.download()
        want = np.empty((count, each), dtype=np.uint8)
        want[:, :len(head)] = np.frombuffer(head, dtype=np.uint8)
        off = len(head)
        for k in range(2):
            want[:, off:off + len(poly)] = np.frombuffer(poly, dtype=np.uint8)
            off += len(poly)
            for l in range(nl):
                want[:, off:off + len(limb)] = np.frombuffer(limb, dtype=np.uint8)
                off += len(limb)
                want[:, off:off + 8 * P.N] = host[:, k, l, :].astype("<u8").view(np.uint8).reshape(count, 8 * P.N)
                off += 8 * P.N
        assert off == each
        buf = pinned_bytes(count * each)
        ctx.ct_serialize_into(s, buf, wait=False)
        ctx.sync()
        assert np.array_equal(buf.reshape(count, each), want)
        # one ciphertext against the oracle's serialiser, and the pageable path over a chunk boundary
        assert buf[5 * each:6 * each].tobytes() == P.ct_serialize(host[5], (head, poly, limb))
        per = (512 << 20) // each
        assert ctx.ct_serialize(s, per - 2, 4) == want[per - 2:per + 2].tobytes()
        pageable = np.zeros(count * each, dtype=np.uint8)
        ctx.ct_serialize_into(s, pageable)
        assert np.array_equal(pageable.reshape(count, each), want)
    finally:
        ctx.leaf_format_set()
        s.free()


# results/baseline/client/bench_*.txt:22 "Received encrypted proof for P(x=1)=..."
P1_PUBLISHED = {"A_2048x1024": (2048, 59828798142202325), "B_4096x2048": (4096, 78852759954010476),
                "C_8192x4096": (8192, 125815544481056462), "D_16384x4096": (16384, 5538402014578059)}


def test_published_p1_through_the_encrypted_pipeline(oracle, config):
    """The one value the reference publishes about every benchmark run, P(x = 1), reproduced THROUGH THE
    CIPHERTEXTS at each configuration's full size and own parameters: the ChaCha20 witness (core/utils.go:46-82)
    -> Encoder.Encode + EncryptNew of all its columns on the device -> matrixInnerSumEval with the evaluation
    vector b_i = (z^cols)^i for z = 1 (cmd/client/main.go:41, fhe/ligero.go:210-216) -> decryption of slot 0 of
    every MatZ ciphertext on the device -> sum_j MatZ[j] * z^j = P(1), as Proof.Verify's evaluation claim
    computes it (ligero.go:569).  4096 columns x 16384 rows x 14 key switches at D: every kernel of the Prove
    path at the headline size, checked by a number on the reference's disk."""
    name, P, ctx, sk = config
    rows, want = P1_PUBLISHED[name]
    cols = CONFIGS[name][0]
    pk = P.keygen_public(sk)
    ctx.load_public_key(pk)
    ctx.load_secret_key(sk)
    ctx.encoder_set(psi_T(P.logN))
    matrix = oracle.witness(rows, cols, T_REF)  # [rows][cols]
    assert int(np.sum(matrix.astype(object))) % T_REF == want  # the plain KAT (tests/test_oracle_kat.py)
    columns = np.ascontiguousarray(matrix.T)
    seed = np.frombuffer(bytes(range(32)), dtype=np.uint8)
    cts = ctx.encrypt_values(columns, seed, 0)
    gl = P.inner_sum_galois_elements(rows)
    for g in gl:
        ctx.load_galois_key(g, P.keygen_galois(sk, g))
    b = np.ones(rows, dtype=np.uint64)  # z = 1: (z^cols)^i = 1
    mat_z = ctx.matrix_inner_sum(cts, P.encode(b), rows)
    assert mat_z.count == cols and mat_z.nl == 2
    vals = ctx.decrypt(mat_z, 1, P.rescale_scale(P.L, 2))[:, 0]
    assert int(np.sum(vals.astype(object))) % T_REF == want, "P(1) through the encrypted inner products"
    # and column by column against the witness
    assert np.array_equal(vals, (np.sum(matrix.astype(object), axis=0) % T_REF).astype(np.uint64))


def test_commit_at_full_size_every_leaf_decrypts_to_the_plain_encoding(oracle, config):
    """Commit's data path at every configuration's FULL size and own parameters (D: 4096 input columns -> 8192
    leaves of 16384 slots): witness -> device encryption -> fhe.Encode (one Enc(0) padding column) -> Rescale
    to level 1 -> every leaf ciphertext decrypted on the device must be the plain Reed-Solomon encoding of the
    witness rows (what TestEncode asserts at its one shape, fhe/code_test.go:110-116).  The plain side is
    core.Encode of every row run on the same kernels (a context whose one modulus is T), itself
    held to the oracle on a sample of rows; the Merkle root over the leaves' digests equals the oracle's tree
    over the same digests."""
    from lumenos_amd import params as lp
    from lumenos_amd.hip import Context
    name, P, ctx, sk = config
    rows = P1_PUBLISHED[name][0]
    cols, rho = CONFIGS[name][0], 2
    S = cols * rho
    pk = P.keygen_public(sk)
    ctx.load_public_key(pk)
    ctx.load_secret_key(sk)
    ctx.encoder_set(psi_T(P.logN))
    matrix = oracle.witness(rows, cols, T_REF)
    columns = np.ascontiguousarray(matrix.T)
    seed = np.frombuffer(bytes(range(7, 39)), dtype=np.uint8)
    cts = ctx.encrypt_values(columns, seed, 0)
    zero = ctx.encrypt_pk(None, 1, seed, cols).download()[0]
    roots = oracle.field_roots(T_REF, S)
    ctx.field_set(roots)
    enc = ctx.encode(cts, zero, rho)
    cts.free()
    lvl1 = ctx.rescale(enc, 2)
    enc.free()
    dig = ctx.leaf_digests(lvl1)
    assert ctx.merkle_build(dig)[1] == oracle.merkle(dig)[1]
    # the plain encoding of all rows on the device
    log_half = (rows // 2).bit_length() - 1
    pctx = Context(log_half, [T_REF], [], [lp.encoder_psi(T_REF, log_half)], T_REF)
    pctx.field_set(roots)
    pm = pctx.upload(columns.reshape(cols, 2, 1, rows // 2))
    want = pctx.encode(pm, np.zeros((2, 1, rows // 2), dtype=np.uint64), rho).download().reshape(S, rows)
    pctx.close()
    for i in (0, 1, rows // 2, rows - 1):  # the plain path against the oracle's core.Encode
        assert np.array_equal(want[:, i], oracle.plain_encode(matrix[i], rho, T_REF, roots)), i
    scale = P.rescale_scale(P.L, 2)
    bad = 0
    for first in range(0, S, 1024):  # in slices: 8192 x 16384 decoded values are 1 GB
        view = lvl1.slice(first, min(1024, S - first))
        got = ctx.decrypt(view, rows, scale)
        view.free()
        bad += int(np.any(got != want[first:first + got.shape[0]], axis=1).sum())
    assert bad == 0, f"{bad} of {S} leaves do not decrypt to the plain encoding"
    lvl1.free()
