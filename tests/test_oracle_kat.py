"""CPU: the oracle against every known answer the reference holds on disk for this path
 and against the published vectors of the third-party primitives it restates."""
import hashlib

import numpy as np
import pytest

from helpers import T_REF
from oracle.loader import Transcript


# results/baseline/client/bench_{2048x1024_12,4096x2048_12,8192x4096_13,16384x4096_14}.txt:22
P1_KATS = [
    (2048, 1024, 59828798142202325),
    (4096, 2048, 78852759954010476),
    (8192, 4096, 125815544481056462),
]


@pytest.mark.parametrize("rows,cols,p1", P1_KATS)
def test_witness_matches_reference_p1(oracle, rows, cols, p1):
    """core/utils.go:46-82 witness recipe: P(x=1) = sum of all entries mod T."""
    m = oracle.witness(rows, cols, T_REF)
    assert int(np.sum(m.astype(object))) % T_REF == p1


def test_witness_p1_16384x4096_streamed(oracle):
    """Same KAT at the headline shape (results/baseline/client/bench_16384x4096_14.txt:22); the
    witness is a continuous keystream, so it is folded in row blocks."""
    import ctypes as C
    rows, cols, want = 16384, 4096, 5538402014578059
    key = np.zeros(32, np.uint8)
    key[0] = 1
    nonce = np.zeros(12, np.uint8)
    u8p = C.POINTER(C.c_uint8)
    acc, block_rows, counter = 0, 1024, 0
    for r0 in range(0, rows, block_rows):
        buf = np.zeros(block_rows * cols * 8, np.uint8)
        oracle.lib.lo_chacha20_xor(key.ctypes.data_as(u8p), nonce.ctypes.data_as(u8p), counter,
                                   buf.ctypes.data_as(u8p), len(buf))
        counter += len(buf) // 64
        vals = buf.view("<u8") % np.uint64(T_REF)
        acc = (acc + int(np.sum(vals.astype(object)))) % T_REF
    assert acc == want


def test_queries_is_309(oracle):
    """results/baseline/server/bench_*.txt:19 'Number of queried columns: 309' (ligero.go:65-71)"""
    assert oracle.lib.lo_calculate_queries(128.0, 2) == 309


@pytest.mark.parametrize("cols,log_n,chain", [(1024, 12, 10), (2048, 12, 11), (4096, 13, 12), (4096, 14, 12)])
def test_q_chain_length(oracle, cols, log_n, chain):
    """results/baseline/server/bench_*.txt:16 'ModQ chain length' (fhe/bfv.go:154-169)"""
    import ctypes as C
    logq = (C.c_int * 24)()
    logp = (C.c_int * 4)()
    nq, npp = C.c_int(), C.c_int()
    k = oracle.lib.lo_bgv_param_bits(cols, log_n, T_REF, logq, C.byref(nq), logp, C.byref(npp))
    assert k == chain and nq.value == chain and npp.value == 2
    assert list(logq[:chain]) == [58] + [56] * (chain - 1) and list(logp[:2]) == [55, 55]


def test_field_roots_values(oracle):
    """SURVEY Appendix B.5 derived values of core.PrimeField.RootsForward (field.go:171-194)."""
    r = oracle.field_roots(T_REF, 2048)
    assert int(r[0]) == 33554304 and int(r[1]) == 33218973335662200
    assert int(r[4]) == 95661681840738641 and int(r[8]) == 116325211982151034
    assert pow(int(r[8]), 3, T_REF) == 82769008105103124
    # independent of FieldN for the base-case indices
    r2 = oracle.field_roots(T_REF, 8192)
    assert [int(r2[i]) for i in (0, 1, 4, 8)] == [int(r[i]) for i in (0, 1, 4, 8)]


@pytest.mark.parametrize("S,count,digest,zeros", [
    (16, 17, "187859bbf91131d7acdb95b2cb72c969", None),
    (2048, 9217, "bba3a85047ebf4141fcc5387c8365313", 3988),
    (4096, 20481, "55d98011f0aebbc6aaba94b4a6bd4e45", 9274),
    (8192, 45057, "b62b04b0babd7e6f7fcdc2cd55954a03", 21866),
])
def test_ntt_twiddle_schedule_digest(oracle, S, count, digest, zeros):
    """program-order twiddle-index sequence of nttInner (ntt.go:20-281)."""
    tr = oracle.twiddle_trace(S, S)
    assert len(tr) == count
    assert hashlib.sha256(tr.astype("<i4").tobytes()).hexdigest().startswith(digest)
    if zeros is not None:
        assert int((tr == 0).sum()) == zeros
    if S == 16:
        assert list(tr[:12]) == [4, 4, 4, 4, 1, 2, 3, 2, 4, 6, 6, 12]


def test_sqrt_factor(oracle):
    """core/math.go:25-36"""
    assert [oracle.lib.lo_sqrt_factor(n) for n in (16, 32, 64, 128, 2048, 4096, 8192)] == [4, 4, 8, 8, 32, 64, 64]


def test_sha256_fips_vectors(oracle):
    for msg in (b"", b"abc", b"abcdbcdecdefdefgefghfghighijhijkijkljklmklmnlmnomnopnopq", b"a" * 1000, bytes(range(256)) * 9):
        assert oracle.sha256(msg) == hashlib.sha256(msg).digest()


def test_chacha20_rfc8439_block(oracle):
    """RFC 8439 section 2.4.2 (key 00..1f, nonce 00 00 00 00 00 00 00 4a 00 00 00 00, counter 1)."""
    import ctypes as C
    key = np.arange(32, dtype=np.uint8)
    nonce = np.array([0, 0, 0, 0, 0, 0, 0, 0x4A, 0, 0, 0, 0], dtype=np.uint8)
    pt = (b"Ladies and Gentlemen of the class of '99: If I could offer you only one tip for the future, "
          b"sunscreen would be it.")
    buf = np.frombuffer(pt, dtype=np.uint8).copy()
    u8p = C.POINTER(C.c_uint8)
    oracle.lib.lo_chacha20_xor(key.ctypes.data_as(u8p), nonce.ctypes.data_as(u8p), 1, buf.ctypes.data_as(u8p), len(buf))
    assert buf.tobytes().hex().startswith("6e2e359a2568f98041ba0728dd0d6981e97e7aec1d4360c20a27afccfd9fae0b")


def test_merlin_published_vector(oracle):
    """merlin crate test `equivalence_simple` (same vector in gtank/merlin v0.1.1, go.mod)."""
    t = Transcript(oracle, "test protocol")
    t.append("some label", b"some data")
    assert t.challenge("challenge", 32).hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"


def test_merkle_tree_and_paths(oracle):
    """core/tree.go:76-268: odd levels duplicate the last node; paths verify; tampering fails."""
    for n in (1, 2, 3, 5, 8, 37):
        leaves = [hashlib.sha256(bytes([i]) * 10).digest() for i in range(n)]
        dig = np.frombuffer(b"".join(leaves), dtype=np.uint8).reshape(n, 32)
        nodes, root = oracle.merkle(dig)
        # independent recomputation
        lvl = leaves
        while len(lvl) > 1:
            lvl = [hashlib.sha256(lvl[i] + (lvl[i + 1] if i + 1 < len(lvl) else lvl[i])).digest()
                   for i in range(0, len(lvl), 2)]
        assert root == lvl[0]
        for idx in range(n):
            path = oracle.merkle_path(nodes, n, idx)
            assert oracle.merkle_verify(leaves[idx], path, root, idx)
            if n > 1:
                bad = path.copy()
                bad[0, 0] ^= 1
                assert not oracle.merkle_verify(leaves[idx], bad, root, idx)


# ------------------------------------------------------------------ size logs: key material and ring-switched proofs
def humanize_bytes(s):
    """dustin/go-humanize Bytes(): SI units, one decimal below 10, none above (what fmt.Println shows in
    cmd/client/main.go:132 and fhe/ligero.go:672-692)."""
    import math
    if s < 10:
        return f"{s} B"
    e = math.floor(math.log(s, 1000))
    val = math.floor(s / 1000 ** e * 10 + 0.5) / 10
    return (f"{val:.1f} " if val < 10 else f"{val:.0f} ") + ["B", "kB", "MB", "GB", "TB"][e]


def base64_len(n):
    return 4 * ((n + 2) // 3)


# (rows, cols, LogN, "Marshaled keys length" baseline, with ring switch)
# results/baseline/client/bench_*.txt:19, results/experimental/client/bench_*.txt:20
SIZE_KATS = [
    (2048, 1024, 12, "69 MB", "74 MB"),
    (4096, 2048, 12, "103 MB", "110 MB"),
    (8192, 4096, 13, "237 MB", "252 MB"),
    (16384, 4096, 14, "504 MB", "533 MB"),
]


def _keys_request_len(pk, evks, rs_evk=None):
    """len(json.Marshal(KeysRequest{...})) (cmd/client/main.go:26-32,105-130): []byte fields are base64
    strings; without a ring switch the last two fields are null"""
    n = len('{"public_key":"","relinearization_key":"","rotation_keys":[],"ring_switch_evk":,"params_lit":}')
    n += base64_len(pk) + base64_len(evks[0])
    n += sum(base64_len(k) + 2 for k in evks[1:]) + max(len(evks) - 2, 0)
    if rs_evk is None:
        return n + 2 * len("null")
    return n + base64_len(rs_evk) + 2 + 150  # ParametersLiteral JSON: LogN, Q, P, PlaintextModulus...


@pytest.mark.parametrize("rows,cols,log_n,keys,keys_rs", SIZE_KATS)
def test_key_and_ring_switch_key_sizes_match_reference_logs(oracle, rows, cols, log_n, keys, keys_rs):
    """The reference's size logs pin three things no test of its own does:
      * the client generates pk + rlk + len(GaloisElementsForInnerSum(1, rows)) = 12 / 14 / 15 / 16 Galois keys,
        each beta x 1 x 2 polynomials over L+K limbs (no power-of-two digits);
      * the ring-switch key adds exactly ONE key of that same size -- so with two special primes its
        BaseTwoDecomposition = 13 produces no extra digits (a base-2^13 gadget would add five).
    Every serialised object carries a header of unknown exact length (MetaData, length words): the model
    must give the logged string for any header between 0 and 2 KiB per key (the recalled framing -- 8 bytes
    per limb, polynomial and vector -- is about 1.5 KiB for the largest key)."""
    from helpers import make_params
    from lumenos_amd import params as lp
    logq, logp = lp.bgv_param_bits(cols, log_n, T_REF)
    L, K, N = len(logq), len(logp), 1 << log_n
    P = make_params(oracle, log_n, L, num_p=K)
    beta = oracle.lib.lo_beta(P.h, L)
    assert beta == (L + K - 1) // K
    gal = lp.galois_elements_for_inner_sum(log_n, 1, rows)
    assert len(gal) == {2048: 12, 4096: 14, 8192: 15, 16384: 16}[rows]
    assert set(lp.galois_elements_used_by_inner_sum(log_n, rows)) <= set(gal)
    assert lp.galois_elements_used_by_inner_sum(log_n, rows) == P.inner_sum_galois_elements(rows)
    rns, pw2 = P.rs_key_shape(13)
    assert (rns, pw2) == (beta, 1)
    assert oracle.lib.lo_rs_key_words(P.h, 13) == oracle.lib.lo_evk_words(P.h)
    for hdr in (0, 2048):
        pk = 2 * (L + K) * N * 8 + hdr
        evk = beta * 2 * (L + K) * N * 8 + hdr
        rs_evk = rns * pw2 * 2 * (L + K) * N * 8 + hdr
        assert humanize_bytes(_keys_request_len(pk, [evk] * (1 + len(gal)))) == keys, hdr
        assert humanize_bytes(_keys_request_len(pk, [evk] * (1 + len(gal)), rs_evk)) == keys_rs, hdr
        # the gadget the previous rounds restated (ceil(58/13) = 5 digits) is ruled out by the same log
        assert humanize_bytes(_keys_request_len(pk, [evk] * (1 + len(gal)), 5 * rs_evk)) != keys_rs


# results/baseline/server/bench_*.txt:31-36 and results/experimental/server/bench_*.txt:32-37:
# (rows, cols, LogN, MatR, QueriedCols, whole proof) without and with the ring switch to LogN = 10
PROOF_SIZES = [
    (2048, 1024, 12, "135 MB", "41 MB", "310 MB", "17 MB", "41 MB", "75 MB"),
    (4096, 2048, 12, "269 MB", "41 MB", "579 MB", "34 MB", "41 MB", "109 MB"),
    (8192, 4096, 13, "1.1 GB", "81 MB", "2.2 GB", "68 MB", "81 MB", "218 MB"),
    (16384, 4096, 14, "2.1 GB", "162 MB", "4.5 GB", "68 MB", "162 MB", "299 MB"),
]


def proof_size_lines(h1, h0):
    """What EncryptedProof.WriteTo (fhe/ligero.go:659-705) prints for every configuration when a level-1
    ciphertext of the big ring serialises to 2*2*N*8 + h1 bytes and a ring-switched one (level 0, N = 2^10)
    to 2*1*1024*8 + h0: metadata (11 bytes) | MatR | MatZ | 309 queried level-1 columns | 309 paths of
    log2(2*cols) digests | root."""
    out = []
    for rows, cols, log_n, *_ in PROOF_SIZES:
        ct1, ct0 = 2 * 2 * (1 << log_n) * 8 + h1, 2 * 1 * 1024 * 8 + h0
        tail = 309 * ct1 + 309 * int(np.log2(2 * cols)) * 32 + 32
        out.append((rows, cols, log_n, humanize_bytes(cols * ct1), humanize_bytes(309 * ct1),
                    humanize_bytes(11 + 2 * cols * ct1 + tail), humanize_bytes(cols * ct0),
                    humanize_bytes(309 * ct1), humanize_bytes(11 + 2 * cols * ct0 + tail)))
    return out


def test_proof_sizes_match_reference_logs_and_bound_the_framing():
    """All 24 size lines the reference logs for its proofs are reproduced by
    `payload + framing`, and together they leave little room for the framing Lattigo's
    rlwe.Ciphertext.WriteTo adds (MetaData block + length words), which this build cannot read offline:
    325..444 bytes for a level-1 ciphertext, 252..351 for a ring-switched one.  Under the recalled framing
    (MetaData | LE64(2) | per polynomial LE64(limbs) | per limb LE64(N): MetaData + 56 resp. + 40 bytes) the
    MetaData block is 269..311 bytes -- tests/test_host_mirror.py holds the C++ mirror's block to that."""
    ok1 = [h1 for h1 in range(0, 1200) if any(proof_size_lines(h1, h0) == PROOF_SIZES for h0 in range(200, 400, 10))]
    assert (min(ok1), max(ok1)) == (325, 444)
    # the whole-proof lines couple the two: the bounds of h0 are reached at the ends of h1's range
    ok0 = [h0 for h0 in range(0, 1200) if any(proof_size_lines(h1, h0) == PROOF_SIZES for h1 in (325, 444))]
    assert (min(ok0), max(ok0)) == (252, 351)
    md = [m for m in range(0, 1200) if proof_size_lines(m + 56, m + 40) == PROOF_SIZES]
    assert (min(md), max(md)) == (269, 311)
    # a ring-switched ciphertext really is one limb of 2^10 words per polynomial: no other power of two fits
    for n_small in (512, 2048):
        assert humanize_bytes(1024 * (2 * n_small * 8 + 300)) != "17 MB"


def test_mulmod_against_python_integers(oracle):
    """oracle/lo_modarith.c lo_mulmod against Python integers: the path's moduli, T, tiny and huge moduli, operands at
    and beyond the modulus."""
    import ctypes as C
    lib = oracle.lib
    lib.lo_mulmod.restype, lib.lo_mulmod.argtypes = C.c_uint64, [C.c_uint64] * 3
    rng = np.random.default_rng(2025)
    qs = [2, 3, 97, 65537, (1 << 31) - 1, 144115188075593729, (1 << 58) + 425985, (1 << 56) - 557055, (1 << 55) + 425985,
          (1 << 61) - 1, (1 << 62) - 57, (1 << 62) + 135, (1 << 64) - 59]
    for q in qs:
        for k in range(2000):
            a, b = (int(x) for x in rng.integers(0, 1 << 64, size=2, dtype=np.uint64))
            if k % 3 == 0:
                a, b = a % q, b % q
            if k % 11 == 0:
                a = b = q - 1
            assert lib.lo_mulmod(a, b, q) == a * b % q, (a, b, q)
