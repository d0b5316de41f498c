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
