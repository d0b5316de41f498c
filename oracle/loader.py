"""ctypes loader for the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module (see oracle/lo_common.h).  Nothing under lumenos_amd/ does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# LUMEN_ORACLE_LIB: another build of the same sources (oracle/_san/liblumen_oracle.so from `make san`,
# tests/test_sanitizers.py)
_LIB = os.environ.get("LUMEN_ORACLE_LIB") or os.path.join(_HERE, "liblumen_oracle.so")

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
u8p = C.POINTER(C.c_uint8)


def build(force=False):
    if force or not os.path.exists(_LIB):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB


def _p64(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u64p)


def _p32(a):
    assert a.dtype == np.uint32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u32p)


def _p8(a):
    assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u8p)


class Oracle:
    """Thin numpy-facing wrapper; every method names the C function it calls."""

    def __init__(self):
        build()
        L = self.lib = C.CDLL(_LIB)
        vp = C.c_void_p
        sigs = {
            "lo_mulmod": (C.c_uint64, [C.c_uint64] * 3),
            "lo_powmod": (C.c_uint64, [C.c_uint64] * 3),
            "lo_invmod": (C.c_uint64, [C.c_uint64] * 2),
            "lo_is_prime": (C.c_int, [C.c_uint64]),
            "lo_primitive_root": (C.c_uint64, [C.c_uint64]),
            "lo_gen_primes": (C.c_int, [C.c_int, C.c_uint64, C.c_int, u64p, C.c_int, u64p]),
            "lo_field_roots_forward": (C.c_int, [C.c_uint64, C.c_uint32, u64p]),
            "lo_sqrt_factor": (C.c_uint32, [C.c_uint32]),
            "lo_plain_ntt": (None, [u64p, C.c_uint32, C.c_uint32, C.c_uint64, u64p, C.c_uint32]),
            "lo_plain_encode": (None, [u64p, C.c_uint32, C.c_uint32, C.c_uint64, u64p, C.c_uint32, u64p]),
            "lo_ntt_twiddle_trace": (C.c_size_t, [C.c_uint32, C.c_uint32, C.c_uint32, i32p, C.c_size_t]),
            "lo_omega8_cubed": (C.c_uint64, [C.c_uint64, u64p]),
            "lo_chacha20_xor": (None, [u8p, u8p, C.c_uint32, u8p, C.c_size_t]),
            "lo_witness_row_major": (None, [C.c_uint32, C.c_uint32, C.c_uint64, u64p]),
            "lo_sha256": (None, [u8p, C.c_size_t, u8p]),
            "lo_merkle_build": (C.c_size_t, [u8p, C.c_uint32, u8p, C.c_size_t, u8p]),
            "lo_merkle_path": (C.c_uint32, [u8p, C.c_uint32, C.c_uint32, u8p]),
            "lo_merkle_verify": (C.c_int, [u8p, u8p, C.c_uint32, u8p, C.c_uint32]),
            "lo_transcript_new": (vp, [C.c_char_p]),
            "lo_transcript_free": (None, [vp]),
            "lo_transcript_append": (None, [vp, C.c_char_p, u8p, C.c_uint32]),
            "lo_transcript_challenge": (None, [vp, C.c_char_p, u8p, C.c_uint32]),
            "lo_transcript_sample_u64": (C.c_uint64, [vp, C.c_char_p]),
            "lo_bgv_param_bits": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint64, i32p, i32p, i32p, i32p]),
            "lo_params_new": (vp, [C.c_uint32, C.c_uint32, C.c_uint32, u64p, C.c_uint64]),
            "lo_params_for_ntt": (vp, [C.c_uint32, C.c_uint32, C.c_uint64]),
            "lo_params_free": (None, [vp]),
            "lo_params_modulus": (C.c_uint64, [vp, C.c_uint32]),
            "lo_params_psi": (C.c_uint64, [vp, C.c_uint32]),
            "lo_params_L": (C.c_uint32, [vp]),
            "lo_params_K": (C.c_uint32, [vp]),
            "lo_limb_ntt": (None, [vp, C.c_uint32, u64p]),
            "lo_limb_intt": (None, [vp, C.c_uint32, u64p]),
            "lo_centered_scalar": (C.c_uint64, [C.c_uint64] * 3),
            "lo_ct_ntt": (None, [vp, u64p, C.c_uint32, C.c_uint32, C.c_uint32, u64p, C.c_uint32]),
            "lo_ct_encode": (None, [vp, u64p, C.c_uint32, C.c_uint32, C.c_uint32, u64p, u64p, C.c_uint32, u64p]),
            "lo_rescale": (None, [vp, u64p, C.c_uint32, u64p]),
            "lo_rescale_to_level1": (None, [vp, u64p, C.c_uint32, u64p]),
            "lo_rescale_scale": (C.c_uint64, [vp, C.c_uint32, C.c_uint32]),
            "lo_mul_plain": (None, [vp, u64p, u64p, C.c_uint32, u64p]),
            "lo_beta": (C.c_uint32, [vp, C.c_uint32]),
            "lo_evk_words": (C.c_size_t, [vp]),
            "lo_galois_element": (C.c_uint64, [vp, C.c_int64]),
            "lo_galois_row_swap": (C.c_uint64, [vp]),
            "lo_automorphism_index": (None, [vp, C.c_uint64, u32p]),
            "lo_automorphism": (None, [vp, u64p, C.c_uint32, C.c_uint64, u64p, u64p]),
            "lo_inner_sum_galois_elements": (C.c_uint32, [vp, C.c_uint32, u64p]),
            "lo_inner_sum": (None, [vp, u64p, C.c_uint32, C.c_uint32, C.POINTER(u64p), u64p]),
            "lo_rng_seed": (None, [vp, C.c_uint64]),
            "lo_keygen_secret": (None, [vp, vp, u64p]),
            "lo_keygen_public": (None, [vp, vp, u64p, u64p]),
            "lo_keygen_evk": (None, [vp, vp, u64p, u64p, u64p]),
            "lo_keygen_galois": (None, [vp, vp, u64p, C.c_uint64, u64p]),
            "lo_encode": (None, [vp, u64p, C.c_uint32, C.c_uint32, u64p]),
            "lo_encrypt_pk": (None, [vp, vp, u64p, u64p, C.c_uint32, u64p]),
            "lo_det_small": (None, [u8p, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_int8)]),
            "lo_encrypt_pk_det": (None, [vp, u64p, u64p, C.c_uint32, u8p, C.c_uint64, u64p]),
            "lo_decrypt_phase": (None, [vp, u64p, u64p, C.c_uint32, u64p]),
            "lo_decode_coeffs": (None, [vp, u64p, C.c_uint64, u64p, C.c_uint32]),
            "lo_decrypt_decode": (C.c_int, [vp, u64p, u64p, C.c_uint32, C.c_uint64, u64p, C.c_uint32]),
            "lo_decrypt_decode_batch": (C.c_int, [vp, u64p, u64p, C.c_uint32, C.c_uint32, C.c_uint64, u64p, C.c_uint32]),
            "lo_rs_num_digits": (C.c_uint32, [vp, C.c_uint32]),
            "lo_rs_key_shape": (None, [vp, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
            "lo_rs_key_words": (C.c_size_t, [vp, C.c_uint32]),
            "lo_keygen_secret_small": (None, [vp, vp, C.c_uint32, C.POINTER(C.c_int64)]),
            "lo_keygen_ringswitch": (None, [vp, vp, u64p, C.POINTER(C.c_int64), C.c_uint32, C.c_uint32, u64p]),
            "lo_ring_switch": (None, [vp, u64p, C.c_uint32, u64p, C.c_uint32, C.c_uint32, u64p]),
            "lo_decrypt_small_coeffs": (None, [vp, C.POINTER(C.c_int64), C.c_uint32, u64p, u64p]),
            "lo_decrypt_big_coeffs_l0": (None, [vp, u64p, u64p, C.c_uint32, u64p]),
            "lo_calculate_queries": (C.c_int, [C.c_double, C.c_int]),
            "lo_ct_serialized_size": (C.c_size_t, [C.c_uint32, C.c_uint32]),
            "lo_ct_serialize": (None, [u64p, C.c_uint32, C.c_uint32, u8p]),
            "lo_ct_serialized_size_fmt": (C.c_size_t, [vp, C.c_uint32, C.c_uint32]),
            "lo_ct_serialize_fmt": (None, [u64p, C.c_uint32, C.c_uint32, vp, u8p]),
            "lo_commit_leaves": (None, [vp, u64p, C.c_uint32, C.c_uint32, u64p, u8p]),
            "lo_commit_leaves_fmt": (None, [vp, u64p, C.c_uint32, C.c_uint32, vp, u64p, u8p]),
            "lo_matrix_inner_sum": (None, [vp, u64p, C.c_uint32, C.c_uint32, u64p, C.c_uint32, C.POINTER(u64p), u64p]),
            "lo_sample_query_indices": (None, [vp, C.c_uint32, C.c_uint32, u32p]),
            "lo_prove_b_vector": (None, [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, u64p]),
        }
        for name, (res, args) in sigs.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args

    # -- small helpers -----------------------------------------------------
    def field_roots(self, T, fieldN):
        r = np.zeros(fieldN, dtype=np.uint64)
        rc = self.lib.lo_field_roots_forward(T, fieldN, _p64(r))
        if rc:
            raise ValueError(f"lo_field_roots_forward rc={rc}")
        return r

    def witness(self, rows, cols, T):
        m = np.zeros((rows, cols), dtype=np.uint64)
        self.lib.lo_witness_row_major(rows, cols, T, _p64(m))
        return m

    def plain_encode(self, row, rho_inv, T, roots):
        row = np.ascontiguousarray(row, dtype=np.uint64)
        out = np.zeros(len(row) * rho_inv, dtype=np.uint64)
        self.lib.lo_plain_encode(_p64(row), len(row), rho_inv, T, _p64(roots), len(roots), _p64(out))
        return out

    def twiddle_trace(self, S, fieldN):
        cap = 1 << 17
        out = np.zeros(cap, dtype=np.int32)
        n = self.lib.lo_ntt_twiddle_trace(S, S, fieldN, out.ctypes.data_as(i32p), cap)
        assert n <= cap
        return out[:n].copy()

    def sha256(self, data: bytes):
        a = np.frombuffer(data, dtype=np.uint8).copy() if len(data) else np.zeros(1, np.uint8)
        out = np.zeros(32, dtype=np.uint8)
        self.lib.lo_sha256(_p8(a), len(data), _p8(out))
        return out.tobytes()

    def merkle(self, digests):
        d = np.ascontiguousarray(digests, dtype=np.uint8).reshape(-1, 32)
        n = d.shape[0]
        nodes = np.zeros((2 * n + 64, 32), dtype=np.uint8)
        root = np.zeros(32, dtype=np.uint8)
        total = self.lib.lo_merkle_build(_p8(d), n, _p8(nodes), nodes.shape[0], _p8(root))
        assert total > 0
        return nodes[:total].copy(), root.tobytes()

    def merkle_path(self, nodes, nleaves, index):
        path = np.zeros((64, 32), dtype=np.uint8)
        depth = self.lib.lo_merkle_path(_p8(nodes), nleaves, index, _p8(path))
        return path[:depth].copy()

    def merkle_verify(self, leaf_digest, path, root, index):
        ld = np.frombuffer(leaf_digest, dtype=np.uint8).copy()
        rt = np.frombuffer(root, dtype=np.uint8).copy()
        path = np.ascontiguousarray(path, dtype=np.uint8)
        return bool(self.lib.lo_merkle_verify(_p8(ld), _p8(path), path.shape[0], _p8(rt), index))


class CtFormat(C.Structure):
    """lo_ct_format: head | per poly: poly | per limb: limb | data (oracle/lo_common.h)"""
    _fields_ = [("head", C.c_char_p), ("poly", C.c_char_p), ("limb", C.c_char_p),
                ("head_len", C.c_uint32), ("poly_len", C.c_uint32), ("limb_len", C.c_uint32)]

    @classmethod
    def make(cls, head: bytes, poly: bytes, limb: bytes):
        f = cls(head, poly, limb, len(head), len(poly), len(limb))
        f._keep = (head, poly, limb)
        return f


class Transcript:
    def __init__(self, oracle, label: str):
        self.o = oracle
        self.h = oracle.lib.lo_transcript_new(label.encode())

    def append(self, label: str, msg: bytes):
        a = np.frombuffer(msg, dtype=np.uint8).copy() if msg else np.zeros(1, np.uint8)
        self.o.lib.lo_transcript_append(self.h, label.encode(), _p8(a), len(msg))

    def challenge(self, label: str, n: int) -> bytes:
        out = np.zeros(n, dtype=np.uint8)
        self.o.lib.lo_transcript_challenge(self.h, label.encode(), _p8(out), n)
        return out.tobytes()

    def sample_u64(self, label: str) -> int:
        return int(self.o.lib.lo_transcript_sample_u64(self.h, label.encode()))

    def __del__(self):
        try:
            self.o.lib.lo_transcript_free(self.h)
        except Exception:
            pass


class Params:
    """lo_params handle + the BGV harness (keys live as numpy arrays)."""

    def __init__(self, oracle, handle, logN):
        self.o = oracle
        self.h = handle
        self.logN = logN
        self.N = 1 << logN
        lib = oracle.lib
        self.L = lib.lo_params_L(handle)
        self.K = lib.lo_params_K(handle)
        self.moduli = [int(lib.lo_params_modulus(handle, i)) for i in range(self.L + self.K)]
        self.psi = [int(lib.lo_params_psi(handle, i)) for i in range(self.L + self.K)]
        self._rng = (C.c_uint64 * 4)()

    @classmethod
    def for_ntt(cls, oracle, cols, logN, T):
        h = oracle.lib.lo_params_for_ntt(cols, logN, T)
        if not h:
            raise ValueError("lo_params_for_ntt failed")
        p = cls(oracle, h, logN)
        p.T = T
        return p

    @classmethod
    def from_moduli(cls, oracle, logN, q, pmods, T):
        m = np.array(list(q) + list(pmods), dtype=np.uint64)
        h = oracle.lib.lo_params_new(logN, len(q), len(pmods), _p64(m), T)
        if not h:
            raise ValueError("lo_params_new failed")
        p = cls(oracle, h, logN)
        p.T = T
        return p

    def seed(self, s):
        self.o.lib.lo_rng_seed(C.byref(self._rng), s)

    def _r(self):
        # xoshiro with an all-zero state returns zeros for ever, and the rejection samplers then never return
        # (a test that forgets P.seed() hangs instead of failing): seed it on first use
        if not any(self._rng):
            self.seed(0x5EED)
        return C.cast(C.byref(self._rng), C.c_void_p)

    # transforms
    def limb_ntt(self, a, mi):
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        self.o.lib.lo_limb_ntt(self.h, mi, _p64(a))
        return a

    def limb_intt(self, a, mi):
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        self.o.lib.lo_limb_intt(self.h, mi, _p64(a))
        return a

    def _check_table(self, size, roots):
        """nttInner's hard-coded base cases read RootForwardUint64(4) / (8) whatever the table's size
        (fhe/ntt.go:60,134-143): with fewer roots the reference panics (index out of range)."""
        if size >= 4 and len(roots) <= 8:  # the highest entry any schedule reads is RootForward(8)
            tr = self.o.twiddle_trace(size, len(roots))
            need = max([8 if t < 0 else int(t) for t in tr], default=0)
            if need >= len(roots):
                raise IndexError(f"fhe.NTT of {size} values reads RootForward({need}): the field table has {len(roots)} "
                                 "entries (the reference panics here)")

    def ct_ntt(self, cts, size, roots):
        """cts: [count][2][nl][N] -> transformed copy (fhe.NTT)."""
        cts = np.ascontiguousarray(cts, dtype=np.uint64).copy()
        count, _, nl, N = cts.shape
        self._check_table(size, roots)
        self.o.lib.lo_ct_ntt(self.h, _p64(cts), count, nl, size, _p64(roots), len(roots))
        return cts

    def ct_encode(self, matrix, rho_inv, zero_ct, roots):
        matrix = np.ascontiguousarray(matrix, dtype=np.uint64)
        cols, _, nl, N = matrix.shape
        out = np.zeros((cols * rho_inv, 2, nl, N), dtype=np.uint64)
        zero_ct = np.ascontiguousarray(zero_ct, dtype=np.uint64)
        self._check_table(cols * rho_inv, roots)
        self.o.lib.lo_ct_encode(self.h, _p64(matrix), cols, nl, rho_inv, _p64(zero_ct), _p64(roots), len(roots), _p64(out))
        return out

    def rescale(self, ct):
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        _, nl, N = ct.shape
        out = np.zeros((2, nl - 1, N), dtype=np.uint64)
        self.o.lib.lo_rescale(self.h, _p64(ct), nl, _p64(out))
        return out

    def rescale_to_level1(self, ct):
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        _, nl, N = ct.shape
        out = np.zeros((2, min(nl, 2), N), dtype=np.uint64)
        self.o.lib.lo_rescale_to_level1(self.h, _p64(ct), nl, _p64(out))
        return out

    def rescale_scale(self, nl_from, nl_to):
        return int(self.o.lib.lo_rescale_scale(self.h, nl_from, nl_to))

    def mul_plain(self, ct, pt):
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        pt = np.ascontiguousarray(pt, dtype=np.uint64)
        out = np.zeros_like(ct)
        self.o.lib.lo_mul_plain(self.h, _p64(ct), _p64(pt), ct.shape[1], _p64(out))
        return out

    def beta(self, nl=None):
        return int(self.o.lib.lo_beta(self.h, self.L if nl is None else nl))

    def evk_shape(self):
        return (self.beta(), 2, self.L + self.K, self.N)

    def galois_element(self, k):
        return int(self.o.lib.lo_galois_element(self.h, k))

    def automorphism_index(self, gal_el):
        idx = np.zeros(self.N, dtype=np.uint32)
        self.o.lib.lo_automorphism_index(self.h, gal_el, _p32(idx))
        return idx

    def automorphism(self, ct, gal_el, evk):
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        out = np.zeros_like(ct)
        self.o.lib.lo_automorphism(self.h, _p64(ct), ct.shape[1], gal_el, _p64(evk), _p64(out))
        return out

    def inner_sum_galois_elements(self, n):
        g = np.zeros(64, dtype=np.uint64)
        cnt = self.o.lib.lo_inner_sum_galois_elements(self.h, n, _p64(g))
        return [int(x) for x in g[:cnt]]

    def _evk_ptrs(self, evks):
        arr = (u64p * len(evks))()
        for i, e in enumerate(evks):
            arr[i] = _p64(e)
        return arr

    def inner_sum(self, ct, n, evks):
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        out = np.zeros_like(ct)
        self.o.lib.lo_inner_sum(self.h, _p64(ct), ct.shape[1], n, self._evk_ptrs(evks), _p64(out))
        return out

    # BGV harness
    def keygen_secret(self):
        sk = np.zeros((self.L + self.K, self.N), dtype=np.uint64)
        self.o.lib.lo_keygen_secret(self.h, self._r(), _p64(sk))
        return sk

    def keygen_public(self, sk):
        pk = np.zeros((2, self.L + self.K, self.N), dtype=np.uint64)  # over QP, like rlwe.PublicKey
        self.o.lib.lo_keygen_public(self.h, self._r(), _p64(sk), _p64(pk))
        return pk

    def keygen_galois(self, sk, gal_el):
        evk = np.zeros(self.evk_shape(), dtype=np.uint64)
        self.o.lib.lo_keygen_galois(self.h, self._r(), _p64(sk), gal_el, _p64(evk))
        return evk

    def encode(self, values, nl=None):
        nl = self.L if nl is None else nl
        v = np.ascontiguousarray(values, dtype=np.uint64)
        pt = np.zeros((nl, self.N), dtype=np.uint64)
        self.o.lib.lo_encode(self.h, _p64(v), len(v), nl, _p64(pt))
        return pt

    def encrypt(self, pk, pt, nl=None):
        nl = self.L if nl is None else nl
        ct = np.zeros((2, nl, self.N), dtype=np.uint64)
        self.o.lib.lo_encrypt_pk(self.h, self._r(), _p64(pk), _p64(pt) if pt is not None else None, nl, _p64(ct))
        return ct

    def det_small(self, seed, index, stream):
        """Small polynomial `stream` (0: ternary u, 1/2: Gaussian e0/e1) of ciphertext `index` (lo_encdet.c)."""
        seed = np.ascontiguousarray(seed, dtype=np.uint8)
        assert seed.size == 32
        out = np.zeros(self.N, dtype=np.int8)
        self.o.lib.lo_det_small(_p8(seed), index, stream, self.N, out.ctypes.data_as(C.POINTER(C.c_int8)))
        return out

    def encrypt_det(self, pk, pt, seed, index, nl=None):
        """Deterministic pk encryption shared bit for bit with lumen_encrypt_pk."""
        nl = self.L if nl is None else nl
        seed = np.ascontiguousarray(seed, dtype=np.uint8)
        assert seed.size == 32
        ct = np.zeros((2, nl, self.N), dtype=np.uint64)
        self.o.lib.lo_encrypt_pk_det(self.h, _p64(pk), _p64(pt) if pt is not None else None, nl, _p8(seed), index,
                                     _p64(ct))
        return ct

    def decrypt_phase(self, sk, ct):
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        nl = ct.shape[1]
        ph = np.zeros((nl, self.N), dtype=np.uint64)
        self.o.lib.lo_decrypt_phase(self.h, _p64(sk), _p64(ct), nl, _p64(ph))
        return ph

    def decode_coeffs(self, m, scale, nvalues):
        m = np.ascontiguousarray(m, dtype=np.uint64)
        out = np.zeros(nvalues, dtype=np.uint64)
        self.o.lib.lo_decode_coeffs(self.h, _p64(m), scale, _p64(out), nvalues)
        return out

    def decrypt(self, sk, ct, nvalues, scale=1):
        """Decryptor.DecryptNew + Encoder.Decode at any level (exact CRT by Garner's mixed radix, in C)."""
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        out = np.zeros(nvalues, dtype=np.uint64)
        rc = self.o.lib.lo_decrypt_decode(self.h, _p64(sk), _p64(ct), ct.shape[1], scale, _p64(out), nvalues)
        assert rc == 0
        return out

    def decrypt_batch(self, sk, cts, nvalues, scale=1):
        """[count][2][nl][N] -> [count][nvalues], columns in parallel (OpenMP)."""
        cts = np.ascontiguousarray(cts, dtype=np.uint64)
        out = np.zeros((cts.shape[0], nvalues), dtype=np.uint64)
        rc = self.o.lib.lo_decrypt_decode_batch(self.h, _p64(sk), _p64(cts), cts.shape[0], cts.shape[2], scale,
                                                _p64(out), nvalues)
        assert rc == 0
        return out

    def decrypt_bigint(self, sk, ct, nvalues, scale=1):
        """The same through Python integers (cross-check of the Garner path)."""
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        nl = ct.shape[1]
        ph = self.decrypt_phase(sk, ct)
        mods = self.moduli[:nl]
        Q = 1
        for q in mods:
            Q *= q
        acc = np.zeros(self.N, dtype=object)
        for i, q in enumerate(mods):
            Qi = Q // q
            c = Qi * pow(Qi % q, -1, q)
            acc = (acc + ph[i].astype(object) * c) % Q
        half = Q // 2
        m = np.array([int(((y - Q) if y > half else y) % self.T) for y in acc], dtype=np.uint64)
        return self.decode_coeffs(m, scale, nvalues)

    # ring switch (fhe/ring_switch.go)
    def rs_num_digits(self, w=13):
        return int(self.o.lib.lo_rs_num_digits(self.h, w))

    def rs_key_shape(self, w=13):
        """(rns, pw2) of the ring-switch key's GadgetCiphertext.Value (lo_rs_key_shape)"""
        rns, pw2 = C.c_uint32(), C.c_uint32()
        self.o.lib.lo_rs_key_shape(self.h, w, C.byref(rns), C.byref(pw2))
        return rns.value, pw2.value

    def keygen_secret_small(self, logn_small):
        c = np.zeros(1 << logn_small, dtype=np.int64)
        self.o.lib.lo_keygen_secret_small(self.h, self._r(), logn_small, c.ctypes.data_as(C.POINTER(C.c_int64)))
        return c

    def small_secret_ntt(self, sk_small):
        """skNew of the SAME ring degree (TestRingSwitch) as NTT-domain residues [L+K][N], like keygen_secret's"""
        assert sk_small.size == self.N
        out = np.zeros((self.L + self.K, self.N), dtype=np.uint64)
        for i, q in enumerate(self.moduli):
            v = np.array([int(x) % q for x in sk_small], dtype=np.uint64)
            out[i] = self.limb_ntt(v, i)
        return out

    def keygen_ringswitch(self, sk, sk_small, logn_small, w=13):
        key = np.zeros((*self.rs_key_shape(w), 2, self.L + self.K, self.N), dtype=np.uint64)
        self.o.lib.lo_keygen_ringswitch(self.h, self._r(), _p64(sk), sk_small.ctypes.data_as(C.POINTER(C.c_int64)),
                                        logn_small, w, _p64(key))
        return key

    def ring_switch(self, ct, key, logn_small, w=13):
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        out = np.zeros((2, 1 << logn_small), dtype=np.uint64)
        self.o.lib.lo_ring_switch(self.h, _p64(ct), ct.shape[1], _p64(key), w, logn_small, _p64(out))
        return out

    def decrypt_small_coeffs(self, sk_small, logn_small, ct_small):
        ct_small = np.ascontiguousarray(ct_small, dtype=np.uint64)
        m = np.zeros(1 << logn_small, dtype=np.uint64)
        self.o.lib.lo_decrypt_small_coeffs(self.h, sk_small.ctypes.data_as(C.POINTER(C.c_int64)), logn_small,
                                           _p64(ct_small), _p64(m))
        return m

    def decrypt_big_coeffs_l0(self, sk, ct):
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        m = np.zeros(self.N, dtype=np.uint64)
        self.o.lib.lo_decrypt_big_coeffs_l0(self.h, _p64(sk), _p64(ct), ct.shape[1], _p64(m))
        return m

    def commit_leaves(self, encoded, fmt=None):
        """fmt: (head, poly, limb) byte strings of the serialisation layout, or None for the default framing"""
        encoded = np.ascontiguousarray(encoded, dtype=np.uint64)
        count, _, nl, N = encoded.shape
        level1 = np.zeros((count, 2, 2, N), dtype=np.uint64)
        digests = np.zeros((count, 32), dtype=np.uint8)
        f = CtFormat.make(*fmt) if fmt else None
        self.o.lib.lo_commit_leaves_fmt(self.h, _p64(encoded), count, nl, C.byref(f) if f else None, _p64(level1),
                                        _p8(digests))
        return level1, digests

    def matrix_inner_sum(self, matrix, pt, rows, evks):
        matrix = np.ascontiguousarray(matrix, dtype=np.uint64)
        cols, _, nl, N = matrix.shape
        out = np.zeros((cols, 2, min(nl, 2), N), dtype=np.uint64)
        self.o.lib.lo_matrix_inner_sum(self.h, _p64(matrix), cols, nl, _p64(pt), rows, self._evk_ptrs(evks), _p64(out))
        return out

    def ct_serialize(self, ct, fmt=None):
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        _, nl, N = ct.shape
        f = CtFormat.make(*fmt) if fmt else None
        fp = C.byref(f) if f else None
        sz = self.o.lib.lo_ct_serialized_size_fmt(fp, nl, N)
        out = np.zeros(sz, dtype=np.uint8)
        self.o.lib.lo_ct_serialize_fmt(_p64(ct), nl, N, fp, _p8(out))
        return out.tobytes()

    def __del__(self):
        try:
            self.o.lib.lo_params_free(self.h)
        except Exception:
            pass
