"""ctypes binding of the C-ABI library declared in include/lumenos_hip.h.

This is plumbing for the Python test/bench harness only; the product is the
shared library (lumenos_amd/csrc/liblumenos_hip.so).  There is no CPU
fallback: if the library is missing, or no HIP device is visible, calls raise.
"""
import ctypes as C
import os

import numpy as np

from . import _build

_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)
_u8p = C.POINTER(C.c_uint8)
LUMEN_MAX_LIMBS = 24
LUMEN_ABI_VERSION = 4


class LumenError(RuntimeError):
    pass


class ParamsDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("log_n", C.c_uint32),
        ("num_q", C.c_uint32),
        ("num_p", C.c_uint32),
        ("plaintext_modulus", C.c_uint64),
        ("moduli", C.c_uint64 * LUMEN_MAX_LIMBS),
        ("psi", C.c_uint64 * LUMEN_MAX_LIMBS),
        ("device", C.c_int32),
    ]


# every symbol include/lumenos_hip.h declares: name -> (restype, argtypes)
_vp = C.c_void_p
_vpp = C.POINTER(C.c_void_p)
SYMBOLS = {
    "lumen_ctx_create": (C.c_int, [C.POINTER(ParamsDesc), _vpp]),
    "lumen_ctx_destroy": (None, [_vp]),
    "lumen_ctx_clone": (C.c_int, [_vp, _vpp]),
    "lumen_host_alloc": (_vp, [C.c_size_t]),
    "lumen_host_free": (None, [_vp]),
    "lumen_host_gather": (C.c_int, [_u64p, _vpp, C.c_size_t, C.c_size_t, C.c_uint32]),
    "lumen_host_scatter": (C.c_int, [_u64p, _vpp, C.c_size_t, C.c_size_t, C.c_uint32]),
    "lumen_last_error": (C.c_char_p, [_vp]),
    "lumen_ctx_trim": (C.c_int, [_vp]),
    "lumen_ctx_wait": (C.c_int, [_vp, _vp]),
    "lumen_ctx_set_tuning": (C.c_int, [_vp, C.c_char_p, C.c_long]),
    "lumen_test_allow_shared_device_rccl": (C.c_int, [_vp, C.c_int]),
    "lumen_sync": (C.c_int, [_vp]),
    "lumen_mul_counter": (C.c_uint64, [_vp]),
    "lumen_set_create": (C.c_int, [_vp, C.c_uint32, C.c_uint32, _vpp]),
    "lumen_set_create_lanes": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.c_uint32, _vpp]),
    "lumen_set_log_world": (C.c_uint32, [_vp]),
    "lumen_lanes_split": (C.c_int, [_vp, _vp, C.c_uint32, _vpp]),
    "lumen_lanes_assemble": (C.c_int, [_vp, _vp, _vpp]),
    "lumen_leaf_digests_end_device": (C.c_int, [_vp, _vpp]),
    "lumen_merkle_root_device": (C.c_int, [_vp, _vp, C.c_uint32, _u8p]),
    "lumen_set_destroy": (None, [_vp, _vp]),
    "lumen_set_count": (C.c_uint32, [_vp]),
    "lumen_set_limbs": (C.c_uint32, [_vp]),
    "lumen_set_device_ptr": (_vp, [_vp]),
    "lumen_set_slice": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, _vpp]),
    "lumen_set_upload": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, _u64p]),
    "lumen_set_download": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, _u64p]),
    "lumen_set_fill_random": (C.c_int, [_vp, _vp, C.c_uint64]),
    "lumen_set_ntt": (C.c_int, [_vp, _vp, C.c_int]),
    "lumen_field_set": (C.c_int, [_vp, _u64p, C.c_uint32]),
    "lumen_ct_ntt": (C.c_int, [_vp, _vp, C.c_uint32]),
    "lumen_encode": (C.c_int, [_vp, _vp, _u64p, C.c_uint32, _vpp]),
    "lumen_encode_shard": (C.c_int, [_vp, _vp, _u64p, C.c_uint32, C.c_uint32, C.c_uint32, _vpp, _u32p, _u32p]),
    "lumen_rescale": (C.c_int, [_vp, _vp, C.c_uint32, _vpp]),
    "lumen_leaf_format_set": (C.c_int, [_vp, _u8p, C.c_uint32, _u8p, C.c_uint32, _u8p, C.c_uint32]),
    "lumen_ct_serialized_size": (C.c_size_t, [_vp, C.c_uint32]),
    "lumen_ct_serialize": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, _u8p, C.c_size_t]),
    "lumen_ct_serialize_async": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, _u8p, C.c_size_t]),
    "lumen_ct_deserialize": (C.c_int, [_vp, _u8p, C.c_size_t, C.c_uint32, C.c_uint32, _vpp]),
    "lumen_leaf_digests": (C.c_int, [_vp, _vp, _u8p]),
    "lumen_load_public_key": (C.c_int, [_vp, _u64p]),
    "lumen_encrypt_pk": (C.c_int, [_vp, _u64p, C.c_uint32, _u8p, C.c_uint64, C.POINTER(_vp)]),
    "lumen_encoder_set": (C.c_int, [_vp, C.c_uint64]),
    "lumen_load_secret_key": (C.c_int, [_vp, _u64p]),
    "lumen_decrypt": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint32, _u64p]),
    "lumen_encrypt_values": (C.c_int, [_vp, _u64p, C.c_uint32, C.c_uint32, _u8p, C.c_uint64, C.POINTER(_vp)]),
    "lumen_leaf_digests_begin": (C.c_int, [_vp, _vp]),
    "lumen_leaf_digests_end": (C.c_int, [_vp, _u8p]),
    "lumen_merkle_build": (C.c_int, [_vp, _u8p, C.c_uint32, _u8p, C.c_size_t, C.POINTER(C.c_size_t), _u8p]),
    "lumen_load_galois_key": (C.c_int, [_vp, C.c_uint64, _u64p]),
    "lumen_load_galois_key_ex": (C.c_int, [_vp, C.c_uint64, _u64p, C.c_uint32]),
    "lumen_inner_sum_galois_elements": (C.c_uint32, [_vp, C.c_uint32, _u64p]),
    "lumen_matrix_inner_sum": (C.c_int, [_vp, _vp, _u64p, C.c_uint32, _vpp]),
    "lumen_mul_plain": (C.c_int, [_vp, _vp, _u64p, _vpp]),
    "lumen_inner_sum": (C.c_int, [_vp, _vp, C.c_uint32, _vpp]),
    "lumen_gather": (C.c_int, [_vp, _vp, _u32p, C.c_uint32, _vpp]),
    "lumen_plain_inner_products": (C.c_int, [_vp, _vp, _u64p, _u64p]),
    "lumen_ringswitch_rns_digits": (C.c_uint32, [_vp]),
    "lumen_ringswitch_digits": (C.c_uint32, [_vp, C.c_uint32]),
    "lumen_load_ringswitch_key": (C.c_int, [_vp, C.c_uint32, C.c_uint32, _u64p, C.c_size_t]),
    "lumen_ring_switch": (C.c_int, [_vp, _vp, _u64p]),
    "lumen_group_create": (C.c_int, [_vpp, C.c_uint32, C.c_uint32, _vpp]),
    "lumen_group_unique_id": (C.c_int, [_u8p]),
    "lumen_group_create_rank": (C.c_int, [_vp, C.c_uint32, C.c_uint32, _u8p, _vpp]),
    "lumen_group_destroy": (None, [_vp]),
    "lumen_group_world": (C.c_uint32, [_vp]),
    "lumen_group_local": (C.c_uint32, [_vp]),
    "lumen_group_transport": (C.c_char_p, [_vp]),
    "lumen_group_transport_note": (C.c_char_p, [_vp]),
    "lumen_group_rccl_ranks": (C.c_uint32, [_vp]),
    "lumen_group_sync": (C.c_int, [_vp]),
    "lumen_group_all_to_all": (C.c_int, [_vp, _vpp, _vpp]),
    "lumen_group_upload": (C.c_int, [_vp, _vpp, _vpp]),
    "lumen_group_download": (C.c_int, [_vp, _vpp, _vpp]),
    "lumen_group_encode": (C.c_int, [_vp, _vpp, _u64p, C.c_uint32, _vpp]),
    "lumen_group_all_gather_digests": (C.c_int, [_vp]),
    "lumen_group_merkle_root": (C.c_int, [_vp, _u8p]),
    "lumen_group_digests": (C.c_int, [_vp, _u8p, C.c_size_t, _u32p]),
    "lumen_group_gather": (C.c_int, [_vp, _vpp, _u32p, C.c_uint32, _vpp]),
    "lumen_group_stats": (C.c_int, [_vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "lumen_group_stats_reset": (C.c_int, [_vp]),
    "lumen_ctx_scratch_info": (C.c_int, [_vp, C.c_char_p, _vpp, C.POINTER(C.c_size_t)]),
    "lumen_ks_mac_probe": (C.c_int, [_vp, C.c_uint32, _vp, _vp, _vp, _vp, C.c_uint32, C.POINTER(C.c_float)]),
    "lumen_timer_start": (C.c_int, [_vp]),
    "lumen_timer_stop": (C.c_int, [_vp, C.POINTER(C.c_float)]),
    "lumen_prof_enable": (C.c_int, [_vp, C.c_int]),
    "lumen_prof_read": (C.c_int, [_vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "lumen_prof_reset": (C.c_int, [_vp]),
    "lumen_prof_names": (C.c_size_t, [_vp, C.c_char_p, C.c_size_t]),
}

_lib = None


def load(build=True):
    """Load liblumenos_hip.so (building it first if sources are newer)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("LUMEN_HIP_LIB")  # A/B runs of two builds of this same library
    if not path:
        path = _build.LIB
        if build:
            path = _build.build()
    if not os.path.exists(path):
        raise LumenError(f"{path} is missing: build it with `python -m lumenos_amd._build` "
                         "(there is no CPU fallback for the HIP path)")
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def pinned_bytes(n):
    """uint8 array of n bytes in page-locked host memory (the proof's wire image)"""
    return pinned_empty(((n + 7) // 8,)).view(np.uint8)[:n]


def pinned_empty(shape):
    """uint64 array in page-locked host memory (lumen_host_alloc): uploads/downloads of it are plain DMA."""
    lib = load()
    n = int(np.prod(shape))
    p = lib.lumen_host_alloc(max(n, 1) * 8)
    if not p:
        raise LumenError(lib.lumen_last_error(None).decode())
    buf = (C.c_uint64 * max(n, 1)).from_address(p)
    arr = np.frombuffer(buf, dtype=np.uint64, count=n).reshape(shape)
    _pinned_keep[arr.ctypes.data] = p
    return arr


def host_gather(dst, limbs, threads=0):
    """lumen_host_gather: `limbs` (a list of separately allocated uint64 arrays of equal length) -> the flat array
    `dst`, on `threads` host threads"""
    lib = load()
    n, words = len(limbs), limbs[0].size
    assert dst.dtype == np.uint64 and dst.flags["C_CONTIGUOUS"] and dst.size == n * words
    ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in limbs])
    if lib.lumen_host_gather(_p64(dst.reshape(-1)), ptrs, n, words, threads):
        raise LumenError(lib.lumen_last_error(None).decode())


def host_scatter(src, limbs, threads=0):
    lib = load()
    n, words = len(limbs), limbs[0].size
    assert src.dtype == np.uint64 and src.flags["C_CONTIGUOUS"] and src.size == n * words
    ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in limbs])
    if lib.lumen_host_scatter(_p64(src.reshape(-1)), ptrs, n, words, threads):
        raise LumenError(lib.lumen_last_error(None).decode())


def pinned_free(arr):
    p = _pinned_keep.pop(arr.ctypes.data, None)
    if p:
        load().lumen_host_free(p)


_pinned_keep = {}


def _p64(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"], "need a contiguous uint64 array"
    return a.ctypes.data_as(_u64p)


class DeviceSet:
    """An HBM-resident array of ciphertexts [count][2][nl][N] (lumen_set)."""

    def __init__(self, ctx, handle):
        self.ctx = ctx
        self.h = handle
        lib = ctx.lib
        self.count = lib.lumen_set_count(handle)
        self.nl = lib.lumen_set_limbs(handle)
        self.log_world = lib.lumen_set_log_world(handle)  # > 0: a lane shard, limbs of N >> log_world words

    @property
    def shape(self):
        return (self.count, 2, self.nl, self.ctx.N >> self.log_world)

    @property
    def device_ptr(self):
        return self.ctx.lib.lumen_set_device_ptr(self.h)

    @property
    def nbytes(self):
        return int(np.prod(self.shape)) * 8

    def upload(self, host, first=0):
        host = np.ascontiguousarray(host, dtype=np.uint64)
        n = host.shape[0]
        assert host.shape[1:] == self.shape[1:], (host.shape, self.shape)
        self.ctx._ck(self.ctx.lib.lumen_set_upload(self.ctx.h, self.h, first, n, _p64(host)))
        return self

    def download(self, first=0, n=None):
        n = self.count - first if n is None else n
        out = np.empty((n,) + self.shape[1:], dtype=np.uint64)
        if n:
            self.ctx._ck(self.ctx.lib.lumen_set_download(self.ctx.h, self.h, first, n, _p64(out)))
        return out

    def download_into(self, out, first=0):
        """into an existing (e.g. pinned) array of n ciphertexts"""
        assert out.dtype == np.uint64 and out.flags["C_CONTIGUOUS"] and out.shape[1:] == self.shape[1:]
        if out.shape[0]:
            self.ctx._ck(self.ctx.lib.lumen_set_download(self.ctx.h, self.h, first, out.shape[0], _p64(out)))
        return out

    def slice(self, first, n):
        h = C.c_void_p()
        self.ctx._ck(self.ctx.lib.lumen_set_slice(self.ctx.h, self.h, first, n, C.byref(h)))
        v = DeviceSet(self.ctx, h)
        v._parent = self  # keep the storage alive
        return v

    def fill_random(self, seed):
        self.ctx._ck(self.ctx.lib.lumen_set_fill_random(self.ctx.h, self.h, seed))
        return self

    def free(self):
        if self.h:
            # a set outliving its context (interpreter shutdown order) is released without pooling
            self.ctx.lib.lumen_set_destroy(self.ctx.h if self.ctx.h else None, self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """lumen_ctx: the device-side counterpart of fhe.ServerBFV (fhe/bfv.go:13-58)."""

    def __init__(self, log_n, q, p, psi, plaintext_modulus, device=0):
        self.lib = load()
        d = ParamsDesc()
        d.abi_version = LUMEN_ABI_VERSION
        d.log_n = log_n
        d.num_q = len(q)
        d.num_p = len(p)
        d.plaintext_modulus = plaintext_modulus
        for i, m in enumerate(list(q) + list(p)):
            d.moduli[i] = m
        for i, r in enumerate(psi):
            d.psi[i] = r
        d.device = device
        h = C.c_void_p()
        rc = self.lib.lumen_ctx_create(C.byref(d), C.byref(h))
        if rc:
            raise LumenError(self.lib.lumen_last_error(None).decode())
        self.h = h
        self.device = device  # HIP device ordinal the context lives on
        self.log_n, self.N = log_n, 1 << log_n
        self.L, self.K = len(q), len(p)
        self.q, self.p, self.T = list(q), list(p), plaintext_modulus

    _pending_leaves = 0

    def clone(self):
        """ServerBFV.CopyNew: a context sharing this one's tables and keys, with its own streams and scratch."""
        h = C.c_void_p()
        self._ck(self.lib.lumen_ctx_clone(self.h, C.byref(h)))
        c = object.__new__(Context)
        c.lib, c.h = self.lib, h
        c.device = self.device
        c.log_n, c.N, c.L, c.K = self.log_n, self.N, self.L, self.K
        c.q, c.p, c.T = self.q, self.p, self.T
        if hasattr(self, "_rs_logn"):
            c._rs_logn = self._rs_logn
        return c

    def _ck(self, rc):
        if rc:
            raise LumenError(self.lib.lumen_last_error(self.h).decode())

    def close(self):
        if self.h:
            self.lib.lumen_ctx_destroy(self.h)
            self.h = None

    def sync(self):
        self._ck(self.lib.lumen_sync(self.h))

    def trim(self):
        """hand the context's pooled set storage and scratch buffers back to the driver"""
        self._ck(self.lib.lumen_ctx_trim(self.h))

    def wait_for(self, other):
        """this context's stream waits (on the device) for everything enqueued on `other` so far"""
        self._ck(self.lib.lumen_ctx_wait(self.h, other.h))

    def set_tuning(self, name, value):
        """A/B switch of the tools and tests (lumen_ctx_set_tuning); the environment is only read at creation"""
        self._ck(self.lib.lumen_ctx_set_tuning(self.h, name.encode(), int(value)))

    def test_allow_shared_device_rccl(self, on=True):
        """test hook: lumen_group_create lets the RCCL transport through although ranks share a device (the test double)"""
        self._ck(self.lib.lumen_test_allow_shared_device_rccl(self.h, 1 if on else 0))

    def scratch_info(self, name):
        """-> (device address or None, bytes) of a named scratch buffer of the context (placement diagnostics)"""
        p, n = C.c_void_p(), C.c_size_t()
        self._ck(self.lib.lumen_ctx_scratch_info(self.h, name.encode(), C.byref(p), C.byref(n)))
        return p.value, n.value

    def ks_mac_probe(self, batch=0, ext=None, acc=None, key=None, u=None, reps=100):
        """ms per launch of the key switch's gadget product alone on the given device addresses (None = the
        library's own blocks)"""
        ms = C.c_float()
        self._ck(self.lib.lumen_ks_mac_probe(self.h, batch, C.c_void_p(ext), C.c_void_p(acc), C.c_void_p(key),
                                             C.c_void_p(u), reps, C.byref(ms)))
        return ms.value

    def upload_into(self, s, host, first=0):
        """upload ciphertexts [first, first + len(host)) of a set (possibly another context's) on THIS context's
        stream: a clone feeds the producer's input set while the producer computes on what already arrived"""
        assert host.dtype == np.uint64 and host.flags["C_CONTIGUOUS"] and host.shape[1:] == s.shape[1:]
        if host.shape[0]:
            self._ck(self.lib.lumen_set_upload(self.h, s.h, first, host.shape[0], _p64(host)))

    def download_into(self, s, out, first=0):
        """download a set (possibly created by another context of this device) on THIS context's stream"""
        assert out.dtype == np.uint64 and out.flags["C_CONTIGUOUS"] and out.shape[1:] == s.shape[1:]
        if out.shape[0]:
            self._ck(self.lib.lumen_set_download(self.h, s.h, first, out.shape[0], _p64(out)))
        return out

    def new_set(self, count, nl):
        h = C.c_void_p()
        self._ck(self.lib.lumen_set_create(self.h, count, nl, C.byref(h)))
        return DeviceSet(self, h)

    def upload(self, host):
        host = np.ascontiguousarray(host, dtype=np.uint64)
        s = self.new_set(host.shape[0], host.shape[2])
        return s.upload(host)

    # ---- lane shards (multi-GPU Encode)
    def new_set_lanes(self, count, nl, log_world):
        h = C.c_void_p()
        self._ck(self.lib.lumen_set_create_lanes(self.h, count, nl, log_world, C.byref(h)))
        return DeviceSet(self, h)

    def upload_lanes(self, host, log_world):
        """host: [count][2][nl][N >> log_world]"""
        host = np.ascontiguousarray(host, dtype=np.uint64)
        assert host.shape[3] == self.N >> log_world
        return self.new_set_lanes(host.shape[0], host.shape[2], log_world).upload(host)

    def lanes_split(self, columns, log_world):
        h = C.c_void_p()
        self._ck(self.lib.lumen_lanes_split(self.h, columns.h, log_world, C.byref(h)))
        return DeviceSet(self, h)

    def lanes_assemble(self, lanes):
        h = C.c_void_p()
        self._ck(self.lib.lumen_lanes_assemble(self.h, lanes.h, C.byref(h)))
        return DeviceSet(self, h)

    def leaf_digests_end_device(self):
        """-> (device pointer, count): the digests stay in HBM (the buffer an all-gather reads)"""
        p = C.c_void_p()
        n = self._pending_leaves
        self._ck(self.lib.lumen_leaf_digests_end_device(self.h, C.byref(p)))
        self._pending_leaves = 0
        return p.value, n

    def merkle_root_device(self, dev_ptr, n_leaves):
        root = np.zeros(32, dtype=np.uint8)
        self._ck(self.lib.lumen_merkle_root_device(self.h, C.c_void_p(dev_ptr), n_leaves, root.ctypes.data_as(_u8p)))
        return root.tobytes()

    def set_ntt(self, s, inverse=False):
        self._ck(self.lib.lumen_set_ntt(self.h, s.h, 1 if inverse else 0))

    def field_set(self, roots):
        roots = np.ascontiguousarray(roots, dtype=np.uint64)
        self._ck(self.lib.lumen_field_set(self.h, _p64(roots), len(roots)))

    def ct_ntt(self, s, size):
        self._ck(self.lib.lumen_ct_ntt(self.h, s.h, size))

    def encode(self, matrix, zero_ct, rho_inv):
        zero_ct = np.ascontiguousarray(zero_ct, dtype=np.uint64)
        h = C.c_void_p()
        self._ck(self.lib.lumen_encode(self.h, matrix.h, _p64(zero_ct), rho_inv, C.byref(h)))
        return DeviceSet(self, h)

    def encode_shard(self, matrix, zero_ct, rho_inv, rank, world):
        """-> (DeviceSet of this rank's encoded columns, their global indices, ascending)"""
        zero_ct = np.ascontiguousarray(zero_ct, dtype=np.uint64)
        idx = np.zeros(matrix.count * rho_inv, dtype=np.uint32)
        n = C.c_uint32()
        h = C.c_void_p()
        self._ck(self.lib.lumen_encode_shard(self.h, matrix.h, _p64(zero_ct), rho_inv, rank, world, C.byref(h),
                                             idx.ctypes.data_as(_u32p), C.byref(n)))
        return DeviceSet(self, h), idx[:n.value].copy()

    def rescale(self, s, target_limbs=2):
        h = C.c_void_p()
        self._ck(self.lib.lumen_rescale(self.h, s.h, target_limbs, C.byref(h)))
        return DeviceSet(self, h)

    def leaf_format_set(self, head=None, poly_head=None, limb_head=None):
        """Serialisation layout of a ciphertext (lumen_leaf_format_set); all None: the default framing."""
        if head is None and poly_head is None and limb_head is None:
            self._ck(self.lib.lumen_leaf_format_set(self.h, None, 0, None, 0, None, 0))
            return
        segs = [np.frombuffer(bytes(x or b""), dtype=np.uint8).copy() if len(x or b"") else np.zeros(1, np.uint8)
                for x in (head, poly_head, limb_head)]
        lens = [len(x or b"") for x in (head, poly_head, limb_head)]
        self._ck(self.lib.lumen_leaf_format_set(self.h, segs[0].ctypes.data_as(_u8p), lens[0],
                                                segs[1].ctypes.data_as(_u8p), lens[1],
                                                segs[2].ctypes.data_as(_u8p), lens[2]))

    def ct_serialize(self, s, first=0, n=None):
        """bytes of ciphertexts [first, first+n) of `s` in the current format, concatenated"""
        n = s.count - first if n is None else n
        each = self.lib.lumen_ct_serialized_size(self.h, s.nl)
        out = np.zeros(max(each * n, 1), dtype=np.uint8)
        self._ck(self.lib.lumen_ct_serialize(self.h, s.h, first, n, out.ctypes.data_as(_u8p), each * n))
        return out[:each * n].tobytes()

    def ct_serialized_size(self, nl):
        return int(self.lib.lumen_ct_serialized_size(self.h, nl))

    def ct_serialize_into(self, s, out, offset=0, first=0, n=None, wait=True):
        """wire bytes of ciphertexts [first, first+n) of `s` into the uint8 array `out` at `offset`; wait=False
        needs a page-locked array (pinned_bytes) and returns once the work is enqueued on this context's stream"""
        n = s.count - first if n is None else n
        size = self.ct_serialized_size(s.nl) * n
        assert out.dtype == np.uint8 and out.flags["C_CONTIGUOUS"] and offset + size <= out.size
        fn = self.lib.lumen_ct_serialize if wait else self.lib.lumen_ct_serialize_async
        self._ck(fn(self.h, s.h, first, n, C.cast(out.ctypes.data + offset, _u8p), size))
        return size

    def ct_deserialize(self, blob, n, nl):
        """n serialised ciphertexts of nl limbs (bytes, or a uint8 array -- page-locked for one DMA) -> a set"""
        a = np.frombuffer(blob, dtype=np.uint8) if isinstance(blob, (bytes, bytearray, memoryview)) else blob
        assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
        h = C.c_void_p()
        self._ck(self.lib.lumen_ct_deserialize(self.h, C.cast(a.ctypes.data, _u8p), a.size, n, nl, C.byref(h)))
        return DeviceSet(self, h)

    def leaf_digests(self, s):
        out = np.zeros((s.count, 32), dtype=np.uint8)
        self._ck(self.lib.lumen_leaf_digests(self.h, s.h, out.ctypes.data_as(_u8p)))
        return out

    def load_public_key(self, pk):
        pk = np.ascontiguousarray(pk, dtype=np.uint64)
        assert pk.shape == (2, self.L + self.K, self.N), pk.shape  # over QP, as rlwe.PublicKey holds it
        self._ck(self.lib.lumen_load_public_key(self.h, _p64(pk)))

    def encrypt_pk(self, plaintexts, count, seed, first_index=0):
        """count ciphertexts of `plaintexts` ([count][L][N], or None: zeros) under the loaded public key."""
        seed = np.ascontiguousarray(seed, dtype=np.uint8)
        assert seed.size == 32
        if plaintexts is not None:
            plaintexts = np.ascontiguousarray(plaintexts, dtype=np.uint64)
            assert plaintexts.shape == (count, self.L, self.N), plaintexts.shape
        h = C.c_void_p()
        self._ck(self.lib.lumen_encrypt_pk(self.h, _p64(plaintexts) if plaintexts is not None else None, count,
                                           seed.ctypes.data_as(_u8p), first_index, C.byref(h)))
        return DeviceSet(self, h)

    def encoder_set(self, psi_t):
        self._ck(self.lib.lumen_encoder_set(self.h, psi_t))

    def encrypt_values(self, values, seed, first_index=0):
        """Encoder.Encode + EncryptNew of every row of `values` ([count][rows] slot values)."""
        seed = np.ascontiguousarray(seed, dtype=np.uint8)
        values = np.ascontiguousarray(values, dtype=np.uint64)
        assert seed.size == 32 and values.ndim == 2
        h = C.c_void_p()
        self._ck(self.lib.lumen_encrypt_values(self.h, _p64(values), values.shape[1], values.shape[0],
                                               seed.ctypes.data_as(_u8p), first_index, C.byref(h)))
        return DeviceSet(self, h)

    def load_secret_key(self, sk):
        sk = np.ascontiguousarray(sk, dtype=np.uint64)[:self.L]
        assert sk.shape == (self.L, self.N), sk.shape
        self._ck(self.lib.lumen_load_secret_key(self.h, _p64(np.ascontiguousarray(sk))))

    def decrypt(self, s, nvalues, scale=1):
        """Slot values of every ciphertext of `s` (one or two limbs): [count][nvalues]."""
        out = np.zeros((s.count, nvalues), dtype=np.uint64)
        self._ck(self.lib.lumen_decrypt(self.h, s.h, scale, nvalues, _p64(out)))
        return out

    def leaf_digests_begin(self, s):
        """Start hashing the leaves of `s` on the context's side stream (overlaps later calls)."""
        self._pending_leaves = s.count
        if s.count:
            self._ck(self.lib.lumen_leaf_digests_begin(self.h, s.h))

    def leaf_digests_end(self):
        out = np.zeros((self._pending_leaves, 32), dtype=np.uint8)
        if self._pending_leaves:
            self._ck(self.lib.lumen_leaf_digests_end(self.h, out.ctypes.data_as(_u8p)))
        self._pending_leaves = 0
        return out

    def merkle_build(self, digests):
        digests = np.ascontiguousarray(digests, dtype=np.uint8).reshape(-1, 32)
        n = digests.shape[0]
        nodes = np.zeros((2 * n + 64, 32), dtype=np.uint8)
        root = np.zeros(32, dtype=np.uint8)
        cnt = C.c_size_t()
        self._ck(self.lib.lumen_merkle_build(self.h, digests.ctypes.data_as(_u8p), n,
                                             nodes.ctypes.data_as(_u8p), nodes.shape[0], C.byref(cnt),
                                             root.ctypes.data_as(_u8p)))
        return nodes[:cnt.value].copy(), root.tobytes()

    def load_galois_key(self, gal_el, evk, montgomery=False):
        """evk: [beta][2][L+K][N]; montgomery=True: the words are x * 2^64 mod q_i (Lattigo's own storage form)"""
        evk = np.ascontiguousarray(evk, dtype=np.uint64)
        assert evk.shape == ((self.L + self.K - 1) // max(self.K, 1), 2, self.L + self.K, self.N), evk.shape
        self._ck(self.lib.lumen_load_galois_key_ex(self.h, gal_el, _p64(evk), 1 if montgomery else 0))

    def inner_sum_galois_elements(self, n):
        g = np.zeros(64, dtype=np.uint64)
        cnt = self.lib.lumen_inner_sum_galois_elements(self.h, n, _p64(g))
        return [int(x) for x in g[:cnt]]

    def mul_plain(self, s, pt):
        pt = np.ascontiguousarray(pt, dtype=np.uint64)
        h = C.c_void_p()
        self._ck(self.lib.lumen_mul_plain(self.h, s.h, _p64(pt), C.byref(h)))
        return DeviceSet(self, h)

    def inner_sum(self, s, n):
        h = C.c_void_p()
        self._ck(self.lib.lumen_inner_sum(self.h, s.h, n, C.byref(h)))
        return DeviceSet(self, h)

    def matrix_inner_sum(self, matrix, pt, rows):
        pt = np.ascontiguousarray(pt, dtype=np.uint64)
        h = C.c_void_p()
        self._ck(self.lib.lumen_matrix_inner_sum(self.h, matrix.h, _p64(pt), rows, C.byref(h)))
        return DeviceSet(self, h)

    def plain_inner_products(self, s, vec):
        """out[j] = sum_i s[j][i] * vec[i] mod q_0 for a one-limb set (the plain prover, ligero.go:886-918)"""
        vec = np.ascontiguousarray(vec, dtype=np.uint64)
        assert vec.size == 2 * self.N
        out = np.zeros(s.count, dtype=np.uint64)
        self._ck(self.lib.lumen_plain_inner_products(self.h, s.h, _p64(vec), _p64(out)))
        return out

    def gather(self, s, idx):
        idx = np.ascontiguousarray(idx, dtype=np.uint32)
        h = C.c_void_p()
        self._ck(self.lib.lumen_gather(self.h, s.h, idx.ctypes.data_as(_u32p), len(idx), C.byref(h)))
        return DeviceSet(self, h)

    def ringswitch_key_shape(self, w=13):
        """(rns, pw2, 2, L+K, N): rlwe.GadgetCiphertext.Value of the ring-switch key, flattened"""
        return (int(self.lib.lumen_ringswitch_rns_digits(self.h)), int(self.lib.lumen_ringswitch_digits(self.h, w)),
                2, self.L + self.K, self.N)

    def load_ringswitch_key(self, log_n_small, key, w=13):
        """key: the whole evaluation key [rns][pw2][2][L+K][N], or just its RNS digit 0 [pw2][2][L+K][N]"""
        key = np.ascontiguousarray(key, dtype=np.uint64)
        shape = self.ringswitch_key_shape(w)
        if key.shape not in (shape, shape[1:]):
            raise ValueError(f"ring-switch key has shape {key.shape}, expected {shape} or {shape[1:]}")
        self._ck(self.lib.lumen_load_ringswitch_key(self.h, log_n_small, w, _p64(key), key.size))
        self._rs_logn = log_n_small

    def ring_switch(self, s, out=None):
        """RingSwitchNew of every ciphertext of `s`: [count][2][n] residues (into `out` if given, e.g. page-locked)"""
        if out is None:
            out = np.zeros((s.count, 2, 1 << self._rs_logn), dtype=np.uint64)
        assert out.shape == (s.count, 2, 1 << self._rs_logn) and out.dtype == np.uint64 and out.flags["C_CONTIGUOUS"]
        self._ck(self.lib.lumen_ring_switch(self.h, s.h, _p64(out)))
        return out

    def mul_counter(self):
        return int(self.lib.lumen_mul_counter(self.h))

    # timing / profiling
    def timer_start(self):
        self._ck(self.lib.lumen_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_float()
        self._ck(self.lib.lumen_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def prof_enable(self, on=True):
        self._ck(self.lib.lumen_prof_enable(self.h, 1 if on else 0))

    def prof_reset(self):
        self._ck(self.lib.lumen_prof_reset(self.h))

    def prof_names(self):
        buf = C.create_string_buffer(4096)
        self.lib.lumen_prof_names(self.h, buf, 4096)
        return [x for x in buf.value.decode().split(",") if x]

    def prof_read(self, kernel):
        ms, n, u = C.c_double(), C.c_uint64(), C.c_uint64()
        self._ck(self.lib.lumen_prof_read(self.h, kernel.encode(), C.byref(ms), C.byref(n), C.byref(u)))
        return ms.value, n.value, u.value

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


TRANSPORT = {"auto": 0, "copy": 1, "rccl": 2}


class Group:
    """lumen_group: W = 2^k ranks, one Context per GPU, with the exchange steps of the sharded Commit inside the
    library (include/lumenos_hip.h).  Group(ctxs) owns all ranks in this process (a Go server's topology);
    Group.join(ctx, rank, world, uid) is one process per GPU (torch.distributed.run).  Every list argument has
    one entry per LOCAL context."""

    def __init__(self, ctxs, transport="auto", _handle=None, _world=None):
        self.lib = load()
        self.ctxs = list(ctxs)
        if _handle is not None:
            self.h, self.world = _handle, _world
            return
        W = len(self.ctxs)
        assert W >= 1 and W & (W - 1) == 0, "a group has a power-of-two number of ranks"
        arr = (C.c_void_p * W)(*[c.h for c in self.ctxs])
        h = C.c_void_p()
        if self.lib.lumen_group_create(arr, W.bit_length() - 1, TRANSPORT[transport], C.byref(h)):
            raise LumenError(self.lib.lumen_last_error(None).decode())
        self.h, self.world = h, W

    @staticmethod
    def unique_id():
        lib = load()
        uid = np.zeros(128, dtype=np.uint8)
        if lib.lumen_group_unique_id(uid.ctypes.data_as(_u8p)):
            raise LumenError(lib.lumen_last_error(None).decode())
        return uid

    @classmethod
    def join(cls, ctx, rank, world, uid):
        lib = load()
        uid = np.ascontiguousarray(uid, dtype=np.uint8)
        assert uid.size == 128 and world & (world - 1) == 0
        h = C.c_void_p()
        if lib.lumen_group_create_rank(ctx.h, rank, world.bit_length() - 1, uid.ctypes.data_as(_u8p), C.byref(h)):
            raise LumenError(lib.lumen_last_error(None).decode())
        return cls([ctx], _handle=h, _world=world)

    def _ck(self, rc):
        if rc:
            raise LumenError(self.lib.lumen_last_error(None).decode())

    @property
    def transport(self):
        return self.lib.lumen_group_transport(self.h).decode()

    @property
    def transport_note(self):
        return self.lib.lumen_group_transport_note(self.h).decode()

    @property
    def rccl_ranks(self):
        return int(self.lib.lumen_group_rccl_ranks(self.h))

    def _handles(self, sets):
        assert len(sets) == len(self.ctxs)
        return (C.c_void_p * len(sets))(*[s.h for s in sets])

    def sync(self):
        self._ck(self.lib.lumen_group_sync(self.h))

    def upload(self, sets, hosts):
        """every local rank's set from its host array, all DMAs in flight together"""
        ptrs = (C.c_void_p * len(sets))(*[h.ctypes.data for h in hosts])
        assert all(h.dtype == np.uint64 and h.flags["C_CONTIGUOUS"] and h.nbytes == s.nbytes for h, s in zip(hosts, sets))
        self._ck(self.lib.lumen_group_upload(self.h, self._handles(sets), ptrs))

    def download(self, sets, hosts):
        ptrs = (C.c_void_p * len(sets))(*[h.ctypes.data for h in hosts])
        assert all(h.dtype == np.uint64 and h.flags["C_CONTIGUOUS"] and h.nbytes == s.nbytes for h, s in zip(hosts, sets))
        self._ck(self.lib.lumen_group_download(self.h, self._handles(sets), ptrs))

    def all_to_all(self, send, recv):
        self._ck(self.lib.lumen_group_all_to_all(self.h, self._handles(send), self._handles(recv)))

    def encode(self, matrix, zero_ct, rho_inv):
        """matrix[i]: local rank i's block of full-width input columns -> its block of encoded columns"""
        zero_ct = np.ascontiguousarray(zero_ct, dtype=np.uint64)
        out = (C.c_void_p * len(self.ctxs))()
        self._ck(self.lib.lumen_group_encode(self.h, self._handles(matrix), _p64(zero_ct), rho_inv, out))
        return [DeviceSet(c, C.c_void_p(out[i])) for i, c in enumerate(self.ctxs)]

    def all_gather_digests(self):
        """ends every local context's leaf_digests_begin job and all-gathers the digests on the devices"""
        self._ck(self.lib.lumen_group_all_gather_digests(self.h))
        for c in self.ctxs:
            c._pending_leaves = 0

    def merkle_root(self):
        root = np.zeros(32, dtype=np.uint8)
        self._ck(self.lib.lumen_group_merkle_root(self.h, root.ctypes.data_as(_u8p)))
        return root.tobytes()

    def digests(self, n_leaves):
        out = np.zeros((n_leaves, 32), dtype=np.uint8)
        n = C.c_uint32()
        self._ck(self.lib.lumen_group_digests(self.h, out.ctypes.data_as(_u8p), out.size, C.byref(n)))
        assert n.value == n_leaves, (n.value, n_leaves)
        return out

    def gather(self, src, idx):
        """queried columns (global indices) collected on rank 0 in query order; None where rank 0 is not local"""
        idx = np.ascontiguousarray(idx, dtype=np.uint32)
        h = C.c_void_p()
        self._ck(self.lib.lumen_group_gather(self.h, self._handles(src), idx.ctypes.data_as(_u32p), len(idx), C.byref(h)))
        return DeviceSet(self.ctxs[0], h) if h.value else None

    def stats(self, name):
        ms, b, n = C.c_double(), C.c_uint64(), C.c_uint64()
        self._ck(self.lib.lumen_group_stats(self.h, name.encode(), C.byref(ms), C.byref(b), C.byref(n)))
        return ms.value, b.value, n.value

    def stats_reset(self):
        self._ck(self.lib.lumen_group_stats_reset(self.h))

    def close(self):
        """destroy the group -- BEFORE its contexts (the C rule: lumen_group_destroy waits on their streams)"""
        if self.h:
            if all(c.h for c in self.ctxs):
                self.lib.lumen_group_destroy(self.h)
            # a context already closed: its streams are gone, the group handle is abandoned rather than touched
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
