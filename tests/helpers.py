"""Shared test helpers: one parameter set drives both the CPU oracle and the HIP context."""
import ctypes as C

import numpy as np

from oracle.loader import Params

T_REF = 144115188075593729  # cmd/server/main.go:22, fhe/ligero_test.go:16


def gen_primes(oracle, bits, n, two_n, exclude=(T_REF,)):
    out = np.zeros(n, dtype=np.uint64)
    ex = np.array(list(exclude), dtype=np.uint64)
    rc = oracle.lib.lo_gen_primes(bits, two_n, n, ex.ctypes.data_as(C.POINTER(C.c_uint64)), len(ex),
                                  out.ctypes.data_as(C.POINTER(C.c_uint64)))
    assert rc == 0
    return [int(x) for x in out]


def make_params(oracle, log_n, num_q, num_p=2, T=T_REF):
    """Custom-depth parameter set in the reference's style: LogQ = [58, 56, ...], LogP = [55, 55]."""
    two_n = 2 << log_n
    q = gen_primes(oracle, 58, 1, two_n) + (gen_primes(oracle, 56, num_q - 1, two_n) if num_q > 1 else [])
    p = gen_primes(oracle, 55, num_p, two_n) if num_p else []
    return Params.from_moduli(oracle, log_n, q, p, T)


def make_context(P, device=0):
    from lumenos_amd.hip import Context
    return Context(P.logN, P.moduli[:P.L], P.moduli[P.L:], P.psi, P.T, device=device)


def random_cts(P, count, nl, seed):
    """Uniform residues in [0, q_i): kernels are data-independent."""
    rng = np.random.default_rng(seed)
    out = np.empty((count, 2, nl, P.N), dtype=np.uint64)
    for l in range(nl):
        out[:, :, l, :] = rng.integers(0, P.moduli[l], size=(count, 2, P.N), dtype=np.uint64)
    return out
