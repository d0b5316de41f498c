"""Generates the golden fixtures under tests/golden/ from the CPU oracle.

The reference (Go + un-vendored Lattigo) cannot run in this environment, so
these vectors pin the build's OWN oracle (regression vectors for the oracle and
the size-independent inputs the GPU tests replay); the reference-derived known
answers live in test_oracle_kat.py.  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from helpers import T_REF, make_params, random_cts  # noqa: E402
from oracle.loader import Oracle  # noqa: E402


def main():
    o = Oracle()
    P = make_params(o, 8, 3)
    P.seed(2024)
    meta = dict(log_n=8, q=np.array(P.moduli[:3], dtype=np.uint64), p=np.array(P.moduli[3:], dtype=np.uint64),
                psi=np.array(P.psi, dtype=np.uint64), T=np.uint64(T_REF))
    # fhe.Encode for S = 16, 32, 64
    for S in (16, 32, 64):
        cols = S // 2
        roots = o.field_roots(T_REF, S)
        m = random_cts(P, cols, 3, seed=S)
        z = random_cts(P, 1, 3, seed=S + 1)[0]
        out = P.ct_encode(m, 2, z, roots)
        np.savez_compressed(os.path.join(HERE, f"encode_S{S}.npz"), matrix=m, zero=z, roots=roots, encoded=out, **meta)
    # Rescale to level 1 + leaf digest + ct x pt + InnerSum on real keys
    sk = P.keygen_secret()
    rows = 128
    gl = P.inner_sum_galois_elements(rows)
    evks = np.stack([P.keygen_galois(sk, g) for g in gl])
    cts = random_cts(P, 4, 3, seed=7)
    pt = P.encode(np.arange(1, rows + 1, dtype=np.uint64))
    lvl1, dig = P.commit_leaves(cts)
    mis = P.matrix_inner_sum(cts, pt, rows, list(evks))
    np.savez_compressed(os.path.join(HERE, "evaluator.npz"), cts=cts, pt=pt, rows=np.uint32(rows),
                        gal_els=np.array(gl, dtype=np.uint64), evks=evks, level1=lvl1, digests=dig,
                        matrix_inner_sum=mis, **meta)
    encrypt_fixture(o)
    print("golden fixtures written to", HERE)


def encrypt_fixture(o):
    """Deterministic pk encryption (oracle/lo_encdet.c): pins the sampler (ChaCha20 word layout, CDT)
    and the ciphertext assembly for the CPU suite and the GPU replay."""
    P = make_params(o, 8, 2)
    P.seed(77)
    sk = P.keygen_secret()
    pk = P.keygen_public(sk)
    seed = np.frombuffer(bytes((7 * i + 3) % 256 for i in range(32)), dtype=np.uint8)
    vals = np.random.default_rng(5).integers(0, T_REF, size=(3, P.N), dtype=np.uint64)
    pts = np.stack([P.encode(v) for v in vals])
    first = 2**32 + 9
    cts = np.stack([P.encrypt_det(pk, pts[i], seed, first + i) for i in range(3)])
    small = np.stack([np.stack([P.det_small(seed, first + i, s) for s in range(3)]) for i in range(3)])
    np.savez_compressed(os.path.join(HERE, "encrypt_det.npz"), log_n=8, q=np.array(P.moduli[:2], dtype=np.uint64),
                        p=np.array(P.moduli[2:], dtype=np.uint64), psi=np.array(P.psi, dtype=np.uint64),
                        T=np.uint64(T_REF), sk=sk, pk=pk, seed=seed, first=np.uint64(first), values=vals,
                        plaintexts=pts, small=small, ciphertexts=cts)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "encrypt":  # add this fixture without rewriting the others
        encrypt_fixture(Oracle())
    else:
        main()
