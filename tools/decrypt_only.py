"""Times the GPU client-side decryption (lumen_decrypt) of MatR + MatZ at a bench shape.

usage: decrypt_only.py [config]
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lumenos_amd import params as lp
from lumenos_amd.hip import Context

CONFIGS = {"2048x1024": (1024, 12), "4096x2048": (2048, 12), "8192x4096": (4096, 13), "16384x4096": (4096, 14)}


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "16384x4096"
    cols, log_n = CONFIGS[cfg]
    P = lp.generate_bgv_params_for_ntt(cols, log_n)
    ctx = Context(P.log_n, P.q, P.p, P.psi, P.T)
    ctx.encoder_set(lp.encoder_psi(P.T, P.log_n))
    rng = np.random.default_rng(1)
    sk = np.stack([rng.integers(0, q, size=P.N, dtype=np.uint64) for q in P.q])
    ctx.load_secret_key(sk)
    s = ctx.new_set(2 * cols, 2).fill_random(3)  # MatR and MatZ: 2 * cols ciphertexts at level 1
    ctx.decrypt(s, 1)
    t0 = time.perf_counter()
    v = ctx.decrypt(s, 1)
    dt = time.perf_counter() - t0
    print(f"{cfg}: {2 * cols} level-1 ciphertexts decrypted + decoded (slot 0) in {dt * 1e3:.1f} ms")
    ctx.close()


if __name__ == "__main__":
    main()
