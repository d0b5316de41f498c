"""Times the GPU witness encryption (lumen_encrypt_pk) at a bench shape.

usage: encrypt_only.py [config] [count]
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
    count = int(sys.argv[2]) if len(sys.argv) > 2 else cols
    P = lp.generate_bgv_params_for_ntt(cols, log_n)
    ctx = Context(P.log_n, P.q, P.p, P.psi, P.T)
    L = len(P.q)
    rng = np.random.default_rng(1)
    pk = np.stack([np.stack([rng.integers(0, q, size=P.N, dtype=np.uint64) for q in P.q + P.p]) for _ in range(2)])  # over QP
    ctx.load_public_key(pk)
    seed = np.arange(32, dtype=np.uint8)
    ctx.encrypt_pk(None, 64, seed, 0).free()
    ctx.sync()
    ctx.timer_start()
    s = ctx.encrypt_pk(None, count, seed, 0)
    ms = ctx.timer_stop()
    print(f"{cfg}: {count} encryptions of zero in {ms:.1f} ms ({count * L * 3 / ms / 1e3:.2f} M limb-NTT/s, "
          f"{count / ms * 1e3:.0f} ciphertexts/s)")
    s.free()
    n = min(count, 256)
    pts = np.stack([np.stack([rng.integers(0, q, size=P.N, dtype=np.uint64) for q in P.q]) for _ in range(n)])
    t0 = time.perf_counter()
    s = ctx.encrypt_pk(pts, n, seed, 0)
    ctx.sync()
    dt = time.perf_counter() - t0
    print(f"{cfg}: {n} encryptions of host plaintexts ({pts.nbytes / 1e6:.0f} MB over PCIe) in {dt * 1e3:.1f} ms")
    s.free()
    # the whole input side from the raw witness: Encoder.Encode + EncryptNew on the device
    ctx.encoder_set(lp.encoder_psi(P.T, P.log_n))
    rows = P.N
    vals = rng.integers(0, P.T, size=(count, rows), dtype=np.uint64)
    ctx.encrypt_values(vals[:8], seed, 0).free()
    t0 = time.perf_counter()
    s = ctx.encrypt_values(vals, seed, 0)
    ctx.sync()
    dt = time.perf_counter() - t0
    print(f"{cfg}: {count} columns of {rows} witness values ({vals.nbytes / 1e6:.0f} MB over PCIe) encoded + encrypted "
          f"in {dt * 1e3:.1f} ms")
    s.free()
    ctx.close()


if __name__ == "__main__":
    main()
