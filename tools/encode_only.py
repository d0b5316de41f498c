"""Runs only fhe.Encode (the ciphertext-axis transform) at a bench shape, for profiling.

usage: encode_only.py [config] [reps]      config as in bench.py (default 16384x4096)
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lumenos_amd import params as lp
from lumenos_amd.hip import Context

CONFIGS = {"2048x1024": (1024, 12), "4096x2048": (2048, 12), "8192x4096": (4096, 13), "16384x4096": (4096, 14)}


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "16384x4096"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    cols, log_n = CONFIGS[cfg]
    P = lp.generate_bgv_params_for_ntt(cols, log_n)
    ctx = Context(P.log_n, P.q, P.p, P.psi, P.T)
    S = cols * 2
    ctx.field_set(np.array(lp.field_roots_forward(P.T, S), dtype=np.uint64))
    L = len(P.q)
    m = ctx.new_set(cols, L).fill_random(1)
    rng = np.random.default_rng(1)
    zero = np.stack([rng.integers(0, q, size=(2, P.N), dtype=np.uint64) for q in P.q], axis=1)
    zero = np.ascontiguousarray(zero)
    e = ctx.encode(m, zero, 2)
    e.free()
    ctx.sync()
    ctx.timer_start()
    for _ in range(reps):
        ctx.encode(m, zero, 2).free()
    ms = ctx.timer_stop() / reps
    gb = (cols + S) * 2 * L * P.N * 8 / 1e9
    print(f"Encode {cfg}: {ms:.2f} ms per call ({cols} -> {S} ciphertexts, {gb:.1f} GB in+out)")
    for world in (2, 8):  # one rank's share of a multi-GPU Encode (non-final passes replicated)
        ctx.encode_shard(m, zero, 2, 0, world)[0].free()
        ctx.sync()
        ctx.timer_start()
        for _ in range(reps):
            ctx.encode_shard(m, zero, 2, world - 1, world)[0].free()
        print(f"Encode shard {world - 1}/{world}: {ctx.timer_stop() / reps:.2f} ms per call")
    ctx.close()


if __name__ == "__main__":
    main()
