"""EXPERIMENT (round 6): two key-switch lanes on disjoint XCD halves.  Needs the library built from
tools/exp_xcd_halves.patch (lumen_ks_overlap_probe is not part of the product)."""
import ctypes as C
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lumenos_amd import params as lp
from lumenos_amd import hip
from lumenos_amd.hip import Context
P = lp.generate_bgv_params_for_ntt(4096, 14)
ctx = Context(P.log_n, P.q, P.p, P.psi, P.T, device=0)
ctx.set_tuning("LUMEN_KS_PLACEMENT", 0)
rng = np.random.default_rng(1)
beta = (len(P.q) + len(P.p) - 1) // len(P.p)
evk = np.stack([rng.integers(0, m, size=(beta, 2, P.N), dtype=np.uint64) for m in P.q + P.p])
ctx.load_galois_key(5, np.ascontiguousarray(evk.transpose(1, 2, 0, 3)))
fn = ctx.lib.lumen_ks_overlap_probe
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]
names = {7: "CU-masked streams: A on CUs 0-15 of every XCD, B on CUs 16-31, opposite phase", 8: "batch A alone on CUs 0-15 of every XCD", 9: "product of A alone on CUs 0-15 of every XCD", 3: "batch A alone on XCDs 0-3", 4: "batch A alone on the whole chip", 5: "extension of A alone on XCDs 0-3", 6: "product of A alone on XCDs 0-3", 0: "one stream, whole chip, A then B", 1: "two streams, A on XCDs 0-3 / B on XCDs 4-7, opposite phase", 2: "two streams, whole chip (free-running)"}
for rnd in range(3):
    for mode in (0, 4, 7, 8, 9, 1, 2, 3, 5, 6):
        ms = C.c_float()
        ctx._ck(fn(ctx.h, 64, mode, 40, C.byref(ms)))
        print(f"round {rnd} mode {mode} ({names[mode]}): {ms.value:.4f} ms per (extension + product) x 2 batches", flush=True)
ctx.close()
