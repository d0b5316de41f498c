"""EXPERIMENT: the extension kernel and the gadget product alone (lumen_ks_overlap_probe of tools/exp_xcd_halves.patch, modes 10 / 11 added for
this sweep) as a function of the batch size: us per column.  Where does the extension kernel's cost per column jump between 64 and 128?"""
import ctypes as C
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lumenos_amd import params as lp
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
for B in (128, 112, 96, 80, 72, 64, 56, 48, 32, 16):
    row = []
    for mode in (10, 11):
        ms = C.c_float()
        ctx._ck(fn(ctx.h, B, mode, 30, C.byref(ms)))
        row.append(ms.value)
    print(f"B={B:4d}: extension {row[0] * 1e3:8.1f} us = {row[0] * 1e3 / B:6.3f} us per column | product {row[1] * 1e3:8.1f} us = {row[1] * 1e3 / B:6.3f} us per column", flush=True)
ctx.close()
