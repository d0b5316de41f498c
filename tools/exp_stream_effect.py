"""EXPERIMENT: is part of the gadget product's process-to-process spread a property of the STREAM (hardware queue) it runs on rather
than of where its buffers sit?  One process: the same four device blocks, k_ks_mac alone (lumen_ks_mac_probe) from the main context
and from five clones (each has streams of its own), three rounds."""
import json, os, sys
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
ctx.ks_mac_probe(64, reps=20)
blk = {n: ctx.scratch_info(n)[0] for n in ("ks_ext", "ks_u", "ks_acc")}
ctxs = [ctx] + [ctx.clone() for _ in range(5)]
rows = []
for rnd in range(3):
    rows.append([round(c.ks_mac_probe(64, ext=blk["ks_ext"], u=blk["ks_u"], acc=blk["ks_acc"], reps=300), 5) for c in ctxs])
print(json.dumps({"pid": os.getpid(), "ms_per_launch_by_context": rows}), flush=True)
for c in ctxs[:0:-1]:
    c.close()
ctx.close()
