import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
from oracle.loader import Oracle, Params
from helpers import make_params, make_context, random_cts, T_REF
from lumenos_amd import params as lp

o = Oracle()
def diff(a, b, tag):
    if np.array_equal(a, b):
        print(tag, "OK"); return True
    bad = (a != b)
    print(tag, "MISMATCH total", int(bad.sum()), "of", bad.size)
    for c in range(a.shape[0]):
        for k in range(2):
            for l in range(a.shape[2]):
                n = int(bad[c, k, l].sum())
                if n: print("   ct", c, "poly", k, "limb", l, "bad", n, "first idx", int(np.argmax(bad[c,k,l])))
    return False

for (log_n, L) in ((12, 10), (13, 12), (14, 12), (14, 5), (14, 7)):
    B = lp.generate_bgv_params_for_ntt(1 << L, log_n) if False else None
    P = make_params(o, log_n, L)
    P.seed(1)
    ctx = make_context(P)
    sk = P.keygen_secret()
    cts = random_cts(P, 2, L, seed=3)
    pt = np.stack([np.random.default_rng(5).integers(0, P.moduli[l], size=P.N, dtype=np.uint64) for l in range(L)])
    s = ctx.upload(cts)
    diff(ctx.mul_plain(s, pt).download(), np.stack([P.mul_plain(c, pt) for c in cts]), f"logN={log_n} L={L} mul_plain")
    for n in (2, P.N):
        gl = P.inner_sum_galois_elements(n)
        evks = [P.keygen_galois(sk, g) for g in gl]
        for g, e in zip(gl, evks): ctx.load_galois_key(g, e)
        got = ctx.inner_sum(s, n).download()
        ref = np.stack([P.inner_sum(c, n, evks) for c in cts])
        diff(got, ref, f"logN={log_n} L={L} inner_sum n={n}")
    ctx.close()
