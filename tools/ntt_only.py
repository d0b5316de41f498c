"""Runs only the limb NTT (forward, inverse) at N=2^14, L=12 for profiling with rocprofv3."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lumenos_amd import params as lp
from lumenos_amd.hip import Context

def main():
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 14
    cols = {14: 4096, 13: 4096, 12: 1024}[log_n]
    P = lp.generate_bgv_params_for_ntt(cols, log_n)
    ctx = Context(P.log_n, P.q, P.p, P.psi, P.T)
    s = ctx.new_set(256, len(P.q)).fill_random(1)
    for _ in range(3):
        ctx.set_ntt(s, False)
        ctx.set_ntt(s, True)
    ctx.sync()
    ctx.close()

if __name__ == "__main__":
    main()
