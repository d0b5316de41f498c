"""Quick device timing of the building blocks (not the contract bench)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle.loader import Oracle
from tests.helpers import make_params, make_context, T_REF

def main():
    o = Oracle()
    for log_n, L, cnt in ((12, 10, 512), (13, 12, 256), (14, 12, 256)):
        P = make_params(o, log_n, L)
        ctx = make_context(P)
        s = ctx.new_set(cnt, L).fill_random(1)
        ctx.sync()
        for inv in (False, True):
            ctx.set_ntt(s, inv); ctx.sync()
            ctx.timer_start()
            reps = 5
            for _ in range(reps):
                ctx.set_ntt(s, inv)
            ms = ctx.timer_stop() / reps
            limbs = cnt * 2 * L
            gbs = limbs * 16 * P.N / (ms * 1e-3) / 1e9
            print(f"logN={log_n} L={L} cts={cnt} {'INTT' if inv else 'NTT '}: {ms:.3f} ms, {limbs/(ms*1e-3)/1e6:.2f} M limb-NTT/s, {gbs:.0f} GB/s algorithmic")
        ctx.close()

if __name__ == "__main__":
    main()
