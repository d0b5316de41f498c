"""Runs only the limb NTT (forward, inverse) for profiling with rocprofv3, and prints its rate.

usage: ntt_only.py [log_n] [cts] [reps]
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lumenos_amd import params as lp
from lumenos_amd.hip import Context


def main():
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 14
    cts = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    cols = {14: 4096, 13: 4096, 12: 1024}.get(log_n, 1024)
    P = lp.generate_bgv_params_for_ntt(cols, log_n)
    ctx = Context(P.log_n, P.q, P.p, P.psi, P.T)
    s = ctx.new_set(cts, len(P.q)).fill_random(1)
    ctx.set_ntt(s, False)
    ctx.set_ntt(s, True)
    ctx.sync()
    n = cts * 2 * len(P.q)
    for inv in (False, True):
        ctx.timer_start()
        for _ in range(reps):
            ctx.set_ntt(s, inv)
        ms = ctx.timer_stop()
        print(f"logN={log_n} {'inverse' if inv else 'forward'}: {n * reps / ms / 1e3:.3f} M limb-NTT/s "
              f"({ms / reps:.3f} ms per {n} transforms)")
    ctx.close()


if __name__ == "__main__":
    main()
