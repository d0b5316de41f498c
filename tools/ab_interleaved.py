"""Interleaved A/B of run-time tuning switches inside ONE process ("settle the TGROUP / ks_mac
question with an interleaved A/B").  Usage (GPU box, repo root):

    python tools/ab_interleaved.py --switch LUMEN_MODDOWN_TGROUP --values 4 12 6 --rounds 4 --steps 20 > out.txt

The headline job (16384x4096) is built once; then for `--rounds` rounds the values are visited in order
(A B C A B C ...), each visit = `--steps` timed prover steps (wall clock around the steps, device drained before and
after) followed by ONE extra step with HIP events around every launch (lumen_prof_read) for the per-kernel table.
rocm-smi is sampled right before and right after every visit (sclk, mclk, power, junction / memory temperature), so a
drift of the box shows up beside the numbers instead of inside them.  The last lines are per-value means and the
spread between rounds; interleaving means box drift hits all values alike."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def smi():
    from bench_lib.report import smi_sample  # amdgpu's hwmon files; rocm-smi only where those are not readable
    return smi_sample()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--switch", default=None)
    ap.add_argument("--values", type=int, nargs="+", default=None)
    ap.add_argument("--combos", nargs="+", default=None,
                    help="instead of --switch / --values: settings of SEVERAL switches per visit, e.g. LUMEN_KS_BATCH=128,LUMEN_KS_SUB_BATCH=64")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--config", default="16384x4096")
    ap.add_argument("--kernels", nargs="+", default=["ks_moddown_ntt", "ks_modup_ntt", "ks_mac", "ks_intt_c1", "ks_intt_p",
                                                       "ks_pack_v"])
    args = ap.parse_args()
    import bench
    job = bench.Job(args.config, 0, 1, 0)
    ctx = job.ctx
    for _ in range(3):  # warm-up: pools, scratch, work lists, clocks
        job.step(None)
    ctx.sync()
    rows = []
    if args.combos:
        visits = [(c, [kv.split("=") for kv in c.split(",")]) for c in args.combos]
        args.switch = "combo"
    else:
        visits = [(v, [(args.switch, v)]) for v in args.values]
    args.values = [v for v, _ in visits]
    print(f"# {args.switch} in {args.values}, {args.rounds} rounds x {args.steps} steps, config {args.config}", flush=True)
    for rnd in range(args.rounds):
        for v, settings in visits:
            for name, val in settings:
                ctx.set_tuning(name, int(val))
            job.step(None)  # first step under the new setting builds / fetches its cached lists
            ctx.sync()
            before = smi()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                job.step(None)
            ctx.sync()
            sec = (time.perf_counter() - t0) / args.steps
            after = smi()
            ctx.prof_reset()
            ctx.prof_enable(True)
            job.step(None)
            ctx.sync()
            ctx.prof_enable(False)
            tab = {k: ctx.prof_read(k) for k in ctx.prof_names()}
            row = {"round": rnd, "value": v, "s_per_step": round(sec, 5),
                   "kernel_ms": {k: round(tab[k][0], 2) for k in args.kernels if k in tab}, "smi_before": before, "smi_after": after}
            rows.append(row)
            print(json.dumps(row), flush=True)
    print("# summary: value | mean s/step (min .. max over rounds) | mean ms per kernel")
    for v in args.values:
        mine = [r for r in rows if r["value"] == v]
        ss = [r["s_per_step"] for r in mine]
        km = {k: sum(r["kernel_ms"].get(k, 0) for r in mine) / len(mine) for k in args.kernels}
        print(f"# {args.switch}={v!s:<3} {sum(ss) / len(ss):.4f} s ({min(ss):.4f} .. {max(ss):.4f}) | " +
              " ".join(f"{k}={x:.1f}" for k, x in km.items()), flush=True)
    job.close()


if __name__ == "__main__":
    main()
