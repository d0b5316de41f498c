"""Where in HBM the gadget product's streams sit, and what that costs (`ks_mac` is 341 .. 381 ms
per step depending on the PROCESS, constant inside one).  GPU box, repo root:

    python tools/ks_mac_placement.py [--insitu] [--cands 6] [--reps 200] [--sweep]

Builds the headline context (N = 2^14, L = 12, K = 2; one random Galois key) and times k_ks_mac ALONE
(lumen_ks_mac_probe, HIP events over `reps` launches) on
  * the library's own scratch blocks (what the product runs on in this process), addresses printed;
  * `--cands` freshly allocated candidates for each of ext / u / key, varied one at a time (which stream's
    placement matters, and how much it spreads inside ONE process);
  * --sweep: one slab, the ext / u blocks at a sliding offset from each other (channel / bank aliasing between the
    kernel's streams would show as a periodic pattern).
--insitu also runs whole prover steps (bench.Job) and reports ks_mac's HIP-event time per step beside the probe.
One JSON object per line; run it in several processes (a shell loop) to see the process-to-process spread."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--insitu", action="store_true")
    ap.add_argument("--cands", type=int, default=6)
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--tag", default="")
    ap.add_argument("--select", type=int, default=-1, help="LUMEN_KS_PLACEMENT for the library's own blocks (-1: the default)")
    args = ap.parse_args()
    import bench
    from lumenos_amd import params as lp
    from lumenos_amd.hip import Context
    out = {"tag": args.tag, "pid": os.getpid(), "lib": os.environ.get("LUMEN_HIP_LIB", "product")}
    B = args.batch
    if args.insitu:
        job = bench.Job("16384x4096", 0, 1, 0)
        ctx = job.ctx
        for _ in range(2):
            job.step(None)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(3):
            job.step(None)
        ctx.sync()
        out["s_per_step"] = round((time.perf_counter() - t0) / 3, 4)
        ctx.prof_reset()
        ctx.prof_enable(True)
        job.step(None)
        ctx.sync()
        ctx.prof_enable(False)
        tab = {k: ctx.prof_read(k) for k in ctx.prof_names()}
        out["insitu_ms_per_step"] = {k: round(tab[k][0], 2) for k in ("ks_mac", "ks_modup_ntt", "ks_moddown_ntt", "ks_intt_c1") if k in tab}
        out["insitu_ks_mac_ms_per_launch"] = round(tab["ks_mac"][0] / tab["ks_mac"][1], 5)
    else:
        P = lp.generate_bgv_params_for_ntt(4096, 14)
        ctx = Context(P.log_n, P.q, P.p, P.psi, P.T, device=0)
        rng = np.random.default_rng(1)
        beta = (len(P.q) + len(P.p) - 1) // len(P.p)
        evk = np.stack([rng.integers(0, m, size=(beta, 2, P.N), dtype=np.uint64) for m in P.q + P.p])
        ctx.load_galois_key(5, np.ascontiguousarray(evk.transpose(1, 2, 0, 3)))
    L, K, N = ctx.L, ctx.K, ctx.N
    LK, beta = L + K, (L + K - 1) // K
    if args.select >= 0:  # re-draw the library's blocks with this many candidates per buffer (0: plain hipMalloc)
        ctx.set_tuning("LUMEN_KS_PLACEMENT", args.select)
        ctx.trim()
    base = ctx.ks_mac_probe(B, reps=args.reps)  # allocates the library's blocks if they are not there yet
    out["scratch"] = {n: [hex(a or 0), s] for n in ("ks_ext", "ks_u", "ks_coef", "ks_acc2", "ks_acc")
                      for a, s in [ctx.scratch_info(n)]}
    out["probe_product_blocks_ms"] = [round(base, 5)] + [round(ctx.ks_mac_probe(B, reps=args.reps), 5) for _ in range(2)]

    def block(limbs):  # a fresh device block of at least `limbs` limbs: a set of whole ciphertexts
        s = ctx.new_set((limbs + 2 * L - 1) // (2 * L), L)
        return s

    sizes = {"ext": B * beta * LK, "u": B * 2 * LK, "key": beta * 2 * LK}
    keep = []
    for which in ("ext", "u", "key"):
        rows = []
        for c in range(args.cands):
            s = block(sizes[which])
            keep.append(s)
            if which == "key":
                s.fill_random(3)  # (any words do: the kernel's control flow is data-independent)
            ms = ctx.ks_mac_probe(B, reps=args.reps, **{which: s.device_ptr})
            rows.append([hex(s.device_ptr), round(ms, 5)])
        out["vary_" + which] = rows
    if args.sweep:
        slab = block(sizes["ext"] + sizes["u"] + 1024)  # ext at the bottom, u sliding above it
        keep.append(slab)
        p0 = slab.device_ptr
        rows = []
        for off_kb in [0, 4, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 512, 768, 1024, 1536, 2048, 4096, 8192, 16384,
                       32768, 65536]:  # (the slab has 1024 limbs = 128 MB of slack)
            off = off_kb * 1024
            ms = ctx.ks_mac_probe(B, reps=args.reps, ext=p0, u=p0 + sizes["ext"] * N * 8 + off)
            rows.append([off_kb, round(ms, 5)])
        out["sweep_u_offset_kb"] = rows
    out["probe_product_blocks_again_ms"] = round(ctx.ks_mac_probe(B, reps=args.reps), 5)
    print(json.dumps(out), flush=True)
    for s in keep:
        s.free()
    if args.insitu:
        job.close()
    else:
        ctx.close()


if __name__ == "__main__":
    main()
