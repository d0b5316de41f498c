"""EXPERIMENT: does holding spacer blocks between the rounds of candidate draws (so that the candidates come from regions of HBM far
apart) let the placement selection leave the slow mode of the gadget product?  One process: the headline job, then the scratch is
re-drawn alternately without / with spacers and two steps + one profiled step run on each draw.  Needs tools/exp_spacer.patch."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
job = bench.Job("16384x4096", 0, 1, 0)
ctx = job.ctx
ctx.set_tuning("LUMEN_DEBUG", 1)
for rnd in range(2):
    for sp in (0, 8192, 32768):
        ctx.set_tuning("LUMEN_KS_PLACEMENT_SPACER", sp)
        ctx.trim()
        job.step(None); job.step(None)
        ctx.prof_reset(); ctx.prof_enable(True); job.step(None); ctx.sync(); ctx.prof_enable(False)
        tab = {k: ctx.prof_read(k) for k in ctx.prof_names()}
        print(json.dumps({"pid": os.getpid(), "round": rnd, "spacer_mb": sp, "ks_mac": round(tab["ks_mac"][0], 1), "ks_modup_ntt": round(tab["ks_modup_ntt"][0], 1),
                          "ks_moddown_ntt": round(tab["ks_moddown_ntt"][0], 1)}), flush=True)
job.close()
