"""Experiment: Commit (Encode + rescale + leaf hashing) on a clone context, concurrently with the inner products
on the main one, against the sequential step of bench.py.  usage: exp_overlap_commit.py [config] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402


def step_concurrent(job, cc):
    ctx = job.ctx
    mine = cc.encode(job.matrix, job.zero_ct, bench.RHO_INV)
    lvl1 = cc.rescale(mine, 2)
    mine.free()  # (waits for cc's stream: the host blocks here until Encode + rescale are done)
    cc.leaf_digests_begin(lvl1)
    return lvl1


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "16384x4096"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    job = bench.Job(cfg, 0, 1, 0)
    ctx = job.ctx
    cc = ctx.clone()
    import threading

    def seq():
        job.step()

    def conc():
        box = {}
        th = threading.Thread(target=lambda: box.setdefault("lvl1", step_concurrent(job, cc)))
        th.start()
        mat_r = ctx.matrix_inner_sum(job.matrix, job.r_pt, job.rows)
        mat_z = ctx.matrix_inner_sum(job.matrix, job.b_pt, job.rows)
        th.join()
        lvl1 = box["lvl1"]
        q = ctx.gather(lvl1, job.query_idx)
        dig = cc.leaf_digests_end()
        nodes, root = ctx.merkle_build(dig)
        ctx.sync()
        cc.sync()
        for s in (q, mat_r, mat_z):
            s.free()
        lvl1.free()
        return root

    for name, fn in (("sequential", seq), ("concurrent", conc), ("sequential", seq), ("concurrent", conc)):
        fn()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        ctx.sync()
        cc.sync()
        print(f"{cfg} {name}: {(time.perf_counter() - t0) / steps:.4f} s per step", flush=True)


if __name__ == "__main__":
    main()
