"""bench.py's ONE-PROCESS-PER-GPU code path (what `torch.distributed.run bench.py --gpus N` executes on a multi-GPU node)
rehearsed on ONE GPU: the W ranks are W threads of this process, each with its own bench.Job (own context on device 0), a
thread-based stand-in for the torch.distributed control plane (all_gather_object / barrier: the only calls that path
makes on it), and the library's RCCL group joined through bench.join_ranks -- real lumen_group_unique_id /
lumen_group_create_rank over the test double tests/cpp/fake_rccl.cpp, which must be first on LD_LIBRARY_PATH (run by
tests/test_group_rccl.py; torch is never imported, so no other RCCL holds the soname).

Per rank: join_ranks -> Job.step_group (Encode between the two all-to-alls, rescale, leaf digests, both inner products,
query gather to rank 0, digest all-gather, Merkle root) x steps -> multi_rank_report (per-stage timers, collective
statistics, and `check`: the W-rank result against a one-rank recompute, which joins a SECOND communicator while the
first is alive).  Prints one JSON line; exit code 0 only if every rank's check is ok and all ranks agree on the root."""
import argparse
import json
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


class ThreadPlane:
    """all_gather_object / barrier of torch.distributed for W threads of one process"""

    def __init__(self, world):
        self.world = world
        self.lock = threading.Condition()
        self.slots, self.arrived, self.gen = {}, 0, 0

    def handle(self, rank):
        plane = self

        class H:
            def all_gather_object(self, out, obj):
                with plane.lock:
                    gen = plane.gen
                    plane.slots[rank] = obj
                    plane.arrived += 1
                    if plane.arrived == plane.world:
                        plane.result = [plane.slots[r] for r in range(plane.world)]
                        plane.slots, plane.arrived, plane.gen = {}, 0, gen + 1
                        plane.lock.notify_all()
                    else:
                        ok = plane.lock.wait_for(lambda: plane.gen != gen, timeout=300)
                        assert ok, "a rank never reached the control-plane gather"
                    res = plane.result
                out[:] = res

            def barrier(self):
                self.all_gather_object([None] * plane.world, None)

            def get_backend(self, *_):
                return "threads"

        return H()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--config", default="2048x1024")
    ap.add_argument("--steps", type=int, default=2)
    a = ap.parse_args()
    import bench
    W = a.world
    plane = ThreadPlane(W)
    out, errs = [None] * W, [None] * W

    def rank_main(r):
        try:
            dist = plane.handle(r)
            job = bench.Job(a.config, r, W, 0)
            job.ctx.test_allow_shared_device_rccl(True)
            group, flags = bench.join_ranks(job.ctx, r, W, dist, identity=f"thread-rank-{r}")
            assert group is not None, flags
            job.group = group
            roots = [job.step_group() for _ in range(a.steps)]
            args = argparse.Namespace(no_check=False)
            rep = bench.multi_rank_report(job, args, dist, [(None, None, 0)], 1.0)
            out[r] = {"root": roots[-1].hex(), "transport": group.transport, "note": group.transport_note,
                      "report": rep if r == 0 else None, "same_root_every_step": len(set(roots)) == 1}
            dist.barrier()
            job.close()
        except BaseException as e:  # noqa: BLE001
            import traceback
            errs[r] = traceback.format_exc()

    from bench_lib import multi
    real_join = multi.join_ranks

    def join_as_threads(ctx, rank, world, dist, deadline_s=None, identity=None):
        ctx.test_allow_shared_device_rccl(True)
        return real_join(ctx, rank, world, dist, deadline_s, identity or f"thread-rank-{rank}")

    # (multi_rank_report's check job joins through the module's own name: patch it where it lives)
    multi.join_ranks = bench.join_ranks = join_as_threads
    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(W)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(900)
    if any(t.is_alive() for t in ts) or any(errs):
        print(json.dumps({"ok": False, "stuck": [t.is_alive() for t in ts], "errors": errs}))
        os._exit(1)
    rep = out[0]["report"]
    ok = (len({o["root"] for o in out}) == 1 and all(o["same_root_every_step"] for o in out) and
          bool(rep.get("check", {}).get("ok")) and rep.get("rccl_ranks_seen") == W)
    print(json.dumps({"ok": ok, "world": W, "config": a.config, "transport": out[0]["transport"], "note": out[0]["note"],
                      "root": out[0]["root"], "rccl_ranks_seen": rep.get("rccl_ranks_seen"), "check": rep.get("check"),
                      "collectives": rep.get("collectives"), "per_rank_stage_s": rep.get("per_rank_stage_s")}))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
