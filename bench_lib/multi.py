"""N > 1: the exchange helpers of the torch.distributed path, joining the library's own RCCL communicator, the
self-diagnosing part of the bench line (collectives, per-rank stages, the N-rank-against-one-rank check) and the launcher."""
import json  # noqa: F401
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from lumenos_amd import params as lp  # noqa: E402
from .job import CONFIGS, RHO_INV, Job  # noqa: E402,F401


class _DeviceBytes:
    """A span of device memory as torch sees it (CUDA array interface): lets RCCL collectives read and
    write the library's own buffers -- no staging copy, no host round trip."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def _as_tensor(ptr, nbytes):
    import torch
    return torch.as_tensor(_DeviceBytes(ptr, nbytes), device="cuda")


def all_to_all_sets(dist, send, recv, world, pg=None):
    """Block g of `send` (its g-th slice of count/world ciphertexts, contiguous: the layouts are ct-major)
    goes to rank g; block r of `recv` comes from rank r.  RCCL all-to-all on the sets' device memory; with
    gloo (one-GPU rehearsal) the same routing through the host."""
    import torch
    assert send.nbytes == recv.nbytes and send.count % world == 0
    send.ctx.sync()  # the producing kernels ran on the library's stream, the collective runs on torch's
    if dist.get_backend(pg) == "nccl":
        dist.all_to_all_single(_as_tensor(recv.device_ptr, recv.nbytes), _as_tensor(send.device_ptr, send.nbytes), group=pg)
        torch.cuda.synchronize()
        return
    host = torch.from_numpy(send.download().reshape(world, -1).view(np.int64))
    parts = [torch.empty_like(host) for _ in range(world)]
    dist.all_gather(parts, host, group=pg)  # gloo has no all-to-all: everybody sees everything, keeps its blocks
    rank = dist.get_rank()
    out = np.stack([p[rank].numpy().view(np.uint64) for p in parts]).reshape(recv.shape)
    recv.upload(out)


def all_gather_root(dist, ctx, dev_ptr, n, S, world, pg=None):
    """All-gather of the rank's n = S/world leaf digests (contiguous column blocks, so the gathered buffer
    is already in column order) and core.NewTree's root over them, all in device memory."""
    import torch
    assert n * world == S
    if dist.get_backend(pg) == "nccl":
        full = torch.empty(S * 32, dtype=torch.uint8, device="cuda")
        dist.all_gather_into_tensor(full, _as_tensor(dev_ptr, n * 32), group=pg)
        torch.cuda.synchronize()
        return ctx.merkle_root_device(full.data_ptr(), S)
    mine = torch.as_tensor(_DeviceBytes(dev_ptr, n * 32), device="cuda").cpu()
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=pg)
    full = torch.cat(parts).cuda()
    torch.cuda.synchronize()
    return ctx.merkle_root_device(full.data_ptr(), S)


def all_gather_digests(dist, dig, my_cols, S, world):
    """The one exchange of the multi-GPU path: S x 32 B of leaf digests (plus their column indices)
    over RCCL; returns the digests of all S leaves in column order."""
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    cap = (S + world - 1) // world + 128  # shards differ by at most one group of <= 128 columns
    buf = torch.zeros((cap, 36), dtype=torch.uint8)
    n = len(my_cols)
    buf[:n, :32] = torch.from_numpy(np.ascontiguousarray(dig))
    buf[:n, 32:] = torch.from_numpy(np.ascontiguousarray(my_cols.astype("<u4")).view(np.uint8).reshape(n, 4))
    cnt = torch.tensor([n], dtype=torch.int64)
    buf, cnt = buf.to(dev), cnt.to(dev)
    parts = [torch.empty_like(buf) for _ in range(world)]
    cnts = [torch.empty_like(cnt) for _ in range(world)]
    dist.all_gather(parts, buf)
    dist.all_gather(cnts, cnt)
    full = np.zeros((S, 32), dtype=np.uint8)
    seen = 0
    for p, c in zip(parts, cnts):
        k = int(c.item())
        a = p[:k].cpu().numpy()
        idx = np.ascontiguousarray(a[:, 32:]).view("<u4").reshape(k)
        full[idx] = a[:, :32]
        seen += k
    assert seen == S, f"digest shards cover {seen} of {S} leaves"
    return full


def device_identity(ctx_device):
    """what tells two ranks of one launch that they sit on the same physical GPU: host, the visibility masks the
    process runs under and the device ordinal it uses (torch.distributed.run gives every rank the same masks and its
    own ordinal; a launcher that pins one GPU per process gives every rank ordinal 0 under its own mask)"""
    import socket
    return "|".join([socket.gethostname()] + [os.environ.get(k, "") for k in
                    ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")] + [str(ctx_device)])


def join_ranks(ctx, rank, world, dist, deadline_s=None, identity=None):
    """One process per GPU: every rank joins the library's own RCCL communicator (lumen_group_create_rank) -- or none
    does.  Returns (group or None, [(ok, reason)] of all ranks), the same on every rank.

    ncclCommInitRank only returns once ALL ranks have arrived, so a rank must not find out about a problem inside
    it while its peers are already blocked there.  Hence two steps, both agreed over the control plane (gloo):
      1. what can be checked locally is checked BEFORE anybody joins: librccl loads and answers (ncclGetUniqueId is
         a local call), and no two ranks sit on the same physical device (RCCL refuses that communicator);
      2. the join itself runs under a deadline (LUMEN_BENCH_JOIN_TIMEOUT, default 180 s): a rank still inside
         ncclCommInitRank by then -- its peers failed asymmetrically and moved on -- exits non-zero, so that
         torch.distributed.run tears the whole job down instead of sitting in a 10-minute gloo timeout without a
         JSON line.  (Exit, not recovery: a process that has touched the GPU is never re-executed.)"""
    import threading
    from lumenos_amd.hip import Group, LumenError
    deadline_s = deadline_s or float(os.environ.get("LUMEN_BENCH_JOIN_TIMEOUT", "180"))
    uid, err = None, ""
    try:
        uid = Group.unique_id()  # loads librccl in this process; only rank 0's id is used
    except LumenError as e:
        err = f"rank {rank}: {e}"
    # (identity: the rehearsal in tests/dev/bench_per_rank_threads.py plays the ranks as threads on one GPU)
    mine = (not err, err, identity or device_identity(ctx.device), uid.tobytes() if uid is not None else b"")
    seen = [None] * world
    dist.all_gather_object(seen, mine)
    by_dev = {}
    for r, s_ in enumerate(seen):
        by_dev.setdefault(s_[2], []).append(r)
    shared = [v for v in by_dev.values() if len(v) > 1]
    if shared or not all(s_[0] for s_ in seen):
        why = (f"ranks {shared[0]} share one device: RCCL refuses two ranks on a device" if shared
               else next(s_[1] for s_ in seen if not s_[0]))
        return None, [(0, why)] * world  # nobody entered ncclCommInitRank
    box = {}

    def join():
        try:
            box["g"] = Group.join(ctx, rank, world, np.frombuffer(seen[0][3], dtype=np.uint8))
        except LumenError as e:
            box["err"] = str(e)

    t = threading.Thread(target=join, daemon=True)
    t.start()
    t.join(deadline_s)
    if t.is_alive():
        sys.stderr.write(f"[bench.py] rank {rank}: still inside ncclCommInitRank after {deadline_s:.0f} s -- a peer never "
                         f"arrived (it failed on its own and went on); exiting so that the launcher ends the job\n")
        sys.stderr.flush()
        os._exit(3)
    flags = [None] * world
    dist.all_gather_object(flags, (1 if "g" in box else 0, box.get("err", "")))
    if all(f[0] for f in flags):
        return box["g"], flags
    if "g" in box:
        box["g"].close()
    return None, flags


def attach_group(job, args, dist, new_nccl_group=None):
    """Puts the job's local ranks behind a lumen_group (the exchange inside the library) and returns the text of
    config.transport.  One process per GPU: rank 0 draws the communicator's id, the control-plane process group
    carries it; every rank says whether it could join, and if any could not ALL fall back to the torch.distributed
    path together."""
    from lumenos_amd.hip import Group, LumenError
    if dist is None:  # --single-process: this process owns every rank
        want = {"rccl": "auto", "copy": "copy", "torch": None}[args.transport]
        if want is None:
            raise SystemExit("bench.py: --transport torch needs one process per GPU (drop --single-process)")
        # (LUMEN_TRANSPORT_AUTO falls back to device copies by itself when RCCL cannot be loaded or initialised)
        job.group = Group(job.ctxs, transport=want)
        return f"lumen_group: {job.group.transport} ({job.group.transport_note})"
    import torch
    if args.transport == "torch":
        why = "--share-gpu: RCCL refuses two ranks on one device" if args.share_gpu else "--transport torch"
        return f"torch.distributed {dist.get_backend()} on aliased device memory ({why})"
    group, flags = join_ranks(job.ctx, job.rank, job.world, dist)
    job.group = group
    if job.group is not None:
        return f"lumen_group: {job.group.transport} (the library's own communicator, ncclCommInitRank; {job.group.transport_note})"
    reason = next(f[1] for f in flags if not f[0])
    # fall back together: the collectives of torch.distributed (RCCL) on tensors aliasing the library's memory
    # (with --share-gpu no RCCL of any kind can serve two ranks on the device: the rehearsal falls back to gloo)
    if new_nccl_group:
        job.nccl_pg = new_nccl_group()
        return f"torch.distributed nccl on aliased device memory (FALLBACK: the library's RCCL group failed: {reason})"
    backend, probe = ("gloo" if getattr(args, "share_gpu", False) else "nccl"), ""
    if backend == "nccl":
        # whatever kept the library's communicator from forming may keep torch's from forming too (same RCCL): try one
        # tiny collective, and if any rank cannot, ALL take the host-staged gloo path -- slow, but a number and a check
        import datetime
        try:
            pg = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=120))
            t = torch.ones(1, device="cuda")
            dist.all_reduce(t, group=pg)
            torch.cuda.synchronize()
            good, probe = int(t.item()) == job.world, ""
        except Exception as e:  # noqa: BLE001
            good, probe, pg = False, f"{type(e).__name__}: {e}", None
        verdicts = [None] * job.world
        dist.all_gather_object(verdicts, (good, probe))
        if all(v[0] for v in verdicts):
            job.nccl_pg = pg
        else:
            backend, probe = "gloo", "; torch's nccl group failed too: " + next(v[1] for v in verdicts if not v[0])[:300]
    if backend == "gloo":
        job.nccl_pg = dist.new_group(backend="gloo")
    return (f"torch.distributed {backend} on aliased device memory (FALLBACK: the library's RCCL group failed: {reason}{probe})")


def group_collectives(job):
    """per-collective HIP-event time and rate since the last reset (lumen_group_stats)"""
    out = {}
    for name in ("all_to_all_1", "all_to_all_2", "all_gather", "gather_to_root"):
        ms, sent, calls = job.group.stats(name)
        if calls:
            out[name] = {"calls": calls, "ms_per_call": round(ms / calls, 4), "MB_sent_per_rank_per_call": round(sent / calls / 1e6, 3),
                         "GBps_per_rank": round(sent / (ms * 1e-3) / 1e9, 2) if ms > 0 else None,
                         "GBps_all_ranks": round(sent * job.world / (ms * 1e-3) / 1e9, 2) if ms > 0 else None}
    return out


def check_against_single_rank(device, world, rank, local_devices, group_factory, dist):
    """An N-rank run at 2048x1024 (BASELINE config A) against a single-rank recompute on the same inputs: the Merkle
    root, a sample of every local rank's encoded columns, and the first and last MatR ciphertext of its block.
    Every process recomputes the whole job on its own first device (0.1 s at this size): rank r's input block is
    fill_random(1 + r), whoever generates it."""
    cfg = "2048x1024"
    j = Job(cfg, rank, world, device, 0, False, local_devices)
    if not j.lane_path:
        j.close()
        return {"ok": None, "note": f"{world} ranks cannot run the lane path at {cfg}"}
    group_factory(j)
    res = {"config": cfg, "ranks": world, "path": "lumen_group" if j.group is not None else "torch.distributed"}
    try:
        if j.group is not None:
            enc, lvl1, mat_r, mat_z, q, root = j.step_group(keep=True)
        else:
            enc, lvl1, mat_r, mat_z, q, root = j.step_lanes(dist, keep=True)
        ctx, own, Sw = j.ctx, j.cols // world, j.S // world
        full = ctx.new_set(j.cols, j.L)
        views = [full.slice(r * own, own).fill_random(1 + r) for r in range(world)]
        want_enc = ctx.encode(full, j.zero_ct, RHO_INV)
        want_l1 = ctx.rescale(want_enc, 2)
        want_root = ctx.merkle_build(ctx.leaf_digests(want_l1))[1]
        want_r = ctx.matrix_inner_sum(full, j.r_pt, j.rows)
        res["root_equal"] = bool(root == want_root)
        cols_ok, n_cols, mat_ok = True, 0, True
        for i, r in enumerate(j.local_ranks):
            for k in sorted({0, Sw // 3, Sw - 1}):
                cols_ok &= bool(np.array_equal(enc[i].download(k, 1), want_enc.download(r * Sw + k, 1)))
                n_cols += 1
            for k in (0, own - 1):
                mat_ok &= bool(np.array_equal(mat_r[i].download(k, 1), want_r.download(r * own + k, 1)))
        res["encoded_columns_checked"], res["encoded_columns_equal"] = n_cols, cols_ok
        res["mat_r_samples_equal"] = mat_ok
        if q is not None:
            res["queried_columns_equal"] = bool(np.array_equal(q.download(), ctx.gather(want_l1, j.query_idx).download()))
        res["ok"] = bool(res["root_equal"] and cols_ok and mat_ok and res.get("queried_columns_equal", True))
        for s_ in [q, want_r, want_l1, want_enc] + views + [full] + mat_r + mat_z + lvl1 + enc:
            if s_ is not None:
                s_.free()
    finally:
        j.close()
    return res


def multi_rank_report(job, args, dist, per_rank_prof, sec_per_step):
    """What makes the first run on a real node self-diagnosing: ranks the RCCL communicator saw, every
    collective's time and rate, per-rank stage times (each stage drained before the next) and per-rank roofline,
    and the N-rank-against-one-rank check.  Collected on rank 0 (all_gather_object over the control plane)."""
    mine = {"ranks": job.local_ranks}
    if job.group is not None:
        job.group.stats_reset()
        timers = {}
        job.step_group(timers=timers)
        mine["stage_s"] = {k: round(v, 5) for k, v in timers.items()}
        mine["collectives"] = group_collectives(job)
        mine["rccl_ranks_seen"] = job.group.rccl_ranks
    elif job.lane_path and dist is not None:
        timers = {}
        job.step_lanes(dist, timers=timers)
        mine["stage_s"] = {k: round(v, 5) for k, v in timers.items()}
        own, ct = job.cols // job.world, 2 * job.L * job.N * 8
        sent = {"all_to_all_1": own * ct * (job.world - 1) / job.world, "all_to_all_2": 2 * own * ct * (job.world - 1) / job.world,
                "all_gather": job.S // job.world * 32 * (job.world - 1)}
        mine["collectives"] = {n: {"calls": 1, "ms_per_call": round(timers[k] * 1e3, 4),  # host wall: these calls end drained
                                   "MB_sent_per_rank_per_call": round(b / 1e6, 3),
                                   "GBps_per_rank": round(b / timers[k] / 1e9, 2), "GBps_all_ranks": round(b * job.world / timers[k] / 1e9, 2)}
                               for n, k, b in (("all_to_all_1", "all_to_all_1_s", sent["all_to_all_1"]),
                                               ("all_to_all_2", "all_to_all_2_s", sent["all_to_all_2"]),
                                               ("all_gather", "digest_all_gather_and_root_s", sent["all_gather"]))}
        pg = getattr(job, "nccl_pg", None)
        mine["rccl_ranks_seen"] = dist.get_world_size(pg) if dist.get_backend(pg) == "nccl" else 0
    if per_rank_prof:
        mine["per_rank"] = [{"rank": r, "limb_ntts_executed": ex,
                             "roofline": ({k: rl[k] for k in ("kernel", "frac", "achieved", "avg_launch_ms")} if rl else None),
                             "limb_ntts_executed_per_s": round(ex / sec_per_step, 1) if ex else None}
                            for r, (rl, _, ex) in zip(job.local_ranks, per_rank_prof)]
    if not args.no_check and (job.group is not None or (job.lane_path and dist is not None)):
        def factory(j):
            if job.group is None:  # the torch.distributed path: the check job uses the same process groups
                if hasattr(job, "nccl_pg"):
                    j.nccl_pg = job.nccl_pg
            elif dist is None:
                from lumenos_amd.hip import Group
                j.group = Group(j.ctxs, transport="copy" if job.group.transport.startswith("copy") else "rccl")
            else:
                j.group, flags = join_ranks(j.ctx, job.rank, job.world, dist)
                if j.group is None:  # every rank gets the same answer: all raise, none is left inside a collective
                    raise RuntimeError("the check job's RCCL group could not be formed: " + next(f[1] for f in flags if not f[0]))
        try:
            mine["check"] = check_against_single_rank(job.ctx_device, job.world, job.rank, job.local_devices, factory, dist)
        except Exception as e:  # a failed check must not cost the measurement
            mine["check"] = {"ok": False, "error": f"{type(e).__name__}: {e}"}
    parts = [mine]
    if dist is not None:
        parts = [None] * job.world
        dist.all_gather_object(parts, mine)
    out = {"rccl_ranks_seen": max((p.get("rccl_ranks_seen", 0) for p in parts), default=0)}
    if any("collectives" in p for p in parts):  # a collective is as slow as its slowest rank
        names = sorted({n for p in parts for n in p.get("collectives", {})})
        out["collectives"] = {n: max((p["collectives"][n] for p in parts if n in p.get("collectives", {})),
                                     key=lambda e: e["ms_per_call"]) for n in names}
        out["per_rank_stage_s"] = {",".join(map(str, p["ranks"])): p.get("stage_s") for p in parts}
    pr = [e for p in parts for e in p.get("per_rank", [])]
    if pr:
        out["per_rank"] = pr
        out["limb_ntts_executed_all_ranks"] = sum(e["limb_ntts_executed"] or 0 for e in pr)
    checks = [p["check"] for p in parts if "check" in p]
    if checks:
        out["check"] = dict(checks[0], ok=all(c.get("ok") for c in checks),
                            failures=[c for c in checks if not c.get("ok")] or None)
    return out


def launch_ranks(args, argv):
    """`python bench.py --gpus N` for N > 1: this process never initialises the GPU (no torch.cuda, no HIP) --
    it starts one rank per GPU under torch.distributed.run as a CHILD process (never an exec), relays the
    child's output (rank 0's JSON line) and returns its exit code (torch.distributed.run exits non-zero when
    any rank fails)."""
    import subprocess
    # --standalone: the launcher picks its own free rendezvous port on the loopback interface (no window between
    # "found a free port" and "bound it" for another process to slip into)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", os.path.join(ROOT, "bench.py"), *argv]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:  # rank 0's JSON line goes to stdout; whatever a library chats there (gloo) to stderr
        out = sys.stdout if line.lstrip().startswith("{") else sys.stderr
        out.write(line)
        out.flush()
    return proc.wait()
