#!/usr/bin/env python3
"""Benchmark of the server-side homomorphic Ligero prover hot path on MI355X.

A "step" is one pass of Encode + Commit + InnerProduct(r) + InnerProduct(b) +
QueryCols (fhe/ligero.go:95-291) over one synthetic encrypted witness matrix
that is already resident in HBM.  Default workload = the configuration
BASELINE.json's metric is quoted on: 16384 x 4096, LogN = 14 (12 Q limbs, 2 P
limbs, rhoInv = 2, 309 queries).  It fits one GPU (about 75 GB of the 288 GB).

    python bench.py --gpus N --steps K --warmup W

works as typed for any N: for N > 1 this process never touches the GPU -- it starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>`
as a child, relays rank 0's JSON line and exits with the child's code.  Launched under
torch.distributed.run directly (RANK / WORLD_SIZE set) it is one rank of that job.

N > 1 (strong scaling, fixed job; N a power of two with N/world >= 64 lanes): rank r holds ONLY its block of
cols/N input columns.  Encode is lane-sharded (the ciphertext-axis transform never mixes
lanes): an all-to-all over xGMI turns the ranks' column blocks into lane shards of all
columns, every rank encodes its 1/N of the lanes, a second all-to-all hands every rank whole
ciphertexts of its block of encoded columns.  Everything else is per column (rescale + leaf
hashing; ct x pt + InnerSum + rescale on the rank's input columns; query gather).  The leaf
digests are all-gathered on device buffers (RCCL) and the Merkle root is built on the device.
Nothing is replicated.  Worlds the lane path cannot serve are REFUSED unless --allow-replicated asks for
the round-1 path (every rank holds the matrix and runs the mixing passes itself); config.parallelism names
the path that ran.

The exchange steps run INSIDE the library (include/lumenos_hip.h, lumen_group_*): --transport rccl (default) has
every rank join an RCCL communicator of the library's own (lumen_group_create_rank; torch.distributed, backend
gloo, only carries the 128-byte id, the barriers and the max over ranks); if that cannot be set up the run falls
back to --transport torch (the collectives of torch.distributed on tensors that alias the library's memory) and
says so in config.transport.  --single-process runs all N ranks from ONE process, one context per GPU behind
lumen_group_create -- the topology of the reference's server (one Go process, cmd/server/main.go:187-266); with
--share-gpu its N contexts sit on device 0 (copy transport: the one-GPU rehearsal).  For N > 1 the line carries
rccl_ranks_seen, per-collective times and GB/s, per-rank stage times and roofline, and `check`: root, encoded
columns and inner products of an N-rank run at 2048x1024 against a single-rank recompute.

Rank 0 prints ONE JSON line (see the keys at the bottom of main()).  At N = 1 the default run also carries
  * marshal_s / io_inclusive_s: the proof as wire-format bytes in page-locked host memory
    (EncryptedProof.WriteTo, fhe/ligero.go:659-705) and a step that starts with the input ciphertexts in
    host memory and ends there (never `value`);
  * other_configs: short passes over the other BASELINE.json configurations (2048x1024, 4096x2048,
    8192x4096, and 16384x4096 with the ring switch to LogN = 10), so that the driver's one command attests them.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# the parts (bench_lib/); re-exported here because tools/ and tests/ say bench.Job, bench.join_ranks, ...
from bench_lib.job import CONFIGS, PUBLISHED_SECONDS, RHO_INV, SECURITY_BITS, Job, owned_queries  # noqa: E402,F401
from bench_lib.legs import (cpu_baseline, hashlib_sha, io_leg, other_configs, plain_ntt_rates,  # noqa: E402,F401
                            plain_prover_seconds, stage_seconds, timed_steps)
from bench_lib.multi import (all_gather_digests, all_gather_root, all_to_all_sets, attach_group,  # noqa: E402,F401
                             check_against_single_rank, device_identity, group_collectives, join_ranks, launch_ranks,
                             multi_rank_report)
from bench_lib.report import (NTT_KERNELS, BoxProbe, algorithmic_bytes, limb_ntt_census, pmc_entry,  # noqa: E402,F401
                              pmc_table, profile_kernels, step_spread, valu_roof)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="16384x4096", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-profile", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --share-gpu rehearses the N>1 path on a one-GPU box")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks use device 0 (rehearsal only)")
    ap.add_argument("--single-process", action="store_true",
                    help="N > 1: ONE process drives all N ranks, one context per GPU behind lumen_group_create (the "
                         "reference server's topology); with --share-gpu the N contexts share device 0")
    ap.add_argument("--transport", default=None, choices=["rccl", "copy", "torch"],
                    help="N > 1: rccl = the library's own RCCL communicator (default; falls back to torch if it cannot "
                         "be set up); copy = stream-ordered peer copies (--single-process only); torch = "
                         "torch.distributed collectives on tensors aliasing the library's memory (the round-3 path; the "
                         "default with --share-gpu, where RCCL refuses two ranks on one device -- asking for rccl there "
                         "rehearses the refusal and the fallback)")
    ap.add_argument("--no-check", action="store_true",
                    help="N > 1: skip the comparison of an N-rank run at 2048x1024 with a single-rank recompute")
    ap.add_argument("--allow-replicated", action="store_true",
                    help="N>1 worlds the lane-sharded path cannot serve: run the round-1 replicated-input path "
                         "instead of refusing (named in config.parallelism)")
    ap.add_argument("--include-io", action="store_true", help="kept for old command lines: the io leg is on by default")
    ap.add_argument("--no-io", action="store_true",
                    help="skip marshal_s / io_inclusive_s (upload of the inputs from and the proof's wire bytes into "
                         "page-locked host memory; reported beside `value`, never as `value`)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short passes over the other BASELINE.json configurations")
    ap.add_argument("--other-steps", type=int, default=3)
    ap.add_argument("--ring-switch-logn", type=int, default=0,
                    help="BASELINE config 5: ring-switch MatR/MatZ to this ring degree (fhe/ring_switch.go)")
    args = ap.parse_args()
    # the host driver of this pool only does dmabuf IPC: RCCL (the library's own group and torch's) needs this before
    # the HIP runtime starts in this process
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.transport is None:
        args.transport = "torch" if (args.share_gpu and not args.single_process) else "rccl"

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "RANK" not in os.environ and args.gpus > 1 and not args.single_process:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.single_process:
        if world != 1:
            sys.exit("bench.py: --single-process is one process; do not start it under torch.distributed.run")
        world = args.gpus
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    import torch
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device: the lumenos HIP path has no CPU fallback")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist, transport = None, None
    if world > 1 and not args.single_process:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # control plane (the id of the library's communicator, barriers, max over ranks): gloo.  Data plane: the
        # library's own RCCL communicator, or with --transport torch the process group itself ("nccl" is RCCL on
        # ROCm).  A rank that dies must not leave the others waiting in a collective for ever.
        use_torch = args.transport == "torch"
        # (--share-gpu: RCCL of any kind refuses two ranks on one device -- "Duplicate GPU detected" -- so the
        # torch path of the one-GPU rehearsal is gloo whatever --dist-backend says)
        backend = "gloo" if (not use_torch or args.share_gpu) else args.dist_backend
        dist.init_process_group(backend, timeout=datetime.timedelta(minutes=10))

    local_devices = None
    if args.single_process and world > 1:
        if not args.share_gpu and torch.cuda.device_count() < world:
            sys.exit(f"bench.py: --single-process --gpus {world} needs {world} visible devices (have "
                     f"{torch.cuda.device_count()}); --share-gpu puts every rank on device 0")
        local_devices = [0] * world if args.share_gpu else list(range(world))
    job = Job(args.config, rank, world, local_rank, args.ring_switch_logn, args.allow_replicated, local_devices)
    if world > 1 and job.lane_path:
        transport = attach_group(job, args, dist)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        for c in job.ctxs:
            c.sync()

    probe = BoxProbe(local_rank) if rank == 0 else None  # (clocks / power / temperatures: idle now, once under load, right after)
    for _ in range(args.warmup):
        job.step(dist)
    barrier()
    if probe:
        probe.start_timed_region()
    step_s = []
    t0 = t_prev = time.perf_counter()
    for _ in range(args.steps):
        job.step(dist)
        t_now = time.perf_counter()  # (a step ends drained: its own stream synchronisation -- nothing is added here)
        step_s.append(t_now - t_prev)
        t_prev = t_now
    barrier()
    elapsed = time.perf_counter() - t0
    box = probe.end_timed_region() if probe else None
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    sec_per_step = elapsed / args.steps

    roofline, stages, executed, per_rank_prof = ((None, None, None, None) if args.no_kernel_profile
                                                 else profile_kernels(job, dist, args.config))
    multi = multi_rank_report(job, args, dist, per_rank_prof, sec_per_step) if world > 1 else None
    ntt_kernel = plain_ntt_rates(job) if world == 1 and not args.no_kernel_profile else None
    io = io_leg(job, args.config) if world == 1 and not args.no_io else None
    others = None
    if world == 1 and not args.no_other_configs and not args.ring_switch_logn:
        others = other_configs(job, args, sec_per_step, local_rank, barrier)
    if rank == 0:
        census = limb_ntt_census(job.rows, job.cols, job.L, job.K, job.log_n)
        out = {
            "metric": f"prove_eval_seconds_{args.config}",
            "value": round(sec_per_step, 4),
            "unit": "s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(sec_per_step * 1e3, 2),
            "higher_is_better": False,
            "scaling": "strong",
            "vs_baseline": round(sec_per_step / PUBLISHED_SECONDS[args.config], 6),
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": f"Encode+Commit+InnerProduct(r,b)+QueryCols {args.config} LogN={job.log_n} "
                                   f"L={job.L} K={job.K} rhoInv={RHO_INV} queries={job.queries}"
                                   + (f" +ring-switch->LogN={args.ring_switch_logn}" if args.ring_switch_logn else ""),
                       "parallelism": ("1 GPU: the whole job resident in HBM" if world == 1 else
                                       f"lane path: {world} GPUs, lane-sharded Encode between two xGMI all-to-alls, columns "
                                       "sharded elsewhere, digest all-gather + Merkle root on device buffers"
                                       if job.lane_path else
                                       f"REPLICATED path (--allow-replicated): every one of the {world} ranks holds the whole "
                                       "input and repeats the mixing passes of Encode; columns sharded elsewhere; digest "
                                       "all-gather through the host"),
                       "lane_path": bool(job.lane_path),
                       "transport": transport,
                       "topology": (None if world == 1 else "single process, one context per rank (lumen_group_create)"
                                    if args.single_process else "one process per GPU (torch.distributed.run)"),
                       "baseline_ref": "BASELINE.md: reference Go/Lattigo CPU, m7i.8xlarge 32 vCPU"},
            # limb transforms the device EXECUTES per step (sum of the NTT kernels' units: the rescale to level 1
            # runs on coefficients, 14 transforms per polynomial instead of the reference's 75) ...
            "limb_ntts_executed_per_s": round((multi or {}).get("limb_ntts_executed_all_ranks", executed * world)
                                              / sec_per_step, 1) if executed else None,
            # ... and the reference's own transform count for the same step (SURVEY 8d census) over the same time
            "limb_ntts_reference_equiv_per_s": round(census / sec_per_step, 1),
            "ct_ntts_reference_equiv_per_s": round(census / sec_per_step / (2 * job.L), 1),
            # per-step spread of the timed region and the box it ran on: clocks / power / temperatures idle, under load and
            # right after, device and host identity -- what a cross-round comparison of `value` has to be read against
            "step_ms": step_spread(step_s),
            "box": box,
            "roofline": roofline,
            "ntt_kernel": ntt_kernel,
            # the evaluation keys a client posts, from pageable host memory to usable on the device
            # (lumen_load_galois_key: upload + conversion to the gadget product's form, once per client)
            "load_galois_keys": {"keys": job.key_load_bytes // max(1, (job.L + job.K - 1) // job.K * 2 * (job.L + job.K) * job.N * 8),
                                 "GB": round(job.key_load_bytes / 1e9, 3), "seconds": round(job.key_load_s, 4)},
            "kernels": stages,
        }
        if multi:
            out.update(multi)
        if io:
            out.update(io)
        if others:
            out["other_configs"] = others
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.config)
        print(json.dumps(out), flush=True)
    # orderly teardown: every rank is past its last collective (the report's gather above) before any communicator
    # goes -- the group before its contexts, the library's RCCL communicator before the control plane
    try:
        if dist is not None:
            dist.barrier()
        job.close()
    except Exception as e:  # noqa: BLE001 -- the line is out: a teardown hiccup must not turn the run into a failure
        sys.stderr.write(f"[bench.py] teardown: {type(e).__name__}: {e}\n")
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
