#!/usr/bin/env python3
"""Benchmark of the server-side homomorphic Ligero prover hot path on MI355X.

A "step" is one pass of Encode + Commit + InnerProduct(r) + InnerProduct(b) +
QueryCols (fhe/ligero.go:95-291) over one synthetic encrypted witness matrix
that is already resident in HBM.  Default workload = the configuration
BASELINE.json's metric is quoted on: 16384 x 4096, LogN = 14 (12 Q limbs, 2 P
limbs, rhoInv = 2, 309 queries).  It fits one GPU (about 75 GB of the 288 GB).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1 (strong scaling, fixed job; N a power of two): rank r holds ONLY its block of cols/N
input columns.  Encode is lane-sharded (the ciphertext-axis transform never mixes
lanes): an all-to-all over xGMI turns the ranks' column blocks into lane shards of all
columns, every rank encodes its 1/N of the lanes, a second all-to-all hands every rank whole
ciphertexts of its block of encoded columns.  Everything else is per column (rescale + leaf
hashing; ct x pt + InnerSum + rescale on the rank's input columns; query gather).  The leaf
digests are all-gathered on device buffers (RCCL) and the Merkle root is built on the device.
Nothing is replicated.  (Other N: the round-1 path -- every rank holds the matrix and runs the
mixing passes itself.)

Rank 0 prints ONE JSON line (see the keys at the bottom of main()).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from lumenos_amd import params as lp  # noqa: E402

CONFIGS = {
    # name: rows, cols, logN   (BASELINE.json configs / README shapes)
    "2048x1024": (2048, 1024, 12),
    "4096x2048": (4096, 2048, 12),
    "8192x4096": (8192, 4096, 13),
    "16384x4096": (16384, 4096, 14),
}
RHO_INV = 2
SECURITY_BITS = 128
# published CPU number for this exact metric (Encode+Commit+Prove,
# 16384x4096, m7i.8xlarge 32 vCPU, pure-Go Lattigo)
PUBLISHED_SECONDS = {"16384x4096": 416.6, "8192x4096": 189.1, "4096x2048": 38.84, "2048x1024": 15.81}


def limb_ntt_census(rows, cols, L, K, log_n):
    """Polynomial limb-NTT count of one step (SURVEY 8a formulas)."""
    S = cols * RHO_INV
    beta = (L + K - 1) // K
    rescale = sum(2 * (1 + l) for l in range(2, L))  # per ciphertext, level L-1 -> 1
    rot = (rows.bit_length() - 1)
    per_rot = L + (beta * (L + K) - L) + 2 * K + 2 * L
    commit = S * rescale
    inner = 2 * cols * (rot * per_rot + rescale)
    return commit + inner


PMC_NAMES = {"ks_modup_ntt": "k_modup_ntt", "ks_moddown_ntt": "k_moddown_ntt", "rescale_limb_ntt": "k_rescale_limb",
             "rescale_last_intt": "k_rescale_last", "limb_ntt": "k_limb_ntt", "limb_intt": "k_limb_ntt",
             "ks_intt_c1": "k_limb_ntt", "ks_intt_p": "k_limb_ntt"}


def pmc_table(cfg):
    """The committed rocprofv3 PMC summary of this same command (tools/profile_bench.sh: separate
    FETCH_SIZE / WRITE_SIZE / SQ passes, gfx950 corrections applied by tools/collect_pmc.py).  bench.py
    cannot collect hardware counters itself.  The summary is stamped with the hash of the HIP sources it
    was measured on: a different build gets None, not somebody else's counters."""
    path = os.path.join(ROOT, "profiles", f"pmc_traffic_{cfg}.json")
    if not os.path.exists(path):
        return None
    tab = json.load(open(path))
    from lumenos_amd import _build
    if tab.get("__source_hash__") != _build.source_hash():
        return None
    return tab


def pmc_entry(tab, kernel):
    if not tab:
        return None
    for name, v in tab.items():
        if isinstance(v, dict) and name.startswith(PMC_NAMES.get(kernel, kernel)):
            return v
    return None


def algorithmic_bytes(job, name, launches, units):
    """SURVEY 8d bytes of ALL launches of one profiled kernel family in a step (None: not tabulated)."""
    N, L, K = job.N, job.L, job.K
    LK, beta = L + K, (L + K - 1) // K
    ct = lambda nl: 2 * nl * N * 8
    if name in NTT_KERNELS:
        return 16.0 * N * units                      # one limb transform: read + write N words
    if name == "ks_mac":                             # per column: beta digits x LK limbs read, 2 x LK limbs written;
        return units * (beta * LK + 2 * LK) * N * 8.0 + launches * 2 * beta * LK * N * 8.0  # + the key once per launch
    if name == "ks_pack_v":                          # in place on the digit pairs: read + write
        return None                                  # (units are columns of two different shapes: c1 and P limbs)
    if name == "ct_axis_pass":                       # Encode: read cols + 1 ciphertexts, write S (all passes together)
        return (job.cols + 1 + job.S) * float(ct(L))
    if name == "mul_plain":
        return units * 2.0 * ct(L)
    if name == "rescale_coef":                       # per polynomial: read nl limbs, write 2
        return units * (L + 2) * N * 8.0
    if name == "leaf_sha256":
        return units * float(ct(2))
    return None


NTT_KERNELS = ("ks_modup_ntt", "ks_moddown_ntt", "rescale_limb_ntt", "rescale_last_intt", "ks_intt_c1", "ks_intt_p",
               "limb_ntt", "limb_intt", "rescale_intt", "rescale_ntt")


class Job:
    """Device-resident inputs of one prover run + the step function."""

    def __init__(self, cfg, rank, world, device, ring_switch_logn=0):
        from lumenos_amd.hip import Context
        self.rows, self.cols, self.log_n = CONFIGS[cfg]
        self.rank, self.world = rank, world
        P = lp.generate_bgv_params_for_ntt(self.cols, self.log_n)
        self.P = P
        self.L, self.K, self.N = len(P.q), len(P.p), P.N
        self.S = self.cols * RHO_INV
        self.queries = lp.calculate_queries(SECURITY_BITS, RHO_INV)
        self.ctx = ctx = Context(P.log_n, P.q, P.p, P.psi, P.T, device=device)
        ctx.field_set(np.array(lp.field_roots_forward(P.T, self.S), dtype=np.uint64))
        rng = np.random.default_rng(1)
        # lane-sharded Encode needs a power-of-two world whose lane shards keep at least one tile
        self.lane_path = world > 1 and (world & (world - 1)) == 0 and self.cols % world == 0 and (self.N // world) >= 64
        self.logw = world.bit_length() - 1 if self.lane_path else 0
        # synthetic inputs: uniform residues (kernels are data-independent); with the lane path a
        # rank only ever holds its own block of columns
        own = self.cols // world if self.lane_path else self.cols
        self.matrix = ctx.new_set(own, self.L).fill_random(1 + (rank if self.lane_path else 0))

        def rand_limbs(mods, shape_tail):
            out = np.empty((len(mods),) + shape_tail, dtype=np.uint64)
            for i, m in enumerate(mods):
                out[i] = rng.integers(0, m, size=shape_tail, dtype=np.uint64)
            return out

        self.zero_ct = np.ascontiguousarray(rand_limbs(P.q, (2, self.N)).transpose(1, 0, 2))
        self.r_pt = rand_limbs(P.q, (self.N,))
        self.b_pt = rand_limbs(P.q, (self.N,))
        beta = (self.L + self.K - 1) // self.K
        for g in ctx.inner_sum_galois_elements(self.rows):
            evk = np.ascontiguousarray(
                rand_limbs(P.q + P.p, (beta, 2, self.N)).transpose(1, 2, 0, 3))  # [beta][2][L+K][N]
            ctx.load_galois_key(g, evk)
        self.query_idx = rng.integers(0, self.S, size=self.queries).astype(np.uint32)
        self.ring_switch_logn = ring_switch_logn
        if ring_switch_logn:
            # the whole evaluation key a client posts (cmd/client/main.go:124-131): [rns][pw2][2][L+K][N]; with
            # two special primes that is one Galois key's size (no power-of-two digits)
            rns, pw2 = ctx.ringswitch_key_shape(13)[:2]
            key = np.ascontiguousarray(rand_limbs(P.q + P.p, (rns, pw2, 2, self.N)).transpose(1, 2, 3, 0, 4))
            ctx.load_ringswitch_key(ring_switch_logn, key)
        # column shards (input columns; encoded columns are sharded by the transform itself)
        self.col_lo, self.col_hi = self.cols * rank // world, self.cols * (rank + 1) // world
        if self.lane_path:  # self.matrix IS the rank's block; its slice of the one Enc(0) for the lane Encode
            nw = self.N // world
            self.zero_lanes = np.ascontiguousarray(self.zero_ct[:, :, rank * nw:(rank + 1) * nw])
        ctx.sync()

    # ---- host I/O of a prover run, measured by --include-io
    def io_setup(self):
        """Pinned host buffers (lumen_host_alloc): the input ciphertexts as the Go shim's stage() lays
        them out, and room for the proof's ciphertexts; a clone context whose stream carries downloads
        while the main context computes."""
        from lumenos_amd.hip import pinned_empty
        self.h_matrix = pinned_empty((self.cols, 2, self.L, self.N))
        self.matrix.download_into(self.h_matrix)  # content: the synthetic matrix itself
        self.h_r = pinned_empty((self.col_hi - self.col_lo, 2, 2, self.N))
        self.h_z = pinned_empty((self.col_hi - self.col_lo, 2, 2, self.N))
        self.h_q = pinned_empty((self.queries, 2, 2, self.N))
        self.io_ctx = self.ctx.clone()

    def step_io(self):
        """One step including the PCIe legs a drop-in pays: upload of the input ciphertexts (12.9 GB at D),
        download of MatR / MatZ / the queried columns (4.4 GB at D).  MatR crosses the link on the clone's
        stream while MatZ is computed; the upload cannot overlap (Encode needs every column)."""
        import threading
        ctx = self.ctx
        t = {}
        t0 = time.perf_counter()
        self.matrix.upload(self.h_matrix)
        t["upload_s"] = time.perf_counter() - t0
        mine = ctx.encode(self.matrix, self.zero_ct, RHO_INV)
        lvl1 = ctx.rescale(mine, 2)
        mine.free()
        ctx.leaf_digests_begin(lvl1)
        cols = self.matrix.slice(self.col_lo, self.col_hi - self.col_lo)
        mat_r = ctx.matrix_inner_sum(cols, self.r_pt, self.rows)
        ctx.sync()
        th = threading.Thread(target=lambda: self.io_ctx.download_into(mat_r, self.h_r))
        th.start()
        mat_z = ctx.matrix_inner_sum(cols, self.b_pt, self.rows)
        cols.free()
        q = ctx.gather(lvl1, self.query_idx)
        dig = ctx.leaf_digests_end()
        nodes, root = ctx.merkle_build(dig)
        ctx.sync()
        t1 = time.perf_counter()
        mat_z.download_into(self.h_z)
        q.download_into(self.h_q)
        th.join()
        t["download_tail_s"] = time.perf_counter() - t1
        t["total_s"] = time.perf_counter() - t0
        for s in (q, mat_r, mat_z, lvl1):
            s.free()
        return t

    def step_lanes(self, dist):
        """One step on `world` ranks with the lane-sharded Encode (module docstring)."""
        ctx, W, rank = self.ctx, self.world, self.rank
        Sw = self.S // W
        # ---- Commit: Encode.  own columns -> lane blocks -> all-to-all -> lane shard of ALL columns
        blocks = ctx.lanes_split(self.matrix, self.logw)
        lanes = ctx.new_set_lanes(self.cols, self.L, self.logw)
        all_to_all_sets(dist, blocks, lanes, W)
        blocks.free()
        enc = ctx.encode(lanes, self.zero_lanes, RHO_INV)  # this rank's lanes of all S encoded columns
        lanes.free()
        recv = ctx.new_set_lanes(self.S, self.L, self.logw)
        all_to_all_sets(dist, enc, recv, W)                  # block h of every shard -> rank h
        enc.free()
        mine = ctx.lanes_assemble(recv)                      # whole ciphertexts of columns [rank*Sw, (rank+1)*Sw)
        recv.free()
        # ---- Commit: leaves on this rank's encoded columns, hashed under the inner products
        lvl1 = ctx.rescale(mine, 2)
        mine.free()
        ctx.leaf_digests_begin(lvl1)
        # ---- Prove: inner products on this rank's input columns
        mat_r = ctx.matrix_inner_sum(self.matrix, self.r_pt, self.rows)
        mat_z = ctx.matrix_inner_sum(self.matrix, self.b_pt, self.rows)
        if self.ring_switch_logn:
            ctx.ring_switch(mat_r)
            ctx.ring_switch(mat_z)
        own = self.query_idx[(self.query_idx >= rank * Sw) & (self.query_idx < (rank + 1) * Sw)] - rank * Sw
        q = ctx.gather(lvl1, own.astype(np.uint32))
        # ---- Commit, concluded: all-gather of the digests on device buffers, Merkle root on the device
        ptr, n = ctx.leaf_digests_end_device()
        root = all_gather_root(dist, ctx, ptr, n, self.S, W)
        ctx.sync()
        for s in (q, mat_r, mat_z, lvl1):
            s.free()
        return root

    def step(self, dist=None):
        if self.lane_path and dist is not None:
            return self.step_lanes(dist)
        ctx = self.ctx
        # ---- Commit: Encode (fhe/code.go:8-34); with several ranks each keeps the encoded columns
        # its share of the transform's final pass produces
        if self.world > 1:
            mine, my_cols = ctx.encode_shard(self.matrix, self.zero_ct, RHO_INV, self.rank, self.world)
        else:
            mine, my_cols = ctx.encode(self.matrix, self.zero_ct, RHO_INV), np.arange(self.S, dtype=np.uint32)
        # ---- Commit: leaves (fhe/ligero.go:126-183) on this rank's columns
        lvl1 = ctx.rescale(mine, 2)
        mine.free()
        # the leaves are hashed on a side stream while the inner products run: Prove samples r without
        # the root in the transcript (fhe/ligero.go:198-199), so nothing below depends on them
        ctx.leaf_digests_begin(lvl1)
        # ---- Prove: <r, M> and <b, M> (fhe/ligero.go:231-242, 299-370) on this rank's columns
        cols = self.matrix.slice(self.col_lo, self.col_hi - self.col_lo)
        mat_r = ctx.matrix_inner_sum(cols, self.r_pt, self.rows)
        mat_z = ctx.matrix_inner_sum(cols, self.b_pt, self.rows)
        cols.free()
        if self.ring_switch_logn:  # ligero.go:336-342: RingSwitchNew on every inner-product output
            ctx.ring_switch(mat_r)
            ctx.ring_switch(mat_z)
        # ---- Prove: query columns (fhe/ligero.go:261-280): already at level 1 from Commit
        q = ctx.gather(lvl1, owned_queries(self.query_idx, my_cols))
        # ---- Commit, concluded: digests -> (all-gather) -> Merkle tree (core/tree.go:113-163)
        dig = ctx.leaf_digests_end()
        if dist is not None and self.world > 1:
            dig = all_gather_digests(dist, dig, my_cols, self.S, self.world)
        nodes, root = ctx.merkle_build(dig)
        ctx.sync()
        for s in (q, mat_r, mat_z, lvl1):
            s.free()
        return root


def owned_queries(query_idx, my_cols):
    """Local positions (in this rank's ascending column list) of the queried columns it owns."""
    pos = np.searchsorted(my_cols, query_idx)
    pos = np.clip(pos, 0, len(my_cols) - 1)
    own = my_cols[pos] == query_idx
    return pos[own].astype(np.uint32)


class _DeviceBytes:
    """A span of device memory as torch sees it (CUDA array interface): lets RCCL collectives read and
    write the library's own buffers -- no staging copy, no host round trip."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def _as_tensor(ptr, nbytes):
    import torch
    return torch.as_tensor(_DeviceBytes(ptr, nbytes), device="cuda")


def all_to_all_sets(dist, send, recv, world):
    """Block g of `send` (its g-th slice of count/world ciphertexts, contiguous: the layouts are ct-major)
    goes to rank g; block r of `recv` comes from rank r.  RCCL all-to-all on the sets' device memory; with
    gloo (one-GPU rehearsal) the same routing through the host."""
    import torch
    assert send.nbytes == recv.nbytes and send.count % world == 0
    send.ctx.sync()  # the producing kernels ran on the library's stream, the collective runs on torch's
    if dist.get_backend() == "nccl":
        dist.all_to_all_single(_as_tensor(recv.device_ptr, recv.nbytes), _as_tensor(send.device_ptr, send.nbytes))
        torch.cuda.synchronize()
        return
    host = torch.from_numpy(send.download().reshape(world, -1).view(np.int64))
    parts = [torch.empty_like(host) for _ in range(world)]
    dist.all_gather(parts, host)  # gloo has no all-to-all: everybody sees everything, keeps its blocks
    rank = dist.get_rank()
    out = np.stack([p[rank].numpy().view(np.uint64) for p in parts]).reshape(recv.shape)
    recv.upload(out)


def all_gather_root(dist, ctx, dev_ptr, n, S, world):
    """All-gather of the rank's n = S/world leaf digests (contiguous column blocks, so the gathered buffer
    is already in column order) and core.NewTree's root over them, all in device memory."""
    import torch
    assert n * world == S
    if dist.get_backend() == "nccl":
        full = torch.empty(S * 32, dtype=torch.uint8, device="cuda")
        dist.all_gather_into_tensor(full, _as_tensor(dev_ptr, n * 32))
        torch.cuda.synchronize()
        return ctx.merkle_root_device(full.data_ptr(), S)
    mine = torch.as_tensor(_DeviceBytes(dev_ptr, n * 32), device="cuda").cpu()
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
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


def cpu_baseline(cfg, budget_s=20.0):
    """Time the CPU oracle (a port, not the Go reference) on a bounded sample of the same
    workload and extrapolate to one step.  Test infrastructure used as a reported baseline only."""
    # a 1-GPU box owns a 16-core share of the host (os.cpu_count() reports the whole machine)
    cores = min(len(os.sched_getaffinity(0)), 16)
    os.environ["OMP_NUM_THREADS"] = str(cores)  # before libgomp starts
    from oracle.loader import Oracle, Params
    rows, cols, log_n = CONFIGS[cfg]
    o = Oracle()
    P = Params.for_ntt(o, cols, log_n, lp.T_REFERENCE)
    L, N, S = P.L, P.N, cols * RHO_INV
    rng = np.random.default_rng(3)

    def rand_ct(n, nl, NN=N):
        out = np.empty((n, 2, nl, NN), dtype=np.uint64)
        for l in range(nl):
            out[:, :, l, :] = rng.integers(0, P.moduli[l], size=(n, 2, NN), dtype=np.uint64)
        return out

    # Encode: single-threaded in the reference; sample = 1 limb of a
    # small ring (lanes scale linearly), full ciphertext count
    from tests.helpers import make_params
    Ps = make_params(o, 8, 1, num_p=0)
    roots = o.field_roots(lp.T_REFERENCE, S)
    m = np.empty((cols, 2, 1, Ps.N), dtype=np.uint64)
    m[:] = rng.integers(0, Ps.moduli[0], size=m.shape, dtype=np.uint64)
    z = m[0].copy()
    t0 = time.time()
    Ps.ct_encode(m, RHO_INV, z, roots)
    t_enc = (time.time() - t0) * (2 * L * N) / (2 * 1 * Ps.N)
    # Commit leaves: rescale + serialise + SHA-256 on `cores` columns (OpenMP over columns)
    n_c = cores
    enc = rand_ct(n_c, L)
    t0 = time.time()
    P.commit_leaves(enc)
    t_commit = (time.time() - t0) * S / n_c
    # InnerProduct: MulNew + InnerSum + rescale on `cores` columns, one vector
    gl = P.inner_sum_galois_elements(rows)
    evk = np.empty(P.evk_shape(), dtype=np.uint64)
    for t_i, mod in enumerate(P.moduli):
        evk[:, :, t_i, :] = rng.integers(0, mod, size=(evk.shape[0], 2, N), dtype=np.uint64)
    evks = [evk] * len(gl)
    pt = np.stack([rng.integers(0, P.moduli[l], size=N, dtype=np.uint64) for l in range(L)])
    n_i = cores
    mat = rand_ct(n_i, L)
    t0 = time.time()
    P.matrix_inner_sum(mat, pt, rows, evks)
    t_inner = (time.time() - t0) * (2 * cols) / n_i
    total = t_enc + t_commit + t_inner
    return {
        "value": round(total, 2), "unit": "s", "cores": cores, "kind": "port",
        "sample": (f"oracle (C restatement, OpenMP over columns): Encode on 1/{(2 * L * N) // (2 * Ps.N)} of the lanes "
                   f"(1 thread, as the reference), Commit leaves on {n_c}/{S} columns, InnerProduct on {n_i}/{2 * cols} "
                   f"column-vectors; extrapolated linearly; query reuses Commit's level-1 columns"),
        "stages_s": {"encode": round(t_enc, 2), "commit": round(t_commit, 2), "inner_product": round(t_inner, 2)},
    }


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
    ap.add_argument("--include-io", action="store_true",
                    help="also time steps that upload the input ciphertexts from and download the proof's "
                         "ciphertexts to pinned host memory (reported as io_inclusive_s; never `value`)")
    ap.add_argument("--ring-switch-logn", type=int, default=0,
                    help="BASELINE config 5: ring-switch MatR/MatZ to this ring degree (fhe/ring_switch.go)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    import torch
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device: the lumenos HIP path has no CPU fallback")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.dist_backend)  # "nccl" is RCCL on ROCm

    job = Job(args.config, rank, world, local_rank, args.ring_switch_logn)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        job.ctx.sync()

    for _ in range(args.warmup):
        job.step(dist)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        job.step(dist)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    sec_per_step = elapsed / args.steps

    # ---- dominant-kernel roofline: one more (untimed) step with HIP events around every launch
    roofline, stages, executed = None, None, None
    if not args.no_kernel_profile:
        job.ctx.prof_reset()
        job.ctx.prof_enable(True)
        job.step(dist)
        job.ctx.prof_enable(False)
        tab = {k: job.ctx.prof_read(k) for k in job.ctx.prof_names()}
        pmc = pmc_table(args.config)
        stages = {}
        for k, (ms, launches, units) in sorted(tab.items()):
            e = {"ms": round(ms, 3), "launches": launches, "units": units}
            ab = algorithmic_bytes(job, k, launches, units)
            if ab and ms > 0:  # SURVEY 8d bytes / HIP-event time of the launches, against the 8 TB/s HBM peak
                e["alg_gbps"] = round(ab / (ms * 1e-3) / 1e9, 1)
                e["hbm_frac"] = round(ab / (ms * 1e-3) / 8e12, 4)
            stages[k] = e
        ntt_kernels = {k: v for k, v in tab.items() if k in NTT_KERNELS}
        executed = sum(v[2] for v in ntt_kernels.values())
        if ntt_kernels:
            dom = max(ntt_kernels, key=lambda k: ntt_kernels[k][0])
            ms, launches, units = ntt_kernels[dom]
            alg_bytes_per_launch = 16.0 * job.N * units / launches  # 16*N B per limb transform
            achieved = alg_bytes_per_launch / (ms / launches * 1e-3) / 1e9
            pe = pmc_entry(pmc, dom)
            sq = (pe or {}).get("sq_per_launch") or {}
            bfly = units / launches * job.N / 2 * job.log_n  # butterflies of one launch
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": 8000.0,
                        "unit": "GB/s", "frac": round(achieved / 8000.0, 4),
                        "traffic": round(pe["hbm_bytes_per_launch"]) if pe and pe.get("hbm_bytes_per_launch") else None,
                        "avg_launch_ms": round(ms / launches, 4), "limb_ntts_per_launch": units // launches,
                        # what actually bounds the kernel: 64-bit modular butterflies on the VALU (no MFMA).  From
                        # the committed SQ counters of this build (null if the profile is of another build):
                        # fraction of SIMD cycles issuing VALU work, and wave-level VALU instructions per butterfly
                        "valu_frac": round(pe["valu_busy_frac"], 4) if pe and pe.get("valu_busy_frac") else None,
                        "valu_insts_per_butterfly": round(sq["SQ_INSTS_VALU"] * 64.0 / bfly, 2)
                        if sq.get("SQ_INSTS_VALU") else None,
                        "note": "VALU-bound integer kernel: the >= 50 % HBM target of north_star is not reachable at "
                                "10 multiply-adds per 64-bit Shoup product; see DESIGN.md section 6"}
    io = None
    if args.include_io and world == 1:
        job.io_setup()
        job.step_io()  # warm-up: first touch of the bounce paths
        runs = [job.step_io() for _ in range(max(1, args.steps))]
        best = min(runs, key=lambda r: r["total_s"])
        gb_in = job.cols * 2 * job.L * job.N * 8 / 1e9
        gb_out = (2 * (job.col_hi - job.col_lo) + job.queries) * 4 * job.N * 8 / 1e9
        io = {"io_inclusive_s": round(best["total_s"], 4), "upload_s": round(best["upload_s"], 4),
              "download_tail_s": round(best["download_tail_s"], 4), "upload_GB": round(gb_in, 2),
              "download_GB": round(gb_out, 2), "upload_GBps": round(gb_in / best["upload_s"], 1),
              "staging": "pinned host buffers (lumen_host_alloc); MatR downloads on a clone context's stream "
                         "under the MatZ inner product; input upload not overlapped (Encode needs all columns)"}
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
                       "parallelism": (f"{world} GPUs: lane-sharded Encode between two xGMI all-to-alls, columns sharded "
                                       "elsewhere, digest all-gather + Merkle root on device buffers"
                                       if job.lane_path else f"columns sharded over {world} GPU(s); digest all-gather"),
                       "baseline_ref": "BASELINE.md: reference Go/Lattigo CPU, m7i.8xlarge 32 vCPU"},
            # limb transforms the device EXECUTES per step (sum of the NTT kernels' units: the rescale to level 1
            # runs on coefficients, 14 transforms per polynomial instead of the reference's 75) ...
            "limb_ntts_executed_per_s": round(executed * world / sec_per_step, 1) if executed else None,
            # ... and the reference's own transform count for the same step (SURVEY 8d census) over the same time
            "limb_ntts_reference_equiv_per_s": round(census / sec_per_step, 1),
            "ct_ntts_reference_equiv_per_s": round(census / sec_per_step / (2 * job.L), 1),
            "roofline": roofline,
            "kernels": stages,
        }
        if io:
            out.update(io)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.config)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
