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

    def __init__(self, cfg, rank, world, device, ring_switch_logn=0, allow_replicated=False, local_devices=None):
        """local_devices: None = this process is ONE rank (`rank`, on `device`); a list of `world` device ordinals =
        this process owns ALL ranks (--single-process), rank i on local_devices[i] (the same ordinal repeated:
        several ranks share that GPU as clones of one context)."""
        from lumenos_amd.hip import Context
        self.rows, self.cols, self.log_n = CONFIGS[cfg]
        self.rank, self.world = rank, world
        P = lp.generate_bgv_params_for_ntt(self.cols, self.log_n)
        self.P = P
        self.L, self.K, self.N = len(P.q), len(P.p), P.N
        self.S = self.cols * RHO_INV
        self.queries = lp.calculate_queries(SECURITY_BITS, RHO_INV)
        self.group = None
        self.ctx_device, self.local_devices = (local_devices[0] if local_devices else device), local_devices
        self.local_ranks = list(range(world)) if local_devices else [rank]
        devices = list(local_devices) if local_devices else [device]
        # one context per device; further ranks on a device are clones (they share its tables and keys)
        by_device, self.ctxs = {}, []
        for d in devices:
            if d in by_device:
                self.ctxs.append(by_device[d].clone())
            else:
                by_device[d] = c = Context(P.log_n, P.q, P.p, P.psi, P.T, device=d)
                c.field_set(np.array(lp.field_roots_forward(P.T, self.S), dtype=np.uint64))
                self.ctxs.append(c)
        self._key_ctxs = list(by_device.values())
        self.ctx = ctx = self.ctxs[0]
        rng = np.random.default_rng(1)
        # lane-sharded Encode needs a power-of-two world whose lane shards keep at least one tile
        self.lane_path = world > 1 and (world & (world - 1)) == 0 and self.cols % world == 0 and (self.N // world) >= 64
        self.logw = world.bit_length() - 1 if self.lane_path else 0
        if world > 1 and not self.lane_path and not allow_replicated:
            raise SystemExit(f"bench.py: {world} ranks cannot run the lane-sharded path for {cfg} (needs a power-of-two "
                             f"world dividing cols = {self.cols} with N/world >= 64 lanes); --allow-replicated runs the "
                             "round-1 path instead (every rank holds the whole input and repeats the mixing passes)")
        # synthetic inputs: uniform residues (kernels are data-independent); with the lane path a
        # rank only ever holds its own block of columns
        own = self.cols // world if self.lane_path else self.cols
        self.matrices = [c.new_set(own, self.L).fill_random(1 + (r if self.lane_path else 0))
                         for c, r in zip(self.ctxs, self.local_ranks)]
        self.matrix = self.matrices[0]

        def rand_limbs(mods, shape_tail):
            out = np.empty((len(mods),) + shape_tail, dtype=np.uint64)
            for i, m in enumerate(mods):
                out[i] = rng.integers(0, m, size=shape_tail, dtype=np.uint64)
            return out

        self.zero_ct = np.ascontiguousarray(rand_limbs(P.q, (2, self.N)).transpose(1, 0, 2))
        self.r_pt = rand_limbs(P.q, (self.N,))
        self.b_pt = rand_limbs(P.q, (self.N,))
        beta = (self.L + self.K - 1) // self.K
        self.key_load_s, self.key_load_bytes = 0.0, 0
        for g in ctx.inner_sum_galois_elements(self.rows):
            evk = np.ascontiguousarray(
                rand_limbs(P.q + P.p, (beta, 2, self.N)).transpose(1, 2, 0, 3))  # [beta][2][L+K][N]
            for c in self._key_ctxs:
                t0 = time.perf_counter()
                c.load_galois_key(g, evk)  # (returns when the key is usable: conversion on the device)
                self.key_load_s += time.perf_counter() - t0
                self.key_load_bytes += evk.nbytes
        self.query_idx = rng.integers(0, self.S, size=self.queries).astype(np.uint32)
        self.ring_switch_logn = 0
        self._rand_limbs = rand_limbs
        # column shards (input columns; encoded columns are sharded by the transform itself)
        self.col_lo, self.col_hi = self.cols * rank // world, self.cols * (rank + 1) // world
        if ring_switch_logn:
            self.enable_ring_switch(ring_switch_logn)
        if self.lane_path:  # self.matrix IS the rank's block; its slice of the one Enc(0) for the lane Encode
            nw = self.N // world
            self.zero_lanes = np.ascontiguousarray(self.zero_ct[:, :, rank * nw:(rank + 1) * nw])
        for c in self.ctxs:
            c.sync()

    def enable_ring_switch(self, logn):
        """BASELINE config 5: RingSwitchNew on MatR / MatZ (ligero.go:336-342).  The key is the whole evaluation
        key a client posts (cmd/client/main.go:124-131), [rns][pw2][2][L+K][N]; with two special primes that is
        one Galois key's size (no power-of-two digits)."""
        P = self.P
        rns, pw2 = self.ctx.ringswitch_key_shape(13)[:2]
        key = np.ascontiguousarray(self._rand_limbs(P.q + P.p, (rns, pw2, 2, self.N)).transpose(1, 2, 3, 0, 4))
        for c in self._key_ctxs:
            c.load_ringswitch_key(logn, key)
        for c in self.ctxs:
            c._rs_logn = logn
        self.ring_switch_logn = logn
        from lumenos_amd.hip import pinned_empty
        own = self.matrix.count if self.lane_path else self.col_hi - self.col_lo
        # MatR / MatZ as they leave for the proof, one pair per local rank
        self.h_rs_all = [[pinned_empty((own, 2, 1 << logn)) for _ in range(2)] for _ in self.ctxs]
        self.h_rs = self.h_rs_all[0]

    def close(self):
        for a in ("io_ctx", "up_ctx"):
            c = getattr(self, a, None)
            if c is not None:
                c.close()
        if self.group is not None:
            self.group.close()
        for m in self.matrices:
            m.free()
        for c in self.ctxs[::-1]:
            c.close()

    # ---- host I/O of a prover run: the io leg of the default run
    def io_setup(self):
        """Page-locked host buffers (lumen_host_alloc): the input ciphertexts as the Go shim's stage() lays them
        out, and the proof's wire image -- metadata | MatR | MatZ | QueriedCols | paths | root
        (EncryptedProof.WriteTo, fhe/ligero.go:659-705) -- which the device assembles and DMAs into place; a clone
        context whose stream carries that while the main context computes."""
        from lumenos_amd.hip import pinned_bytes, pinned_empty
        self.h_matrix = pinned_empty((self.cols, 2, self.L, self.N))
        self.matrix.download_into(self.h_matrix)  # content: the synthetic matrix itself
        # the reference's framing as the C++ mirror restates it: a 281-byte MetaData block + LE64 length words
        head = bytes(281) + (2).to_bytes(8, "little")
        self.ctx.leaf_format_set(head, (2).to_bytes(8, "little"), self.N.to_bytes(8, "little"))
        self.ct1 = self.ctx.ct_serialized_size(2)
        depth = (self.S - 1).bit_length()
        self.wire_len = 11 + (2 * self.cols + self.queries) * self.ct1 + self.queries * depth * 32 + 32
        self.wire = pinned_bytes(self.wire_len)
        self.io_ctx = self.ctx.clone()
        self.up_ctx = self.ctx.clone()

    def encrypt_matrix(self):
        """Server-side witness encryption (cmd/server/main.go:188-208): `cols` columns of `rows` slot
        values from page-locked host memory -> Encoder.Encode + EncryptNew under pk on the device
        (lumen_encrypt_values), result resident in HBM as Commit's input.  Seconds, best of 3."""
        from lumenos_amd.hip import pinned_empty
        ctx, P = self.ctx, self.P
        rng = np.random.default_rng(5)
        pk = np.stack([np.stack([rng.integers(0, q, size=self.N, dtype=np.uint64) for q in P.q + P.p]) for _ in range(2)])
        ctx.load_public_key(pk)
        ctx.encoder_set(lp.encoder_psi(P.T, P.log_n))
        vals = pinned_empty((self.cols, self.rows))
        vals[:] = rng.integers(0, P.T, size=(self.cols, self.rows), dtype=np.uint64)
        seed = np.arange(32, dtype=np.uint8)
        best = None
        for _ in range(4):
            ctx.sync()
            t0 = time.perf_counter()
            s_ = ctx.encrypt_values(vals, seed, 0)
            ctx.sync()
            dt = time.perf_counter() - t0
            s_.free()
            best = dt if best is None else min(best, dt)
        return best

    def _marshal_tail(self, off, nodes, root):
        """Merkle paths + root behind the ciphertexts (ligero.go:694-700): host bytes, 309 x depth x 32"""
        depth = (self.S - 1).bit_length()
        lvl_off, n, paths = 0, self.S, np.empty((self.queries, depth, 32), dtype=np.uint8)
        idx = self.query_idx.astype(np.int64).copy()
        for d in range(depth):
            sib = np.minimum(idx ^ 1, n - 1)  # an unpaired last node is its own sibling (core/tree.go:127-131)
            paths[:, d] = nodes[lvl_off + sib]
            lvl_off, n, idx = lvl_off + n, (n + 1) // 2, idx >> 1
        self.wire[off:off + paths.size] = paths.reshape(-1)
        self.wire[off + paths.size:off + paths.size + 32] = np.frombuffer(root, dtype=np.uint8)
        return off + paths.size + 32

    def marshal(self, mat_r, mat_z, q, nodes, root):
        """EncryptedProof.MarshalBinary of results that sit in HBM: the three slices' wire images assembled on the
        device, one DMA each into the page-locked image; returns seconds (the reference's "Marshal proof" span,
        cmd/server/main.go:244-250: 2.3 s at 16384 x 4096)."""
        ctx = self.ctx
        ctx.sync()
        t0 = time.perf_counter()
        self.wire[:11] = np.frombuffer(np.array([self.rows, self.cols], "<u4").tobytes() + bytes([RHO_INV])
                                       + np.array([self.queries], "<u2").tobytes(), dtype=np.uint8)
        off = 11
        for s_ in (mat_r, mat_z, q):
            off += ctx.ct_serialize_into(s_, self.wire, offset=off, wait=False)
        off = self._marshal_tail(off, nodes, root)
        ctx.sync()
        assert off == self.wire_len
        return time.perf_counter() - t0

    def unmarshal(self):
        """EncryptedProof.UnmarshalBinary on the client's side of the wire (ligero.go:654-753): the image of the three
        slices from page-locked memory back into HBM sets, taken apart on the device; returns seconds and checks
        that the bytes come back as the residues they were made from."""
        ctx = self.ctx
        ctx.sync()
        t0 = time.perf_counter()
        off, sets = 11, []
        for count in (self.cols, self.cols, self.queries):
            n = count * self.ct1
            sets.append(ctx.ct_deserialize(self.wire[off:off + n], count, 2))
            off += n
        ctx.sync()
        dt = time.perf_counter() - t0
        return dt, sets

    def step_io(self, slices=8):
        """One step that starts with the input ciphertexts in (page-locked) host memory and ends with the proof's
        wire-format bytes there: upload (12.9 GB at D: one DMA, not overlappable in the fhe API's order -- Encode
        needs every column), the step, and the marshalling overlapped with it: MatR and MatZ are computed in
        column slices, each slice's wire image is assembled and DMA'd by the clone context behind the kernels
        that produce it (lumen_ctx_wait: no host block) while the main context goes on."""
        ctx, io = self.ctx, self.io_ctx
        t = {}
        t0 = time.perf_counter()
        self.matrix.upload(self.h_matrix)
        t["upload_s"] = time.perf_counter() - t0
        mine = ctx.encode(self.matrix, self.zero_ct, RHO_INV)
        lvl1 = ctx.rescale(mine, 2)
        mine.free()
        ctx.leaf_digests_begin(lvl1)
        self.wire[:11] = np.frombuffer(np.array([self.rows, self.cols], "<u4").tobytes() + bytes([RHO_INV])
                                       + np.array([self.queries], "<u2").tobytes(), dtype=np.uint8)
        keep, off = [], 11
        per = (self.cols + slices - 1) // slices
        for pt in (self.r_pt, self.b_pt):
            for c0 in range(0, self.cols, per):
                cols = self.matrix.slice(c0, min(per, self.cols - c0))
                part = ctx.matrix_inner_sum(cols, pt, self.rows)
                io.wait_for(ctx)
                off += io.ct_serialize_into(part, self.wire, offset=off, wait=False)
                keep += [cols, part]
        q = ctx.gather(lvl1, self.query_idx)
        dig = ctx.leaf_digests_end()
        nodes, root = ctx.merkle_build(dig)
        ctx.sync()
        t1 = time.perf_counter()
        off += ctx.ct_serialize_into(q, self.wire, offset=off, wait=False)
        off = self._marshal_tail(off, nodes, root)
        ctx.sync()
        io.sync()
        assert off == self.wire_len
        t["marshal_tail_s"] = time.perf_counter() - t1
        t["total_s"] = time.perf_counter() - t0
        for s_ in keep[::-1] + [q, lvl1]:
            s_.free()
        return t

    def step_io_fused(self, slices=8):
        """The same job in the order a server that owns the whole request can use (cmd/server/main.go:187-250 calls
        Commit and Prove back to back, and Prove's challenges do not depend on the Merkle root,
        fhe/ligero.go:198-199): the input arrives in column slices on a clone's stream and the inner products of a
        slice start as soon as it is resident; Encode (which needs every column) runs when the last slice has
        landed, its leaf hashing under the remaining inner products.  The upload disappears behind compute.
        Same kernels, same results, same wire bytes as step_io."""
        import threading
        ctx, io, up = self.ctx, self.io_ctx, self.up_ctx
        t = {}
        per = (self.cols + slices - 1) // slices
        starts = list(range(0, self.cols, per))
        arrived = [threading.Event() for _ in starts]
        t0 = time.perf_counter()

        def feeder():
            for k, c0 in enumerate(starts):
                up.upload_into(self.matrix, self.h_matrix[c0:c0 + per], first=c0)  # returns when the slice is in HBM
                arrived[k].set()
            t["upload_s"] = time.perf_counter() - t0

        th = threading.Thread(target=feeder)
        th.start()
        self.wire[:11] = np.frombuffer(np.array([self.rows, self.cols], "<u4").tobytes() + bytes([RHO_INV])
                                       + np.array([self.queries], "<u2").tobytes(), dtype=np.uint8)
        keep, lvl1 = [], None
        for k, c0 in enumerate(starts):
            arrived[k].wait()
            if k == len(starts) - 1:  # every column is resident: Commit's Encode + rescale, leaves hashed on the side
                mine = ctx.encode(self.matrix, self.zero_ct, RHO_INV)
                lvl1 = ctx.rescale(mine, 2)
                mine.free()
                ctx.leaf_digests_begin(lvl1)
            n = min(per, self.cols - c0)
            cols = self.matrix.slice(c0, n)
            for w, pt in enumerate((self.r_pt, self.b_pt)):
                part = ctx.matrix_inner_sum(cols, pt, self.rows)
                io.wait_for(ctx)
                io.ct_serialize_into(part, self.wire, offset=11 + (w * self.cols + c0) * self.ct1, wait=False)
                keep.append(part)
            keep.append(cols)
        th.join()
        off = 11 + 2 * self.cols * self.ct1
        q = ctx.gather(lvl1, self.query_idx)
        dig = ctx.leaf_digests_end()
        nodes, root = ctx.merkle_build(dig)
        ctx.sync()
        t1 = time.perf_counter()
        off += ctx.ct_serialize_into(q, self.wire, offset=off, wait=False)
        off = self._marshal_tail(off, nodes, root)
        ctx.sync()
        io.sync()
        assert off == self.wire_len
        t["marshal_tail_s"] = time.perf_counter() - t1
        t["total_s"] = time.perf_counter() - t0
        for s_ in keep[::-1] + [q, lvl1]:
            s_.free()
        return t

    def step_lanes(self, dist, timers=None, keep=False):
        """One step on `world` ranks with the lane-sharded Encode (module docstring), the exchange through
        torch.distributed on tensors aliasing the library's memory: the round-3 path, kept as --transport torch and as
        the fallback when the library's own RCCL group cannot be set up.  timers: per-stage wall seconds of this
        rank (every collective here ends with a device synchronisation anyway)."""
        ctx, W, rank = self.ctx, self.world, self.rank
        pg = getattr(self, "nccl_pg", None)  # the fallback's RCCL process group (the default one is the control plane)
        Sw = self.S // W

        def lap(name, t0):
            if timers is not None:
                ctx.sync()
                timers[name] = timers.get(name, 0.0) + time.perf_counter() - t0
            return time.perf_counter()

        t0 = time.perf_counter()
        # ---- Commit: Encode.  own columns -> lane blocks -> all-to-all -> lane shard of ALL columns
        blocks = ctx.lanes_split(self.matrix, self.logw)
        lanes = ctx.new_set_lanes(self.cols, self.L, self.logw)
        t0 = lap("lanes_split_s", t0)
        all_to_all_sets(dist, blocks, lanes, W, pg)
        t0 = lap("all_to_all_1_s", t0)
        blocks.free()
        enc = ctx.encode(lanes, self.zero_lanes, RHO_INV)  # this rank's lanes of all S encoded columns
        lanes.free()
        recv = ctx.new_set_lanes(self.S, self.L, self.logw)
        t0 = lap("encode_lane_shard_s", t0)
        all_to_all_sets(dist, enc, recv, W, pg)              # block h of every shard -> rank h
        t0 = lap("all_to_all_2_s", t0)
        enc.free()
        mine = ctx.lanes_assemble(recv)                      # whole ciphertexts of columns [rank*Sw, (rank+1)*Sw)
        recv.free()
        # ---- Commit: leaves on this rank's encoded columns, hashed under the inner products
        lvl1 = ctx.rescale(mine, 2)
        if not keep:
            mine.free()
        ctx.leaf_digests_begin(lvl1)
        t0 = lap("rescale_s", t0)
        # ---- Prove: inner products on this rank's input columns
        mat_r = ctx.matrix_inner_sum(self.matrix, self.r_pt, self.rows)
        t0 = lap("inner_product_r_s", t0)
        mat_z = ctx.matrix_inner_sum(self.matrix, self.b_pt, self.rows)
        t0 = lap("inner_product_b_s", t0)
        if self.ring_switch_logn:
            ctx.ring_switch(mat_r, self.h_rs[0])
            ctx.ring_switch(mat_z, self.h_rs[1])
        own = self.query_idx[(self.query_idx >= rank * Sw) & (self.query_idx < (rank + 1) * Sw)] - rank * Sw
        q = ctx.gather(lvl1, own.astype(np.uint32))
        t0 = lap("query_gather_local_s", t0)
        # ---- Commit, concluded: all-gather of the digests on device buffers, Merkle root on the device
        ptr, n = ctx.leaf_digests_end_device()
        root = all_gather_root(dist, ctx, ptr, n, self.S, W, pg)
        t0 = lap("digest_all_gather_and_root_s", t0)
        ctx.sync()
        if keep:
            return [mine], [lvl1], [mat_r], [mat_z], None, root
        for s in (q, mat_r, mat_z, lvl1):
            s.free()
        return root

    def step_group(self, timers=None, keep=False):
        """One step with the exchange inside the library (lumen_group_*): this process's local ranks -- all of them
        (--single-process) or one (a rank of torch.distributed.run) -- enqueue their stages, the group's
        collectives order them against each other on the devices.  timers: a dict that receives per-stage wall
        seconds, each stage drained before the next starts (the diagnostic pass; the timed steps never sync
        between stages)."""
        g, ctxs, W = self.group, self.ctxs, self.world

        def lap(name, t0):
            if timers is not None:
                g.sync()
                timers[name] = timers.get(name, 0.0) + time.perf_counter() - t0
            return time.perf_counter()

        t0 = time.perf_counter()
        # ---- Commit: Encode between the two all-to-alls (lumen_group_encode), leaves hashed on the side streams
        enc = g.encode(self.matrices, self.zero_ct, RHO_INV)
        t0 = lap("encode_with_both_all_to_alls_s", t0)
        lvl1 = [c.rescale(e, 2) for c, e in zip(ctxs, enc)]
        for c, l in zip(ctxs, lvl1):
            c.leaf_digests_begin(l)
        t0 = lap("rescale_s", t0)
        # ---- Prove: inner products on every rank's own input columns
        mat_r = [c.matrix_inner_sum(m, self.r_pt, self.rows) for c, m in zip(ctxs, self.matrices)]
        t0 = lap("inner_product_r_s", t0)
        mat_z = [c.matrix_inner_sum(m, self.b_pt, self.rows) for c, m in zip(ctxs, self.matrices)]
        t0 = lap("inner_product_b_s", t0)
        if self.ring_switch_logn:
            for c, a, b, h in zip(ctxs, mat_r, mat_z, self.h_rs_all):
                c.ring_switch(a, h[0])
                c.ring_switch(b, h[1])
            t0 = lap("ring_switch_s", t0)
        # ---- Prove: the queried columns, collected on rank 0 in query order
        q = g.gather(lvl1, self.query_idx)
        t0 = lap("query_gather_to_root_s", t0)
        # ---- Commit, concluded: ONE all-gather of the digests, Merkle root on the device
        g.all_gather_digests()
        root = g.merkle_root()
        t0 = lap("digest_all_gather_and_root_s", t0)
        g.sync()
        if keep:
            return enc, lvl1, mat_r, mat_z, q, root
        for s in [q] + mat_r + mat_z + lvl1 + enc:
            if s is not None:
                s.free()
        return root

    def step(self, dist=None, keep=False):
        if self.group is not None:
            return self.step_group()
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
            ctx.ring_switch(mat_r, self.h_rs[0])
            ctx.ring_switch(mat_z, self.h_rs[1])
        # ---- Prove: query columns (fhe/ligero.go:261-280): already at level 1 from Commit
        q = ctx.gather(lvl1, owned_queries(self.query_idx, my_cols))
        # ---- Commit, concluded: digests -> (all-gather) -> Merkle tree (core/tree.go:113-163)
        dig = ctx.leaf_digests_end()
        if dist is not None and self.world > 1:
            dig = all_gather_digests(dist, dig, my_cols, self.S, self.world)
        nodes, root = ctx.merkle_build(dig)
        ctx.sync()
        if keep:
            lvl1.free()
            return mat_r, mat_z, q, nodes, root
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
    n_c = 4 * cores  # (about 5 s of work on 16 cores: the whole sample is 10-20 s)
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
    n_i = 2 * cores
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


def hashlib_sha(arr):
    import hashlib
    return hashlib.sha256(memoryview(arr)).hexdigest()


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
           f"--nproc-per-node={args.gpus}", os.path.abspath(__file__), *argv]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:  # rank 0's JSON line goes to stdout; whatever a library chats there (gloo) to stderr
        out = sys.stdout if line.lstrip().startswith("{") else sys.stderr
        out.write(line)
        out.flush()
    return proc.wait()


def timed_steps(job, dist, steps, warmup, barrier):
    for _ in range(warmup):
        job.step(dist)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        job.step(dist)
    barrier()
    return (time.perf_counter() - t0) / steps


def profile_kernels(job, dist, cfg):
    """Dominant-kernel roofline: one more (untimed) step with HIP events around every launch on the contexts'
    streams.  Returns (roofline, per-kernel table, limb transforms executed in the step) of the first local rank
    and the same triple for every local rank."""
    for c in job.ctxs:
        c.prof_reset()
        c.prof_enable(True)
        # one key-switch lane for the measured step: with two (the default below N = 2^14) a kernel's event pair also
        # spans whatever its neighbour on the other stream was doing
        c.set_tuning("LUMEN_KS_LANES", 1)
    job.step(dist)
    for c in job.ctxs:
        c.prof_enable(False)
        c.set_tuning("LUMEN_KS_LANES", 0)  # back to the default by ring degree
    per_rank = [_kernel_table(job, c, cfg) for c in job.ctxs]
    return per_rank[0] + (per_rank,)


def _kernel_table(job, ctx, cfg):
    tab = {k: ctx.prof_read(k) for k in ctx.prof_names()}
    pmc = pmc_table(cfg)
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
    roofline = None
    if ntt_kernels:
        dom = max(ntt_kernels, key=lambda k: ntt_kernels[k][0])
        ms, launches, units = ntt_kernels[dom]
        alg_bytes_per_launch = 16.0 * job.N * units / launches  # 16*N B per limb transform
        achieved = alg_bytes_per_launch / (ms / launches * 1e-3) / 1e9
        pe = pmc_entry(pmc, dom)
        sq = (pe or {}).get("sq_per_launch") or {}
        roofline = {"bound": "hbm", "limiter": "valu and memory phases in series", "kernel": dom, "achieved": round(achieved, 1), "peak": 8000.0,
                    "unit": "GB/s", "frac": round(achieved / 8000.0, 4),
                    "traffic": round(pe["hbm_bytes_per_launch"]) if pe and pe.get("hbm_bytes_per_launch") else None,
                    "avg_launch_ms": round(ms / launches, 4), "limb_ntts_per_launch": units // launches,
                    "valu": valu_roof(job, ms, launches, units, sq),
                    "note": "`bound` names the roofline `frac` is priced against (HBM, as SURVEY 8d prescribes for every "
                            "kernel of this path).  `limiter`: neither roof is saturated -- the butterfly-only VALU ceiling is 0.61 "
                            "of the HBM peak (`valu`, calibrated on this chip: 10 multiply-adds per 64-bit Shoup product), the "
                            "memory side alone (the kernels built without butterflies, profiles/r05_exp_no_butterflies_floor.txt) 0.66 "
                            "for a plain transform and 0.52 for this kernel with its fused basis extension, and in a wave's life "
                            "the two run in series more than they overlap (at N = 2^12 .. 2^14, 1 to 4 resident workgroups per CU): 0.35.  The >= 50 % HBM target of "
                            "north_star is out of reach on both counts; "
                            "DESIGN.md section 6 (and profiles/EXPERIMENTS.md) has the costing"}
    return roofline, stages, executed


# The VALU roof of the transform kernels, calibrated on the MI355X itself (not "4 cycles per instruction"):
# tools/ubench_bfly.hip runs the forward butterfly stages alone -- registers only, no LDS, no global memory, the
# product's hand-scheduled 15-instruction butterfly (10 v_mad_u64_u32 + 5) -- and needs 30.6-32.3 ns per
# wave-butterfly per SIMD at the 4 waves per SIMD the N = 2^14 kernels run with (profiles/r02_ubench_butterfly.txt:
# 73-78 cycles at 2.4 GHz; profiles/r04_ubench_fold.txt measures 71 / 70 / 78 at 4 / 2 / 1 waves).  The best of
# those is the ceiling: a limb transform is N/2 * log2 N / 64 wave-butterflies, the chip has 256 CUs x 4 SIMDs.
# Per instruction class (tools/ubench_valu.hip, profiles/r02_ubench_valu.txt, cycles per wave-instruction per SIMD):
# v_mad_u64_u32 5.4-5.6, other 64-bit / carry / full-rate-multiply forms 4.3-5.0, plain 32-bit ALU 2.4-2.9 -- the
# butterfly's own mix averages 30.6 ns / 15 = 2.04 ns = 4.9 cycles, which is the price put on every VALU instruction
# the SQ counters saw (`issue_frac`); the flat 4 cycles the counters' own "busy" figure assumes under-reads it.
BFLY_NS_PER_WAVE_PER_SIMD = 30.6
BFLY_INSTS = 15
N_SIMD = 256 * 4


def valu_roof(job, ms, launches, units, sq):
    """roofline.valu: the butterfly-only ceiling in limb transforms per second, what the dominant kernel achieves
    against it, and (from the committed SQ counters of this very build, else null) the fraction of the chip's VALU
    issue time its instructions account for at the calibrated price."""
    wave_bfly = job.N // 2 * job.log_n / 64.0                      # wave-butterflies of one limb transform
    ceiling = N_SIMD / (wave_bfly * BFLY_NS_PER_WAVE_PER_SIMD * 1e-9)
    got = units / (ms * 1e-3)
    out = {"ceiling_limb_ntts_per_s": round(ceiling), "achieved_limb_ntts_per_s": round(got),
           "frac": round(got / ceiling, 4),
           "calibration": {"ns_per_wave_butterfly_per_simd": BFLY_NS_PER_WAVE_PER_SIMD, "insts_per_butterfly": BFLY_INSTS,
                           "simds": N_SIMD, "source": "tools/ubench_bfly.hip, tools/ubench_valu.hip -> "
                                                      "profiles/r02_ubench_butterfly.txt, r02_ubench_valu.txt, r04_ubench_fold.txt"},
           "insts_per_butterfly": None, "issue_frac": None, "issue_frac_at_flat_4_cycles": None}
    if sq.get("SQ_INSTS_VALU"):
        insts = sq["SQ_INSTS_VALU"]                                  # wave-level VALU instructions of one launch
        bfly_waves = units / launches * wave_bfly
        launch_s = ms / launches * 1e-3
        out["insts_per_butterfly"] = round(insts / bfly_waves, 2)
        out["issue_frac"] = round(insts * (BFLY_NS_PER_WAVE_PER_SIMD / BFLY_INSTS) * 1e-9 / (launch_s * N_SIMD), 4)
        out["issue_frac_at_flat_4_cycles"] = round(insts * 4 / 2.4e9 / (launch_s * N_SIMD), 4)
    return out


def plain_ntt_rates(job):
    """The plain limb transform north_star names (k_limb_ntt, no fused load/store work): steady-state rate on the
    resident input matrix (98 304 transforms per pass at D; the set is transformed and transformed back, so the
    residues are left as they were)."""
    reps, n_tr = 12, job.matrix.count * 2 * job.L
    rates = {}
    job.ctx.set_ntt(job.matrix, False)
    job.ctx.set_ntt(job.matrix, True)  # warm; a bit-exact round trip
    for name, inv in (("forward", False), ("inverse", True)):
        total_ms = 0.0
        for _ in range(reps):  # untimed passes of the other direction in between restore the data
            if inv:
                job.ctx.set_ntt(job.matrix, False)
            job.ctx.timer_start()
            job.ctx.set_ntt(job.matrix, inv)
            total_ms += job.ctx.timer_stop()
            if not inv:
                job.ctx.set_ntt(job.matrix, True)
        rates[name] = n_tr * reps / (total_ms * 1e-3)
    return {"kernel": "k_limb_ntt", "log_n": job.log_n, "limb_ntts_per_launch": n_tr,
            "forward_per_s": round(rates["forward"], 1), "inverse_per_s": round(rates["inverse"], 1),
            "forward_hbm_frac": round(rates["forward"] * 16.0 * job.N / 8e12, 4),
            "inverse_hbm_frac": round(rates["inverse"] * 16.0 * job.N / 8e12, 4),
            "ct_ntts_forward_per_s": round(rates["forward"] / (2 * job.L), 1)}


def io_leg(job, cfg):
    """What surrounds the metric on a real server and client, measured (never `value`): Marshal / Unmarshal of the
    proof, the client's Decrypt proof, the server's Encrypt matrix, and whole steps that start and end in host
    memory."""
    job.io_setup()
    outs = job.step(keep=True)
    job.marshal(*outs)  # first touch of the wire image and the staging paths
    marshal_s = min(job.marshal(*outs) for _ in range(3))
    unmarshal_s, back = job.unmarshal()
    for a_, b_ in zip(outs[:3], back):  # the round trip of ligero_test.go:118-126, on the device
        assert np.array_equal(a_.download(0, 2), b_.download(0, 2)) and np.array_equal(
            a_.download(a_.count - 1, 1), b_.download(b_.count - 1, 1)), "unmarshalled ciphertexts differ"
    # the client's "Decrypt proof" (EncryptedProof.Decrypt, ligero.go:381-502: slot 0 of every MatR / MatZ
    # ciphertext, all `rows` slots of the 309 opened columns; 48.05 s on the reference's client at this shape)
    rng_k = np.random.default_rng(6)
    job.ctx.load_secret_key(np.stack([rng_k.integers(0, q, size=job.N, dtype=np.uint64) for q in job.P.q]))
    job.ctx.encoder_set(lp.encoder_psi(job.P.T, job.P.log_n))
    job.ctx.decrypt(back[2], job.rows)
    decrypt_s = None
    for _ in range(3):
        job.ctx.sync()
        t0_ = time.perf_counter()
        job.ctx.decrypt(back[0], 1), job.ctx.decrypt(back[1], 1), job.ctx.decrypt(back[2], job.rows)
        dt_ = time.perf_counter() - t0_
        decrypt_s = dt_ if decrypt_s is None else min(decrypt_s, dt_)
    for b_ in back:
        b_.free()
    for s_ in outs[:3]:
        s_.free()
    job.step_io()  # warm-up
    runs = [job.step_io() for _ in range(2)]
    best = min(runs, key=lambda r: r["total_s"])
    want = hashlib_sha(job.wire)
    enc_s = job.encrypt_matrix()
    job.step_io_fused()
    fused = min([job.step_io_fused() for _ in range(2)], key=lambda r: r["total_s"])
    assert hashlib_sha(job.wire) == want, "the fused order produced different proof bytes"
    gb_in = job.cols * 2 * job.L * job.N * 8 / 1e9
    stage = stage_seconds(job)
    return {"marshal_s": round(marshal_s, 4), "unmarshal_s": round(unmarshal_s, 4),
          # client side of the wire, for a client that owns a GPU: unmarshal_s above + this = "Decrypt proof"
          "decrypt_proof_s": round(decrypt_s, 4),
          "io_inclusive_s": round(best["total_s"], 4),
          "io_inclusive_fused_order_s": round(fused["total_s"], 4),
          # what precedes the metric in the reference's server (cmd/server/main.go:188-208, "Encrypt matrix":
          # 66.84 s at 16384x4096): the raw witness columns from host memory, Encoder.Encode + EncryptNew on the device
          "encrypt_matrix_s": round(enc_s, 4),
          "io": {"stage_s": stage,
                 "upload_s": round(best["upload_s"], 4), "marshal_tail_s": round(best["marshal_tail_s"], 4),
                 "upload_GB": round(gb_in, 2), "upload_GBps": round(gb_in / best["upload_s"], 1),
                 "proof_wire_GB": round(job.wire_len / 1e9, 3),
                 "marshal_GBps": round(job.wire_len / 1e9 / marshal_s, 1),
                 "reference_marshal_s": {"16384x4096": 2.254, "8192x4096": 1.135, "4096x2048": 0.347,
                                         "2048x1024": 0.156}.get(cfg),  # results/baseline/server/bench_*.txt:34
                 "note": "marshal_s: EncryptedProof.MarshalBinary of results resident in HBM -- wire images of MatR, "
                         "MatZ and the queried columns assembled on the device (k_ct_wire), one DMA each into "
                         "page-locked memory, Merkle paths + root appended (the reference's 'Marshal proof' span). "
                         "io_inclusive_s: input ciphertexts from page-locked host memory (one DMA, not overlappable "
                         "in the fhe API's order: Encode needs every column), the step, and the same marshalling "
                         "overlapped with it on a clone context (column slices, lumen_ctx_wait); ends with the "
                         "proof's wire bytes in host memory. io_inclusive_fused_order_s: the same bytes (checked) in "
                         "the order a server that owns the whole request can use -- Prove's challenges do not depend "
                         "on the Merkle root (ligero.go:198-199), so the inner products of a column slice start when "
                         "it lands and Encode runs once the last one has: the upload hides behind compute"}}


def stage_seconds(job):
    """SURVEY K11, measured: what the Go shim's stage() costs at this shape.  Lattigo holds one separately allocated
    []uint64 per limb (ct.Value[k].Coeffs[i]): cols x 2 x L arrays of N words (98 304 arrays of 128 KB at 16384 x
    4096) that cgo cannot hand over as they are.  They are gathered into the flat page-locked buffer
    lumen_set_upload takes (lumen_host_gather: the shim pins the limbs and passes their addresses) with 1 host
    thread -- a single goroutine's copy() loop, INTEGRATION.md's stage() -- and with 16.  NOT part of
    io_inclusive_s / io_inclusive_fused_order_s, which start from the flat buffer: either add it, or build the
    ciphertexts over one lumen_host_alloc block (`newAliasedCiphertexts`), which makes the
    copy disappear."""
    from lumenos_amd.hip import host_gather
    n = job.cols * 2 * job.L
    limbs = []
    for _ in range(n):  # separately allocated, pages touched (a first-touch fault is not part of a copy)
        a = np.empty(job.N, dtype=np.uint64)
        a.fill(7)
        limbs.append(a)
    flat = job.h_matrix.reshape(-1)
    keep = flat[:8].copy()
    out = {"limb_arrays": n, "KB_each": job.N * 8 // 1024, "GB": round(n * job.N * 8 / 1e9, 2),
           "included_in_io_inclusive": False}
    for threads in (1, 16):
        host_gather(flat, limbs, threads)  # warm
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            host_gather(flat, limbs, threads)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out[f"threads_{threads}_s"] = round(best, 4)
        out[f"threads_{threads}_GBps"] = round(n * job.N * 8 / best / 1e9, 1)
    assert flat[0] == 7 and keep is not None
    del limbs
    job.matrix.download_into(job.h_matrix)  # the staging buffer holds the synthetic matrix again
    return out


def plain_prover_seconds(rows, cols, device):
    """LigeroProveReference (fhe/ligero.go:799-953) -- the plain prover the reference's CLIENT runs to check the
    decrypted proof ("Ligero local generation": 14 min 22 s at 16384 x 4096 on its 2 vCPUs,
    results/baseline/client/bench_16384x4096_14.txt:36-45) -- on the same kernels: a context whose one modulus is
    T holds the plain matrix column by column.  Witness columns from page-locked host memory,
    core.Encode of every row, leaf digests + Merkle tree, the two matrix-vector products, the opened columns."""
    from lumenos_amd.hip import Context, pinned_empty
    T = lp.T_REFERENCE
    log_n = (rows // 2).bit_length() - 1
    S = cols * RHO_INV
    ctx = Context(log_n, [T], [], [lp.encoder_psi(T, log_n)], T, device=device)
    ctx.field_set(np.array(lp.field_roots_forward(T, S), dtype=np.uint64))
    ctx.leaf_format_set(b"", b"", b"")  # a leaf is the column's bytes (ligero.go:866-872)
    rng = np.random.default_rng(8)
    host = pinned_empty((cols, 2, 1, rows // 2))
    host[:] = rng.integers(0, T, size=host.shape, dtype=np.uint64)
    zero = np.zeros((2, 1, rows // 2), dtype=np.uint64)
    r = rng.integers(0, 2**63, size=rows, dtype=np.uint64)
    b = rng.integers(0, T, size=rows, dtype=np.uint64)
    idx = rng.integers(0, S, size=lp.calculate_queries(SECURITY_BITS, RHO_INV)).astype(np.uint32)
    m = ctx.new_set(cols, 1)
    best = None
    for _ in range(3):
        ctx.sync()
        t0 = time.perf_counter()
        m.upload(host)
        enc = ctx.encode(m, zero, RHO_INV)
        dig = ctx.leaf_digests(enc)
        ctx.merkle_build(dig)
        ctx.plain_inner_products(m, r)
        ctx.plain_inner_products(m, b)
        ctx.gather(enc, idx).download()
        ctx.sync()
        dt = time.perf_counter() - t0
        enc.free()
        best = dt if best is None else min(best, dt)
    m.free()
    ctx.close()
    return best


def other_configs(job, args, sec_per_step, local_rank, barrier):
    """Short passes over the other BASELINE.json configurations, so that the driver's one command attests them."""
    others = {}
    if args.config == "16384x4096":  # BASELINE config 5 on the resident job: + RingSwitchNew -> LogN = 10
        job.enable_ring_switch(10)
        sec = timed_steps(job, None, args.other_steps, 1, barrier)
        others["16384x4096+ring-switch->LogN=10"] = {
            "value": round(sec, 4), "unit": "s", "steps": args.other_steps,
            "reference_s": 417.6, "ring_switch_added_s": round(sec - sec_per_step, 4)}
        job.ring_switch_logn = 0
    main_ctx = job.ctx
    for cfg in ("2048x1024", "4096x2048", "8192x4096"):
        if cfg == args.config:
            continue
        oj = Job(cfg, 0, 1, local_rank)
        sec = timed_steps(oj, None, args.other_steps, 1, lambda: oj.ctx.sync())
        others[cfg] = {"value": round(sec, 4), "unit": "s", "steps": args.other_steps,
                       "reference_s": PUBLISHED_SECONDS[cfg], "L": oj.L, "LogN": oj.log_n}
        oj.close()
    assert job.ctx is main_ctx
    # the client's plain prover on the same kernels, at the configuration's shape
    ref = {"16384x4096": 861.9, "8192x4096": None, "4096x2048": None, "2048x1024": 3.89}.get(args.config)
    others["plain_prover_" + args.config] = {"value": round(plain_prover_seconds(job.rows, job.cols, local_rank), 4),
                                             "unit": "s", "reference_client_s": ref}
    return others


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

    for _ in range(args.warmup):
        job.step(dist)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        job.step(dist)
    barrier()
    elapsed = time.perf_counter() - t0
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
