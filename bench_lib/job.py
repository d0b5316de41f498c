"""The prover job of bench.py: device-resident inputs of one run and the step functions (one GPU, the library's group,
the torch.distributed exchange, the I/O-inclusive orders)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
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
        from .multi import all_gather_root, all_to_all_sets
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
            from .multi import all_gather_digests
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
