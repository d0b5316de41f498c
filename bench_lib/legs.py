"""What the default bench line carries beside `value`: the CPU baseline (the oracle, a reported figure), the plain
transform's rate, the I/O-inclusive legs, the Go shim's staging copy, the client's plain prover, the other BASELINE
configurations."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from lumenos_amd import params as lp  # noqa: E402
from .job import CONFIGS, PUBLISHED_SECONDS, RHO_INV, SECURITY_BITS, Job  # noqa: E402


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
        # the same leg in the driver's earlier lines at 16384x4096 (BENCH_r04 / r05.json): the checker got division-free butterflies in
        # round 5 and has not been touched since -- what moves from box to box now is the host CPU
        "previous_rounds_s": {"r04": 1643.0, "r05": 1478.0} if cfg == "16384x4096" else None,
    }


def hashlib_sha(arr):
    import hashlib
    return hashlib.sha256(memoryview(arr)).hexdigest()


def timed_steps(job, dist, steps, warmup, barrier):
    for _ in range(warmup):
        job.step(dist)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        job.step(dist)
    barrier()
    return (time.perf_counter() - t0) / steps


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
