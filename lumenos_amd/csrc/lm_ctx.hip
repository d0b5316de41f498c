// Context, tables and HBM-resident ciphertext sets.
// Replaces what fhe.NewBackendBFV / ServerBFV.CopyNew capture on the Go side
// (fhe/bfv.go:13-58): ring degree, moduli chains and the per-modulus NTT
// tables (Lattigo SubRing.RootsForward/RootsBackward -- rebuilt here from the
// primitive 2N-th root the host passes, with Shoup companions instead of
// Montgomery form).
#include <cstring>
#include <thread>

#include "lm_ntt_dev.h"

thread_local std::string lm_global_err;

int lm_fail(lumen_ctx *ctx, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    lm_global_err = buf;
    fprintf(stderr, "[lumenos_hip] error: %s\n", buf); // vdec_wrapper.c:19-22 convention
    return 1;
}

void *lm_scratch(lumen_ctx *ctx, const char *name, size_t bytes) {
    auto &e = ctx->scratch[name];
    if (e.second >= bytes && e.first) return e.first;
    if (e.first) {
        lm_sync_all(ctx); // the old block may still be read on any of the context's streams
        hipFree(e.first);
        e.first = nullptr;
        e.second = 0;
    }
    void *p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {
        lm_fail(ctx, "hipMalloc(%zu) for scratch '%s' failed", bytes, name);
        return nullptr;
    }
    e.first = p;
    e.second = bytes;
    return p;
}

// puts a block the caller allocated (hipMalloc) under a scratch name; whatever was there is freed after the context's
// streams have drained.  For buffers whose PLACEMENT was chosen (lm_keyswitch.hip, select_placement).
void lm_scratch_adopt(lumen_ctx *ctx, const char *name, void *p, size_t bytes) {
    auto &e = ctx->scratch[name];
    if (e.first && e.first != p) {
        lm_sync_all(ctx);
        hipFree(e.first);
    }
    e.first = p;
    e.second = bytes;
}

lm_shared::~lm_shared() {
    hipFree(d_tw_fwd);
    hipFree(d_tw_inv);
    hipFree(d_scal);
    for (auto &kv : gkeys) {
        hipFree(kv.second.d_key);
        hipFree(kv.second.d_index);
        hipFree(kv.second.d_inv_index);
    }
    ext.clear();
}

void lm_sync_all(lumen_ctx *ctx) {
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    if (ctx->stream2) hipStreamSynchronize(ctx->stream2);
    if (ctx->stream_aux) hipStreamSynchronize(ctx->stream_aux);
}

void *lm_stage(lumen_ctx *ctx, size_t bytes) {
    if (ctx->ev_stage) hipEventSynchronize(ctx->ev_stage); // the previous copy out of the buffer
    if (ctx->stage_cap < bytes) {
        if (ctx->stage_host) hipHostFree(ctx->stage_host);
        ctx->stage_host = nullptr, ctx->stage_cap = 0;
        if (hipHostMalloc(&ctx->stage_host, bytes, hipHostMallocDefault) != hipSuccess) {
            lm_fail(ctx, "hipHostMalloc(%zu) for host staging failed", bytes);
            return nullptr;
        }
        ctx->stage_cap = bytes;
    }
    return ctx->stage_host;
}

hipEvent_t lm_ev_get(lumen_ctx *ctx) {
    if (!ctx->ev_pool.empty()) {
        hipEvent_t e = ctx->ev_pool.back();
        ctx->ev_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    hipEventCreate(&e);
    return e;
}

lm_prof_scope::lm_prof_scope(lumen_ctx *c, const char *n, uint64_t u) : ctx(c), name(n), units(u) {
    if (!ctx->prof) return;
    a = lm_ev_get(ctx);
    hipEventRecord(a, ctx->stream);
}
lm_prof_scope::~lm_prof_scope() {
    if (!ctx->prof || !a) return;
    hipEvent_t b = lm_ev_get(ctx);
    hipEventRecord(b, ctx->stream);
    ctx->prof_pending.push_back({name, a, b, units});
}

void lm_prof_resolve(lumen_ctx *ctx) {
    if (ctx->prof_pending.empty()) return;
    for (auto &p : ctx->prof_pending) {
        // a pair may sit on the main stream, the second lane or the side stream: wait for the pair
        // itself, and drop a measurement that cannot be read rather than log it as 0 ms
        float ms = 0;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            auto &e = ctx->prof_tab[p.name];
            e.total_ms += ms;
            e.launches += 1;
            e.units += p.units;
        } else {
            (void)hipGetLastError();
        }
        ctx->ev_pool.push_back(p.a);
        ctx->ev_pool.push_back(p.b);
    }
    ctx->prof_pending.clear();
}

lm_modmap lm_map_q(uint32_t nl) {
    lm_modmap m;
    m.period = nl;
    for (uint32_t i = 0; i < LM_MAX_LIMBS; i++) m.idx[i] = (uint8_t)(i < nl ? i : 0);
    return m;
}

mod_t lm_make_mod(uint64_t q) {
    mod_t m;
    m.q = q;
    m.qinv64 = (u64)((((u128)1) << 64) / q);
    // -q^-1 mod 2^64 by Newton iteration
    u64 inv = 1;
    for (int it = 0; it < 6; it++) inv *= 2 - (u64)q * inv;
    m.qneg = (u64)0 - inv;
    u64 r = (u64)((((u128)1) << 64) % q);
    m.r2 = h_mulmod(r, r, q);
    return m;
}

void lm_build_tw(uint64_t q, uint64_t psi, uint32_t logN, std::vector<tw_t> &fwd, std::vector<tw_t> &inv) {
    uint32_t N = 1u << logN;
    fwd.resize(N);
    inv.resize(N);
    uint64_t psi_inv = h_invmod(psi, q), cf = 1, cb = 1;
    for (uint32_t j = 0; j < N; j++) {
        uint32_t r = h_bitrev(j, (int)logN);
        fwd[r] = h_tw(cf, q);
        inv[r] = h_tw(cb, q);
        cf = h_mulmod(cf, psi, q);
        cb = h_mulmod(cb, psi_inv, q);
    }
}

// frees the pooled storage (the caller has waited for the context's streams)
static void pool_drain(lumen_ctx *ctx) {
    for (auto &kv : ctx->pool) {
        hipFree(kv.second.p);
        if (kv.second.ready) ctx->ev_pool.push_back(kv.second.ready);
    }
    ctx->pool.clear();
    ctx->pool_bytes = 0;
}

// streams, events and timers: what every context owns, a clone included
static int ctx_private_init(lumen_ctx *ctx) {
    LM_HIP(ctx, hipSetDevice(ctx->device));
    LM_HIP(ctx, hipStreamCreate(&ctx->stream));
    LM_HIP(ctx, hipStreamCreate(&ctx->stream2));
    LM_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    LM_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    LM_HIP(ctx, hipStreamCreate(&ctx->stream_aux));
    LM_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_aux, hipEventDisableTiming));
    LM_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_stage, hipEventDisableTiming));
    LM_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_xdep, hipEventDisableTiming));
    LM_HIP(ctx, hipEventCreate(&ctx->tm0));
    LM_HIP(ctx, hipEventCreate(&ctx->tm1));
    return 0;
}

// 0: set; 1: no such switch; 2: value out of range (nothing changed)
static int tuning_set(lm_tuning &t, const char *name, long v) {
    const std::string n(name ? name : "");
    auto in = [&](long lo, long hi) { return v >= lo && v <= hi; };
    if (n == "LUMEN_KS_BATCH") {
        if (!in(1, 4096)) return 2;
        t.ks_batch = (uint32_t)v;
    } else if (n == "LUMEN_KS_LANES") { // 0: the default by ring degree
        if (!in(0, 2)) return 2;
        t.ks_lanes = (uint32_t)v;
    } else if (n == "LUMEN_KS_FUSED_DIGITS") { // negative: the derived default
        if (v > LM_MAX_LIMBS) return 2;
        t.ks_fused_digits = v >= 0 ? (int32_t)v : -1;
    } else if (n == "LUMEN_DEBUG") {
        t.debug = v != 0;
    } else if (n == "LUMEN_MODUP_TGROUP") {
        if (!in(1, 31)) return 2;
        t.modup_tgroup = (uint32_t)v;
    } else if (n == "LUMEN_MODDOWN_TGROUP") {
        if (!in(1, 31)) return 2;
        t.moddown_tgroup = (uint32_t)v;
    } else if (n == "LUMEN_KS_PLACEMENT") {
        if (!in(0, 32)) return 2;
        t.ks_placement = (uint32_t)v;
    } else {
        return 1;
    }
    return 0;
}
static void tuning_from_env(lm_tuning &t) {
    for (const char *n : {"LUMEN_KS_BATCH", "LUMEN_KS_LANES", "LUMEN_KS_FUSED_DIGITS", "LUMEN_DEBUG",
                          "LUMEN_MODUP_TGROUP", "LUMEN_MODDOWN_TGROUP", "LUMEN_KS_PLACEMENT"}) {
        const char *e = getenv(n);
        // an empty override counts as unset; a value out of range is reported and leaves the default
        if (e && *e && tuning_set(t, n, atol(e)))
            fprintf(stderr, "[lumenos_hip] %s=%s is out of range: ignored\n", n, e);
    }
}

// A/B tools and tests: the same switches the environment sets at lumen_ctx_create, on a live context
// (value < 0 returns LUMEN_KS_FUSED_DIGITS to its derived default).  Not for production code paths.
extern "C" int lumen_ctx_set_tuning(lumen_ctx *ctx, const char *name, long value) {
    LM_CHECK(nullptr, ctx && name, "lumen_ctx_set_tuning: NULL argument");
    LM_ENTER(ctx);
    const int rc = tuning_set(ctx->tune, name, value);
    LM_CHECK(ctx, rc != 1, "unknown tuning switch %s", name);
    LM_CHECK(ctx, rc == 0, "tuning switch %s: value %ld is out of range (nothing changed)", name, value);
    return 0;
}

// TEST HOOK, not a tuning switch and never read from the environment: lets lumen_group_create pass LUMEN_TRANSPORT_RCCL
// (and lets LUMEN_TRANSPORT_AUTO choose it) although ranks share a device, so that the library's RCCL call sequence can
// run with W > 1 on a one-GPU box against the test double tests/cpp/fake_rccl.cpp.  Real RCCL refuses such a
// communicator itself.  Clones made afterwards inherit the setting.
extern "C" int lumen_test_allow_shared_device_rccl(lumen_ctx *ctx, int on) {
    LM_CHECK(nullptr, ctx, "lumen_test_allow_shared_device_rccl: NULL ctx");
    LM_ENTER(ctx);
    ctx->tune.rccl_shared_device = on != 0;
    return 0;
}

// ctx's stream waits for everything enqueued on other's stream so far -- without blocking the host.  What a
// pipeline of contexts needs: a clone serialises / downloads MatR behind the kernels that produce it while
// the producer context goes on to MatZ (fhe/ligero.go:231-242 runs the two on separate evaluator copies).
extern "C" int lumen_ctx_wait(lumen_ctx *ctx, lumen_ctx *other) {
    LM_CHECK(nullptr, ctx && other, "lumen_ctx_wait: NULL argument");
    if (ctx == other) return 0;
    hipEvent_t ev;
    {
        LM_ENTER(other);
        LM_HIP(other, hipEventRecord(other->ev_xdep, other->stream));
        ev = other->ev_xdep;
    }
    LM_ENTER(ctx);
    LM_CHECK(ctx, ctx->device == other->device, "lumen_ctx_wait: contexts on devices %d and %d", ctx->device, other->device);
    LM_HIP(ctx, hipStreamWaitEvent(ctx->stream, ev, 0));
    return 0;
}

extern "C" int lumen_ctx_create(const lumen_params_desc *desc, lumen_ctx **out) {
    if (!desc || !out) return lm_fail(nullptr, "lumen_ctx_create: NULL argument");
    *out = nullptr;
    LM_CHECK(nullptr, desc->abi_version == LUMEN_ABI_VERSION, "ABI version mismatch: got %u want %u",
             desc->abi_version, LUMEN_ABI_VERSION);
    LM_CHECK(nullptr, lm_logn_supported(desc->log_n),
             "log_n %u unsupported: kernels are instantiated for N = 2^8 and 2^10..2^14", desc->log_n);
    LM_CHECK(nullptr, desc->num_q >= 1 && desc->num_q + desc->num_p <= LM_MAX_LIMBS,
             "bad limb counts L=%u K=%u", desc->num_q, desc->num_p);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return lm_fail(nullptr, "no HIP device visible: the lumenos HIP path has no CPU fallback");
    LM_CHECK(nullptr, desc->device >= 0 && desc->device < ndev, "device %d out of range (have %d)",
             desc->device, ndev);
    std::unique_ptr<lumen_ctx, void (*)(lumen_ctx *)> guard(new lumen_ctx(std::make_shared<lm_shared>()),
                                                            lumen_ctx_destroy);
    lumen_ctx *ctx = guard.get();
    tuning_from_env(ctx->tune); // the only place the library reads its environment
    ctx->device = desc->device;
    ctx->logN = desc->log_n;
    ctx->N = 1u << desc->log_n;
    ctx->L = desc->num_q;
    ctx->K = desc->num_p;
    ctx->T = desc->plaintext_modulus;
    uint32_t LK = ctx->L + ctx->K, N = ctx->N;
    uint64_t two_n = 2ull * N;
    for (uint32_t i = 0; i < LK; i++) {
        uint64_t q = desc->moduli[i], psi = desc->psi[i];
        // the lazy forward NTT takes inputs below 7q (fused basis extension) and lets values grow
        // by 3q per stage before its one reduction
        // ... and the InnerSum accumulator is lazy in [0, 2q) across the rotations (lm_keyswitch.hip, k_moddown_ntt: acc < 2q
        // plus d < 6q must not wrap): 8q < 2^64, implied by the bound below for every log_n >= 0 and stated here because
        // the kernels' comments refer to it
        static_assert(3 * 0 + 8 >= 8, "the modulus bound (3 log_n + 8) q < 2^64 must imply 8 q < 2^64");
        const uint64_t qmax = UINT64_MAX / (3ull * desc->log_n + 8);
        if (q < (1ull << 20) || q > qmax || (q & (two_n - 1)) != 1)
            return lm_fail(nullptr, "modulus %u (%llu) must be == 1 mod 2N and below %llu", i,
                           (unsigned long long)q, (unsigned long long)qmax);
        // psi must be a primitive 2N-th root: psi^N == -1
        if (h_powmod(psi, N, q) != q - 1)
            return lm_fail(nullptr, "psi[%u] is not a primitive 2N-th root of unity", i);
        ctx->mod[i] = q;
        ctx->psi[i] = psi;
        ctx->mods.m[i] = lm_make_mod(q);
        ctx->ninv[i] = h_tw(h_invmod(N % q, q), q);
    }
    for (uint32_t i = LK; i < LM_MAX_LIMBS; i++) ctx->mods.m[i] = ctx->mods.m[0];
    if (int rc = ctx_private_init(ctx)) return rc;
    LM_HIP(ctx, hipMalloc((void **)&ctx->d_tw_fwd, (size_t)LK * N * sizeof(tw_t)));
    LM_HIP(ctx, hipMalloc((void **)&ctx->d_tw_inv, (size_t)LK * N * sizeof(tw_t)));
    std::vector<tw_t> f, b;
    for (uint32_t i = 0; i < LK; i++) {
        lm_build_tw(ctx->mod[i], ctx->psi[i], ctx->logN, f, b);
        LM_HIP(ctx, hipMemcpy(ctx->d_tw_fwd + (size_t)i * N, f.data(), N * sizeof(tw_t), hipMemcpyHostToDevice));
        LM_HIP(ctx, hipMemcpy(ctx->d_tw_inv + (size_t)i * N, b.data(), N * sizeof(tw_t), hipMemcpyHostToDevice));
    }
    *out = guard.release();
    return 0;
}

// ServerBFV.CopyNew (fhe/bfv.go:56-58): what Evaluator.ShallowCopy gives a goroutine -- the same
// parameters, twiddles, field table and keys (shared, read-only), its own streams, scratch, storage
// pool, counters and error text.  Calls on the clone run concurrently with calls on the source.
extern "C" int lumen_ctx_clone(lumen_ctx *src, lumen_ctx **out) {
    if (!src || !out) return lm_fail(nullptr, "lumen_ctx_clone: NULL argument");
    *out = nullptr;
    LM_ENTER(src);
    std::unique_ptr<lumen_ctx, void (*)(lumen_ctx *)> guard(new lumen_ctx(src->sh), lumen_ctx_destroy);
    lumen_ctx *ctx = guard.get();
    ctx->tune = src->tune;
    ctx->device = src->device;
    ctx->logN = src->logN, ctx->N = src->N, ctx->L = src->L, ctx->K = src->K, ctx->T = src->T;
    memcpy(ctx->mod, src->mod, sizeof(ctx->mod));
    memcpy(ctx->psi, src->psi, sizeof(ctx->psi));
    ctx->mods = src->mods;
    memcpy(ctx->ninv, src->ninv, sizeof(ctx->ninv));
    if (int rc = ctx_private_init(ctx)) {
        src->err = ctx->err;
        return rc;
    }
    *out = guard.release();
    return 0;
}

extern "C" void lumen_ctx_destroy(lumen_ctx *ctx) {
    if (!ctx) return;
    {
        LM_ENTER(ctx);
        lm_sync_all(ctx);
        for (auto &kv : ctx->scratch) hipFree(kv.second.first);
        pool_drain(ctx);
        lm_prof_resolve(ctx);
        for (hipEvent_t e : ctx->ev_pool) hipEventDestroy(e);
        if (ctx->tm0) hipEventDestroy(ctx->tm0);
        if (ctx->tm1) hipEventDestroy(ctx->tm1);
        if (ctx->aux_host) hipHostFree(ctx->aux_host);
        if (ctx->stage_host) hipHostFree(ctx->stage_host);
        for (int i = 0; i < 2; i++) {
            if (ctx->io_host[i]) hipHostFree(ctx->io_host[i]);
            if (ctx->ev_io[i]) hipEventDestroy(ctx->ev_io[i]);
        }
        if (ctx->ev_stage) hipEventDestroy(ctx->ev_stage);
        if (ctx->ev_xdep) hipEventDestroy(ctx->ev_xdep);
        if (ctx->ev_aux) hipEventDestroy(ctx->ev_aux);
        if (ctx->stream_aux) hipStreamDestroy(ctx->stream_aux);
        if (ctx->ev_fork) hipEventDestroy(ctx->ev_fork);
        if (ctx->ev_join) hipEventDestroy(ctx->ev_join);
        if (ctx->stream2) hipStreamDestroy(ctx->stream2);
        if (ctx->stream) hipStreamDestroy(ctx->stream);
        ctx->sh.reset(); // the shared tables go with their last user
    }
    delete ctx;
}

// gives the context's cached device memory back to the driver: the pool of freed set storage and every scratch
// buffer (both are rebuilt on demand).  For a long-lived server between jobs of different shapes, and for a
// process that runs several contexts on one GPU.  Waits for the context's streams first.
extern "C" int lumen_ctx_trim(lumen_ctx *ctx) {
    LM_CHECK(nullptr, ctx, "lumen_ctx_trim: NULL ctx");
    LM_ENTER(ctx);
    LM_CHECK(ctx, !ctx->aux_digests, "lumen_ctx_trim: a lumen_leaf_digests_begin job is in flight");
    lm_sync_all(ctx);
    for (auto it = ctx->scratch.begin(); it != ctx->scratch.end();) {
        // the digest buffers are handed out by pointer (lumen_leaf_digests_end_device, the group's gathered
        // leaves) and are a few hundred KB: they stay
        if (it->first == "digests_async" || it->first == "group_digests") {
            ++it;
            continue;
        }
        hipFree(it->second.first);
        it = ctx->scratch.erase(it);
    }
    pool_drain(ctx);
    return 0;
}

// where a named scratch buffer of the context sits (diagnostics: tools/ks_mac_placement.py); *ptr = NULL when the
// context has not allocated it (yet)
extern "C" int lumen_ctx_scratch_info(lumen_ctx *ctx, const char *name, void **ptr, size_t *bytes) {
    LM_CHECK(nullptr, ctx && name && ptr, "lumen_ctx_scratch_info: NULL argument");
    LM_ENTER(ctx);
    auto it = ctx->scratch.find(name);
    *ptr = it == ctx->scratch.end() ? nullptr : it->second.first;
    if (bytes) *bytes = it == ctx->scratch.end() ? 0 : it->second.second;
    return 0;
}

extern "C" const char *lumen_last_error(const lumen_ctx *ctx) {
    return ctx ? ctx->err.c_str() : lm_global_err.c_str();
}

extern "C" int lumen_sync(lumen_ctx *ctx) {
    LM_CHECK(nullptr, ctx, "lumen_sync: NULL ctx");
    LM_ENTER(ctx);
    LM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" uint64_t lumen_mul_counter(const lumen_ctx *ctx) { return ctx ? ctx->mul_counter : 0; }

// ------------------------------------------------------------------ sets
extern "C" int lumen_set_create(lumen_ctx *ctx, uint32_t count, uint32_t num_limbs, lumen_set **out) {
    return lumen_set_create_lanes(ctx, count, num_limbs, 0, out);
}

extern "C" int lumen_set_create_lanes(lumen_ctx *ctx, uint32_t count, uint32_t num_limbs, uint32_t log_world,
                                      lumen_set **out) {
    LM_CHECK(nullptr, ctx && out, "lumen_set_create: NULL argument");
    LM_ENTER(ctx);
    LM_CHECK(ctx, num_limbs >= 1 && num_limbs <= ctx->L, "num_limbs %u out of range [1,%u]", num_limbs, ctx->L);
    LM_CHECK(ctx, log_world <= 6 && (ctx->N >> log_world) >= 64, "a lane shard of 1/%u of N = %u is narrower than 64 coefficients",
             1u << log_world, ctx->N);
    lumen_set *s = new lumen_set();
    s->count = count;
    s->nl = num_limbs;
    s->logw = log_world;
    s->words = (size_t)count * 2 * num_limbs * (ctx->N >> log_world);
    s->home = ctx;
    if (s->words) {
        const size_t bytes = s->words * sizeof(u64);
        auto it = ctx->pool.find(bytes);
        if (it != ctx->pool.end()) { // reuse a block of exactly this size
            s->d = (u64 *)it->second.p;
            if (hipEvent_t ready = it->second.ready) { // returned in stream order: whatever runs next waits on the device
                (void)hipStreamWaitEvent(ctx->stream, ready, 0);
                (void)hipStreamWaitEvent(ctx->stream2, ready, 0);
                (void)hipStreamWaitEvent(ctx->stream_aux, ready, 0);
                ctx->ev_pool.push_back(ready);
            }
            ctx->pool.erase(it);
            ctx->pool_bytes -= bytes;
        } else {
            hipError_t e = hipMalloc((void **)&s->d, bytes);
            if (e != hipSuccess && !ctx->pool.empty()) { // give the pool back and retry once
                (void)hipGetLastError();
                lm_sync_all(ctx);
                pool_drain(ctx);
                e = hipMalloc((void **)&s->d, bytes);
            }
            if (e != hipSuccess) {
                (void)hipGetLastError();
                delete s;
                return lm_fail(ctx, "hipMalloc(%zu bytes) for a %u x %u-limb set failed: %s", bytes, count,
                               num_limbs, hipGetErrorString(e));
            }
        }
    }
    *out = s;
    return 0;
}

extern "C" void lumen_set_destroy(lumen_ctx *ctx, lumen_set *set) {
    if (!set) return;
    if (ctx) {
        LM_ENTER(ctx);
        // nothing enqueued may still touch the storage once it is back in the pool: the main stream and
        // the second lane always, the side stream when its hashing job reads this very storage (a set
        // destroyed between lumen_leaf_digests_begin and _end) -- any other destroy leaves that job alone
        hipStreamSynchronize(ctx->stream);
        hipStreamSynchronize(ctx->stream2);
        if (ctx->aux_digests && set->d && set->d < ctx->aux_hi && set->d + set->words > ctx->aux_lo)
            hipStreamSynchronize(ctx->stream_aux);
        if (set->owner && set->d) {
            const size_t bytes = set->words * sizeof(u64);
            if (ctx == set->home && ctx->pool_bytes + bytes <= ((size_t)96 << 30)) {
                ctx->pool.emplace(bytes, lumen_ctx::pool_block{set->d, nullptr});
                ctx->pool_bytes += bytes;
            } else {
                hipFree(set->d); // never touch a context the caller did not pass
            }
        }
    } else if (set->owner && set->d) {
        hipFree(set->d); // hipFree waits for the device
    }
    delete set;
}

// lumen_set_destroy without the host wait: the block goes back to the pool behind an event on the main stream
// (which first waits for the second lane); lumen_set_create makes the next user's streams wait for that event.
// Falls back to the blocking form when the block cannot be pooled or a hashing job reads it.
void lm_set_release_async(lumen_ctx *ctx, lumen_set *set) {
    if (!set) return;
    LM_ENTER(ctx);
    const size_t bytes = set->words * sizeof(u64);
    const bool hashing = ctx->aux_digests && set->d && set->d < ctx->aux_hi && set->d + set->words > ctx->aux_lo;
    if (!set->owner || !set->d || ctx != set->home || hashing || ctx->pool_bytes + bytes > ((size_t)96 << 30)) {
        lumen_set_destroy(ctx, set);
        return;
    }
    hipEvent_t ready = lm_ev_get(ctx);
    bool ok = ready && hipEventRecord(ctx->ev_join, ctx->stream2) == hipSuccess &&
              hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0) == hipSuccess && hipEventRecord(ready, ctx->stream) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        if (ready) ctx->ev_pool.push_back(ready);
        lumen_set_destroy(ctx, set);
        return;
    }
    ctx->pool.emplace(bytes, lumen_ctx::pool_block{set->d, ready});
    ctx->pool_bytes += bytes;
    delete set;
}

extern "C" int lumen_set_slice(lumen_ctx *ctx, const lumen_set *set, uint32_t first, uint32_t n,
                               lumen_set **view) {
    LM_CHECK(nullptr, ctx && set && view, "lumen_set_slice: NULL argument");
    LM_ENTER(ctx);
    LM_CHECK(ctx, (uint64_t)first + n <= set->count, "slice [%u,%u) exceeds set of %u", first, first + n, set->count);
    lumen_set *v = new lumen_set();
    const size_t ctw = lm_ctw(ctx, set);
    v->count = n;
    v->nl = set->nl;
    v->logw = set->logw;
    v->d = set->d + (size_t)first * ctw;
    v->words = (size_t)n * ctw;
    v->owner = false;
    *view = v;
    return 0;
}

extern "C" uint32_t lumen_set_count(const lumen_set *set) { return set ? set->count : 0; }
extern "C" uint32_t lumen_set_limbs(const lumen_set *set) { return set ? set->nl : 0; }
extern "C" uint32_t lumen_set_log_world(const lumen_set *set) { return set ? set->logw : 0; }
extern "C" void *lumen_set_device_ptr(const lumen_set *set) { return set ? set->d : nullptr; }

// ---- host <-> device staging.  A pinned host buffer (lumen_host_alloc: what the Go
// shim's stage() should fill) is handed to the DMA engine as it is.  A pageable one goes through two
// pinned bounce buffers: the CPU copies chunk k+1 (a few threads: one core moves ~10 GB/s, the link
// ~55) while the engine moves chunk k.
extern "C" void *lumen_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        lm_fail(nullptr, "hipHostMalloc(%zu) failed", bytes);
        return nullptr;
    }
    return p;
}
extern "C" void lumen_host_free(void *p) {
    if (p) hipHostFree(p);
}

// Lattigo keeps ONE Go slice per limb (ct.Value[k].Coeffs[i]), 98 304 separately allocated 128 KB
// arrays for the input matrix of the headline configuration.  The shim pins them (runtime.Pinner), writes their
// addresses into a C array and makes ONE call: the gather into the flat page-locked buffer that
// lumen_set_upload hands to the DMA engine runs on `threads` host threads (one core moves ~10 GB/s, the PCIe
// link ~55).  _scatter is the way back for lumen_set_download's buffer.
static int host_limbs_move(uint64_t *flat, uint64_t *const *limbs, size_t n, size_t words, uint32_t threads, bool gather) {
    if (!n || !words) return 0;
    const size_t bytes = words * sizeof(uint64_t);
    size_t nthr = threads ? threads : std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    nthr = std::min(nthr, n);
    auto work = [=](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) {
            if (gather) memcpy(flat + i * words, limbs[i], bytes);
            else memcpy(limbs[i], flat + i * words, bytes);
        }
    };
    if (nthr == 1) {
        work(0, n);
        return 0;
    }
    std::vector<std::thread> th;
    for (size_t t = 0; t < nthr; t++) th.emplace_back(work, n * t / nthr, n * (t + 1) / nthr);
    for (auto &x : th) x.join();
    return 0;
}
extern "C" int lumen_host_gather(uint64_t *dst, const uint64_t *const *limbs, size_t n, size_t words, uint32_t threads) {
    LM_CHECK(nullptr, (dst && limbs) || !n, "lumen_host_gather: NULL argument");
    for (size_t i = 0; i < n; i++) LM_CHECK(nullptr, limbs[i], "lumen_host_gather: limb %zu is NULL", i);
    return host_limbs_move(dst, const_cast<uint64_t *const *>(limbs), n, words, threads, true);
}
extern "C" int lumen_host_scatter(const uint64_t *src, uint64_t *const *limbs, size_t n, size_t words, uint32_t threads) {
    LM_CHECK(nullptr, (src && limbs) || !n, "lumen_host_scatter: NULL argument");
    for (size_t i = 0; i < n; i++) LM_CHECK(nullptr, limbs[i], "lumen_host_scatter: limb %zu is NULL", i);
    return host_limbs_move(const_cast<uint64_t *>(src), limbs, n, words, threads, false);
}

bool lm_host_is_pinned(const void *p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError(); // an ordinary malloc'd pointer is "invalid value" here
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

#define LM_IO_CHUNK ((size_t)64 << 20)
static void par_memcpy(void *dst, const void *src, size_t bytes) {
    const size_t nthr = bytes >= ((size_t)8 << 20) ? 4 : 1;
    if (nthr == 1) {
        memcpy(dst, src, bytes);
        return;
    }
    std::thread th[4];
    const size_t part = (bytes / nthr + 63) & ~(size_t)63;
    for (size_t t = 0; t < nthr; t++) {
        const size_t lo = std::min(bytes, t * part), hi = std::min(bytes, lo + part);
        th[t] = std::thread([=] { memcpy((char *)dst + lo, (const char *)src + lo, hi - lo); });
    }
    for (size_t t = 0; t < nthr; t++) th[t].join();
}

static int io_buffers(lumen_ctx *ctx) {
    for (int i = 0; i < 2; i++) {
        if (!ctx->io_host[i]) LM_HIP(ctx, hipHostMalloc(&ctx->io_host[i], LM_IO_CHUNK, hipHostMallocDefault));
        if (!ctx->ev_io[i]) LM_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_io[i], hipEventDisableTiming));
    }
    return 0;
}

// host -> device on the context's stream; returns when `host` may be reused.  A page-locked source is handed
// to the DMA engine as it is, a pageable one goes through the two bounce buffers.
int lm_h2d(lumen_ctx *ctx, void *dev, const void *host, size_t bytes) {
    char *dst = (char *)dev;
    if (bytes <= ((size_t)1 << 20) || lm_host_is_pinned(host)) {
        LM_HIP(ctx, hipMemcpyAsync(dst, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    } else {
        if (int rc = io_buffers(ctx)) return rc;
        int i = 0;
        for (size_t off = 0; off < bytes; off += LM_IO_CHUNK, i ^= 1) {
            const size_t len = std::min(LM_IO_CHUNK, bytes - off);
            LM_HIP(ctx, hipEventSynchronize(ctx->ev_io[i])); // the copy that last read this bounce buffer
            par_memcpy(ctx->io_host[i], (const char *)host + off, len);
            LM_HIP(ctx, hipMemcpyAsync(dst + off, ctx->io_host[i], len, hipMemcpyHostToDevice, ctx->stream));
            LM_HIP(ctx, hipEventRecord(ctx->ev_io[i], ctx->stream));
        }
    }
    LM_HIP(ctx, hipStreamSynchronize(ctx->stream)); // `host` is caller memory
    return 0;
}

extern "C" int lumen_set_upload(lumen_ctx *ctx, lumen_set *set, uint32_t first, uint32_t n,
                                const uint64_t *host) {
    LM_CHECK(nullptr, ctx && set && host, "lumen_set_upload: NULL argument");
    LM_ENTER(ctx);
    LM_CHECK(ctx, (uint64_t)first + n <= set->count, "upload range [%u,%u) exceeds set of %u", first, first + n, set->count);
    const size_t ctw = lm_ctw(ctx, set), bytes = (size_t)n * ctw * sizeof(u64);
    return lm_h2d(ctx, set->d + (size_t)first * ctw, host, bytes);
}

// device -> host on the context's stream.  Pageable destinations: chunk k+1 crosses the link while the CPU
// copies chunk k out of its bounce buffer (that path always returns with `host` filled).
int lm_d2h(lumen_ctx *ctx, void *host, const void *dev, size_t bytes, bool wait) {
    const char *src = (const char *)dev;
    if (bytes <= ((size_t)1 << 20) || lm_host_is_pinned(host)) {
        LM_HIP(ctx, hipMemcpyAsync(host, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        if (wait || !lm_host_is_pinned(host)) LM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return 0;
    }
    if (int rc = io_buffers(ctx)) return rc;
    const size_t nchunks = (bytes + LM_IO_CHUNK - 1) / LM_IO_CHUNK;
    auto issue = [&](size_t k) -> int {
        const size_t off = k * LM_IO_CHUNK, len = std::min(LM_IO_CHUNK, bytes - off);
        LM_HIP(ctx, hipMemcpyAsync(ctx->io_host[k & 1], src + off, len, hipMemcpyDeviceToHost, ctx->stream));
        LM_HIP(ctx, hipEventRecord(ctx->ev_io[k & 1], ctx->stream));
        return 0;
    };
    if (nchunks)
        if (int rc = issue(0)) return rc;
    for (size_t k = 0; k < nchunks; k++) {
        if (k + 1 < nchunks)
            if (int rc = issue(k + 1)) return rc;
        const size_t off = k * LM_IO_CHUNK, len = std::min(LM_IO_CHUNK, bytes - off);
        LM_HIP(ctx, hipEventSynchronize(ctx->ev_io[k & 1]));
        par_memcpy((char *)host + off, ctx->io_host[k & 1], len);
    }
    return 0;
}

extern "C" int lumen_set_download(lumen_ctx *ctx, const lumen_set *set, uint32_t first, uint32_t n,
                                  uint64_t *host) {
    LM_CHECK(nullptr, ctx && set && host, "lumen_set_download: NULL argument");
    LM_ENTER(ctx);
    LM_CHECK(ctx, (uint64_t)first + n <= set->count, "download range [%u,%u) exceeds set of %u", first, first + n, set->count);
    const size_t ctw = lm_ctw(ctx, set), bytes = (size_t)n * ctw * sizeof(u64);
    return lm_d2h(ctx, host, set->d + (size_t)first * ctw, bytes, true);
}

__global__ void k_fill_random(u64 *d, size_t words, uint32_t N, uint32_t nl, lm_mods mods, u64 seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < words; i += stride) {
        uint32_t limb = (uint32_t)((i / N) % nl);
        u64 z = seed + 0x9e3779b97f4a7c15ull * (u64)(i + 1);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        z ^= z >> 31;
        d[i] = lm_reduce(z, mods.m[limb].q, mods.m[limb].qinv64);
    }
}

extern "C" int lumen_set_fill_random(lumen_ctx *ctx, lumen_set *set, uint64_t seed) {
    LM_CHECK(nullptr, ctx && set, "lumen_set_fill_random: NULL argument");
    LM_ENTER(ctx);
    if (!set->words) return 0;
    hipLaunchKernelGGL(k_fill_random, dim3(2048), dim3(256), 0, ctx->stream, set->d, set->words,
                       ctx->N >> set->logw, set->nl, ctx->mods, (u64)seed);
    LM_HIP(ctx, hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ timing
extern "C" int lumen_timer_start(lumen_ctx *ctx) {
    LM_CHECK(nullptr, ctx, "NULL ctx");
    LM_ENTER(ctx);
    LM_HIP(ctx, hipEventRecord(ctx->tm0, ctx->stream));
    return 0;
}

extern "C" int lumen_timer_stop(lumen_ctx *ctx, float *elapsed_ms) {
    LM_CHECK(nullptr, ctx && elapsed_ms, "NULL argument");
    LM_ENTER(ctx);
    LM_HIP(ctx, hipEventRecord(ctx->tm1, ctx->stream));
    LM_HIP(ctx, hipEventSynchronize(ctx->tm1));
    LM_HIP(ctx, hipEventElapsedTime(elapsed_ms, ctx->tm0, ctx->tm1));
    return 0;
}

extern "C" int lumen_prof_enable(lumen_ctx *ctx, int on) {
    LM_CHECK(nullptr, ctx, "NULL ctx");
    LM_ENTER(ctx);
    if (!on) lm_prof_resolve(ctx);
    ctx->prof = on != 0;
    return 0;
}

extern "C" int lumen_prof_reset(lumen_ctx *ctx) {
    LM_CHECK(nullptr, ctx, "NULL ctx");
    LM_ENTER(ctx);
    lm_prof_resolve(ctx);
    ctx->prof_tab.clear();
    return 0;
}

extern "C" size_t lumen_prof_names(lumen_ctx *ctx, char *buf, size_t cap) {
    if (!ctx) return 0;
    LM_ENTER(ctx);
    lm_prof_resolve(ctx);
    std::string all;
    for (auto &kv : ctx->prof_tab) {
        if (!all.empty()) all += ",";
        all += kv.first;
    }
    if (buf && cap) {
        size_t n = std::min(cap - 1, all.size());
        memcpy(buf, all.data(), n);
        buf[n] = 0;
    }
    return all.size();
}

extern "C" int lumen_prof_read(lumen_ctx *ctx, const char *kernel, double *total_ms,
                               uint64_t *launches, uint64_t *units) {
    LM_CHECK(nullptr, ctx && kernel, "NULL argument");
    LM_ENTER(ctx);
    lm_prof_resolve(ctx);
    auto it = ctx->prof_tab.find(kernel);
    lm_prof_entry e;
    if (it != ctx->prof_tab.end()) e = it->second;
    if (total_ms) *total_ms = e.total_ms;
    if (launches) *launches = e.launches;
    if (units) *units = e.units;
    return 0;
}
