// Context, tables and HBM-resident ciphertext sets.
// Replaces what fhe.NewBackendBFV / ServerBFV.CopyNew capture on the Go side
// (fhe/bfv.go:13-58): ring degree, moduli chains and the per-modulus NTT
// tables (Lattigo SubRing.RootsForward/RootsBackward -- rebuilt here from the
// primitive 2N-th root the host passes, with Shoup companions instead of
// Montgomery form).
#include <cstring>

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
        hipStreamSynchronize(ctx->stream);
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

static hipEvent_t ev_get(lumen_ctx *ctx) {
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
    a = ev_get(ctx);
    hipEventRecord(a, ctx->stream);
}
lm_prof_scope::~lm_prof_scope() {
    if (!ctx->prof || !a) return;
    hipEvent_t b = ev_get(ctx);
    hipEventRecord(b, ctx->stream);
    ctx->prof_pending.push_back({name, a, b, units});
}

void lm_prof_resolve(lumen_ctx *ctx) {
    if (ctx->prof_pending.empty()) return;
    hipStreamSynchronize(ctx->stream);
    for (auto &p : ctx->prof_pending) {
        float ms = 0;
        hipEventElapsedTime(&ms, p.a, p.b);
        auto &e = ctx->prof_tab[p.name];
        e.total_ms += ms;
        e.launches += 1;
        e.units += p.units;
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

extern "C" int lumen_ctx_create(const lumen_params_desc *desc, lumen_ctx **out) {
    if (!desc || !out) return lm_fail(nullptr, "lumen_ctx_create: NULL argument");
    *out = nullptr;
    LM_CHECK(nullptr, desc->abi_version == LUMEN_ABI_VERSION, "ABI version mismatch: got %u want %u",
             desc->abi_version, LUMEN_ABI_VERSION);
    LM_CHECK(nullptr, lm_logn_supported(desc->log_n),
             "log_n %u unsupported: kernels are instantiated for N = 2^8 and 2^10..2^16", desc->log_n);
    LM_CHECK(nullptr, desc->num_q >= 1 && desc->num_q + desc->num_p <= LM_MAX_LIMBS,
             "bad limb counts L=%u K=%u", desc->num_q, desc->num_p);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return lm_fail(nullptr, "no HIP device visible: the lumenos HIP path has no CPU fallback");
    LM_CHECK(nullptr, desc->device >= 0 && desc->device < ndev, "device %d out of range (have %d)",
             desc->device, ndev);
    lumen_ctx *ctx = new lumen_ctx();
    ctx->device = desc->device;
    LM_HIP(ctx, hipSetDevice(ctx->device));
    LM_HIP(ctx, hipStreamCreate(&ctx->stream));
    LM_HIP(ctx, hipStreamCreate(&ctx->stream2));
    LM_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    LM_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    LM_HIP(ctx, hipStreamCreate(&ctx->stream_aux));
    LM_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_aux, hipEventDisableTiming));
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
        const uint64_t qmax = UINT64_MAX / (3ull * desc->log_n + 8);
        if (q < (1ull << 20) || q > qmax || (q & (two_n - 1)) != 1) {
            delete ctx;
            return lm_fail(nullptr, "modulus %u (%llu) must be == 1 mod 2N and below %llu", i,
                           (unsigned long long)q, (unsigned long long)qmax);
        }
        // psi must be a primitive 2N-th root: psi^N == -1
        if (h_powmod(psi, N, q) != q - 1) {
            delete ctx;
            return lm_fail(nullptr, "psi[%u] is not a primitive 2N-th root of unity", i);
        }
        ctx->mod[i] = q;
        ctx->psi[i] = psi;
        ctx->mods.m[i] = lm_make_mod(q);
        ctx->ninv[i] = h_tw(h_invmod(N % q, q), q);
    }
    for (uint32_t i = LK; i < LM_MAX_LIMBS; i++) ctx->mods.m[i] = ctx->mods.m[0];
    LM_HIP(ctx, hipMalloc((void **)&ctx->d_tw_fwd, (size_t)LK * N * sizeof(tw_t)));
    LM_HIP(ctx, hipMalloc((void **)&ctx->d_tw_inv, (size_t)LK * N * sizeof(tw_t)));
    std::vector<tw_t> f, b;
    for (uint32_t i = 0; i < LK; i++) {
        lm_build_tw(ctx->mod[i], ctx->psi[i], ctx->logN, f, b);
        LM_HIP(ctx, hipMemcpy(ctx->d_tw_fwd + (size_t)i * N, f.data(), N * sizeof(tw_t), hipMemcpyHostToDevice));
        LM_HIP(ctx, hipMemcpy(ctx->d_tw_inv + (size_t)i * N, b.data(), N * sizeof(tw_t), hipMemcpyHostToDevice));
    }
    LM_HIP(ctx, hipEventCreate(&ctx->tm0));
    LM_HIP(ctx, hipEventCreate(&ctx->tm1));
    *out = ctx;
    return 0;
}

extern "C" void lumen_ctx_destroy(lumen_ctx *ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    hipFree(ctx->d_tw_fwd);
    hipFree(ctx->d_tw_inv);
    hipFree(ctx->d_scal);
    for (auto &kv : ctx->gkeys) {
        hipFree(kv.second.d_key);
        hipFree(kv.second.d_index);
        hipFree(kv.second.d_inv_index);
    }
    for (auto &kv : ctx->scratch) hipFree(kv.second.first);
    for (auto &kv : ctx->pool) hipFree(kv.second);
    ctx->ext.clear();
    lm_prof_resolve(ctx);
    for (hipEvent_t e : ctx->ev_pool) hipEventDestroy(e);
    hipEventDestroy(ctx->tm0);
    hipEventDestroy(ctx->tm1);
    if (ctx->stream_aux) hipStreamSynchronize(ctx->stream_aux);
    if (ctx->aux_host) hipHostFree(ctx->aux_host);
    hipEventDestroy(ctx->ev_aux);
    hipStreamDestroy(ctx->stream_aux);
    hipEventDestroy(ctx->ev_fork);
    hipEventDestroy(ctx->ev_join);
    hipStreamDestroy(ctx->stream2);
    hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" const char *lumen_last_error(const lumen_ctx *ctx) {
    return ctx ? ctx->err.c_str() : lm_global_err.c_str();
}

extern "C" int lumen_sync(lumen_ctx *ctx) {
    LM_CHECK(nullptr, ctx, "lumen_sync: NULL ctx");
    LM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" uint64_t lumen_mul_counter(const lumen_ctx *ctx) { return ctx ? ctx->mul_counter : 0; }

// ------------------------------------------------------------------ sets
extern "C" int lumen_set_create(lumen_ctx *ctx, uint32_t count, uint32_t num_limbs, lumen_set **out) {
    LM_CHECK(nullptr, ctx && out, "lumen_set_create: NULL argument");
    LM_CHECK(ctx, num_limbs >= 1 && num_limbs <= ctx->L, "num_limbs %u out of range [1,%u]", num_limbs, ctx->L);
    lumen_set *s = new lumen_set();
    s->count = count;
    s->nl = num_limbs;
    s->words = (size_t)count * 2 * num_limbs * ctx->N;
    s->home = ctx;
    if (s->words) {
        const size_t bytes = s->words * sizeof(u64);
        auto it = ctx->pool.find(bytes);
        if (it != ctx->pool.end()) { // reuse a block of exactly this size
            s->d = (u64 *)it->second;
            ctx->pool.erase(it);
            ctx->pool_bytes -= bytes;
        } else {
            hipError_t e = hipMalloc((void **)&s->d, bytes);
            if (e != hipSuccess && !ctx->pool.empty()) { // give the pool back and retry once
                hipStreamSynchronize(ctx->stream);
                for (auto &kv : ctx->pool) hipFree(kv.second);
                ctx->pool.clear();
                ctx->pool_bytes = 0;
                e = hipMalloc((void **)&s->d, bytes);
            }
            if (e != hipSuccess) {
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
    if (ctx) hipStreamSynchronize(ctx->stream);
    if (set->owner && set->d) {
        lumen_ctx *home = (ctx && ctx == set->home) ? ctx : nullptr; // never touch a context the caller did not pass
        const size_t bytes = set->words * sizeof(u64);
        if (home && home->pool_bytes + bytes <= ((size_t)96 << 30)) {
            home->pool.emplace(bytes, set->d);
            home->pool_bytes += bytes;
        } else {
            hipFree(set->d);
        }
    }
    delete set;
}

extern "C" int lumen_set_slice(lumen_ctx *ctx, const lumen_set *set, uint32_t first, uint32_t n,
                               lumen_set **view) {
    LM_CHECK(nullptr, ctx && set && view, "lumen_set_slice: NULL argument");
    LM_CHECK(ctx, (uint64_t)first + n <= set->count, "slice [%u,%u) exceeds set of %u", first, first + n, set->count);
    lumen_set *v = new lumen_set();
    const size_t ctw = (size_t)2 * set->nl * ctx->N;
    v->count = n;
    v->nl = set->nl;
    v->d = set->d + (size_t)first * ctw;
    v->words = (size_t)n * ctw;
    v->owner = false;
    *view = v;
    return 0;
}

extern "C" uint32_t lumen_set_count(const lumen_set *set) { return set ? set->count : 0; }
extern "C" uint32_t lumen_set_limbs(const lumen_set *set) { return set ? set->nl : 0; }
extern "C" void *lumen_set_device_ptr(const lumen_set *set) { return set ? set->d : nullptr; }

extern "C" int lumen_set_upload(lumen_ctx *ctx, lumen_set *set, uint32_t first, uint32_t n,
                                const uint64_t *host) {
    LM_CHECK(nullptr, ctx && set && host, "lumen_set_upload: NULL argument");
    LM_CHECK(ctx, (uint64_t)first + n <= set->count, "upload range [%u,%u) exceeds set of %u", first, first + n, set->count);
    size_t ctw = (size_t)2 * set->nl * ctx->N;
    LM_HIP(ctx, hipMemcpyAsync(set->d + (size_t)first * ctw, host, (size_t)n * ctw * sizeof(u64),
                               hipMemcpyHostToDevice, ctx->stream));
    LM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int lumen_set_download(lumen_ctx *ctx, const lumen_set *set, uint32_t first, uint32_t n,
                                  uint64_t *host) {
    LM_CHECK(nullptr, ctx && set && host, "lumen_set_download: NULL argument");
    LM_CHECK(ctx, (uint64_t)first + n <= set->count, "download range [%u,%u) exceeds set of %u", first, first + n, set->count);
    size_t ctw = (size_t)2 * set->nl * ctx->N;
    LM_HIP(ctx, hipMemcpyAsync(host, set->d + (size_t)first * ctw, (size_t)n * ctw * sizeof(u64),
                               hipMemcpyDeviceToHost, ctx->stream));
    LM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
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
    if (!set->words) return 0;
    hipLaunchKernelGGL(k_fill_random, dim3(2048), dim3(256), 0, ctx->stream, set->d, set->words,
                       ctx->N, set->nl, ctx->mods, (u64)seed);
    LM_HIP(ctx, hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ timing
extern "C" int lumen_timer_start(lumen_ctx *ctx) {
    LM_CHECK(nullptr, ctx, "NULL ctx");
    LM_HIP(ctx, hipEventRecord(ctx->tm0, ctx->stream));
    return 0;
}

extern "C" int lumen_timer_stop(lumen_ctx *ctx, float *elapsed_ms) {
    LM_CHECK(nullptr, ctx && elapsed_ms, "NULL argument");
    LM_HIP(ctx, hipEventRecord(ctx->tm1, ctx->stream));
    LM_HIP(ctx, hipEventSynchronize(ctx->tm1));
    LM_HIP(ctx, hipEventElapsedTime(elapsed_ms, ctx->tm0, ctx->tm1));
    return 0;
}

extern "C" int lumen_prof_enable(lumen_ctx *ctx, int on) {
    LM_CHECK(nullptr, ctx, "NULL ctx");
    if (!on) lm_prof_resolve(ctx);
    ctx->prof = on != 0;
    return 0;
}

extern "C" int lumen_prof_reset(lumen_ctx *ctx) {
    LM_CHECK(nullptr, ctx, "NULL ctx");
    lm_prof_resolve(ctx);
    ctx->prof_tab.clear();
    return 0;
}

extern "C" size_t lumen_prof_names(lumen_ctx *ctx, char *buf, size_t cap) {
    if (!ctx) return 0;
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
    lm_prof_resolve(ctx);
    auto it = ctx->prof_tab.find(kernel);
    lm_prof_entry e;
    if (it != ctx->prof_tab.end()) e = it->second;
    if (total_ms) *total_ms = e.total_ms;
    if (launches) *launches = e.launches;
    if (units) *units = e.units;
    return 0;
}
