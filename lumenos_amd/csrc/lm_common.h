// Internal definitions shared by the HIP translation units of liblumenos_hip.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>

#include "../../include/lumenos_hip.h"
#include "lm_arith.h"

#define LM_MAX_LIMBS LUMEN_MAX_LIMBS

struct lm_modmap {
    // modulus index of limb-slot j of a buffer whose limbs cycle with `period`
    uint32_t period;
    uint8_t idx[LM_MAX_LIMBS];
};

struct lm_mods {
    mod_t m[LM_MAX_LIMBS];
};

struct lm_prof_entry {
    double total_ms = 0;
    uint64_t launches = 0;
    uint64_t units = 0;
};

struct lm_galois_key {
    u64 *d_key = nullptr;   // [beta][2][L+K][N], Montgomery form
    uint32_t *d_index = nullptr; // automorphism gather table, N entries: out[i] = in[index[i]]
    uint32_t *d_inv_index = nullptr; // its inverse: out[inv_index[p]] = in[p]
};

struct lumen_set {
    uint32_t count = 0;
    uint32_t nl = 0;
    u64 *d = nullptr;
    size_t words = 0;
    bool owner = true;
    lumen_ctx *home = nullptr; // context whose pool the storage returns to
};

struct lumen_ctx {
    int device = 0;
    hipStream_t stream = nullptr;  // where every entry point enqueues (may be swapped to stream2 internally)
    hipStream_t stream2 = nullptr; // second lane for independent column batches (key-switch pipeline)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipStream_t stream_aux = nullptr; // side jobs that overlap the main stream (leaf hashing)
    hipEvent_t ev_aux = nullptr;
    uint32_t aux_digests = 0;         // leaves of the lumen_leaf_digests_begin job in flight
    uint8_t *aux_host = nullptr;      // pinned staging of its digests
    size_t aux_host_cap = 0;
    uint32_t logN = 0, N = 0, L = 0, K = 0;
    uint64_t T = 0;
    uint64_t mod[LM_MAX_LIMBS] = {0};
    uint64_t psi[LM_MAX_LIMBS] = {0};
    lm_mods mods; // device-side constants by value
    // twiddle tables, device: [L+K][N] tw_t
    tw_t *d_tw_fwd = nullptr;
    tw_t *d_tw_inv = nullptr;
    tw_t ninv[LM_MAX_LIMBS]; // N^-1 mod q_i
    // plaintext field table (core.PrimeField)
    std::vector<uint64_t> roots;
    uint32_t fieldN = 0;
    tw_t *d_scal = nullptr; // [L][fieldN+1] centred twiddle scalars per limb
    uint64_t mul_counter = 0;
    // galois keys
    std::map<uint64_t, lm_galois_key> gkeys;
    // scratch
    std::map<std::string, std::pair<void *, size_t>> scratch;
    // freed set storage kept for reuse: a prover run allocates the same set sizes every time, and
    // mapping/unmapping tens of GB of HBM per call costs more than the kernels that fill it
    std::multimap<size_t, void *> pool;
    size_t pool_bytes = 0;
    // kernels whose dynamic-LDS limit has already been raised on this device
    std::set<const void *> lds_attr_done;
    // per-context derived tables owned by other translation units (key-switch constants,
    // ciphertext-transform plans); released with the context
    std::map<std::string, std::shared_ptr<void>> ext;
    // profiling
    bool prof = false;
    std::map<std::string, lm_prof_entry> prof_tab;
    // event pairs recorded around launches while profiling; resolved lazily so
    // that measuring does not serialise the stream
    struct pending_ev {
        std::string name;
        hipEvent_t a, b;
        uint64_t units;
    };
    std::vector<pending_ev> prof_pending;
    std::vector<hipEvent_t> ev_pool;
    hipEvent_t tm0 = nullptr, tm1 = nullptr;
    std::string err;
};

int lm_fail(lumen_ctx *ctx, const char *fmt, ...);
extern thread_local std::string lm_global_err;

#define LM_HIP(ctx, call)                                                                  \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess)                                                              \
            return lm_fail(ctx, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),     \
                           __FILE__, __LINE__);                                            \
    } while (0)

#define LM_CHECK(ctx, cond, ...)                    \
    do {                                            \
        if (!(cond)) return lm_fail(ctx, __VA_ARGS__); \
    } while (0)

// raise a kernel's dynamic-LDS limit once (hipFuncSetAttribute is not free on the launch path)
#define LM_LDS_ATTR(ctx, kernel, bytes)                                                                   \
    do {                                                                                                  \
        const void *fp_ = reinterpret_cast<const void *>(&kernel);                                        \
        if (!(ctx)->lds_attr_done.count(fp_)) {                                                           \
            LM_HIP(ctx, hipFuncSetAttribute(fp_, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            (ctx)->lds_attr_done.insert(fp_);                                                             \
        }                                                                                                 \
    } while (0)

// scratch buffer that persists in the context and only grows
void *lm_scratch(lumen_ctx *ctx, const char *name, size_t bytes);

// profiling bracket: records HIP-event time of what is enqueued between begin/end
struct lm_prof_scope {
    lumen_ctx *ctx;
    const char *name;
    uint64_t units;
    hipEvent_t a = nullptr;
    lm_prof_scope(lumen_ctx *c, const char *n, uint64_t u);
    ~lm_prof_scope();
};
void lm_prof_resolve(lumen_ctx *ctx);

// host modular helpers
static inline uint64_t h_mulmod(uint64_t a, uint64_t b, uint64_t q) {
    return (uint64_t)(((u128)a * b) % q);
}
static inline uint64_t h_powmod(uint64_t a, uint64_t e, uint64_t q) {
    uint64_t r = 1 % q;
    a %= q;
    while (e) {
        if (e & 1) r = h_mulmod(r, a, q);
        a = h_mulmod(a, a, q);
        e >>= 1;
    }
    return r;
}
static inline uint64_t h_invmod(uint64_t a, uint64_t q) { return h_powmod(a, q - 2, q); }
static inline tw_t h_tw(uint64_t w, uint64_t q) {
    tw_t t;
    t.w = w;
    t.wp = (u64)((((u128)w) << 64) / q);
    return t;
}
static inline uint32_t h_bitrev(uint32_t x, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) r = (r << 1) | ((x >> i) & 1);
    return r;
}

// ---- cross-TU launch helpers
// all limbs of `npoly` polynomials stored [npoly][period][N]; limb j uses modulus map.idx[j]
int lm_launch_ntt(lumen_ctx *ctx, u64 *d, uint32_t npoly, const lm_modmap &map, bool inverse);
struct lm_ninv_t;
// inv_scale: per-modulus multiplier applied by the inverse transform instead of N^-1 (NULL = N^-1)
int lm_launch_ntt_strided(lumen_ctx *ctx, const u64 *src, size_t src_poly_stride, u64 *dst,
                          size_t dst_poly_stride, uint32_t npoly, const lm_modmap &map, bool inverse,
                          const char *prof_name, const lm_ninv_t *inv_scale = nullptr);
// explicit_mod: transform modulo a modulus that is not one of the context's limbs (the plaintext
// modulus T of the encoder); otherwise the modulus is mods.m[mod_idx]
int lm_launch_ntt_subring(lumen_ctx *ctx, uint32_t logn, const tw_t *tw, tw_t ninv_scale, const u64 *src,
                          size_t src_poly_stride, u64 *dst, size_t dst_poly_stride, uint32_t npoly,
                          uint32_t mod_idx, bool inverse, const mod_t *explicit_mod = nullptr);
mod_t lm_make_mod(uint64_t q);
// bit-reversed psi-power tables of a negacyclic transform, forward and inverse, Shoup form
void lm_build_tw(uint64_t q, uint64_t psi, uint32_t logN, std::vector<tw_t> &fwd, std::vector<tw_t> &inv);
int lm_rescale_polys(lumen_ctx *ctx, const u64 *src, uint32_t nl, u64 *dst, uint32_t target,
                     uint32_t npoly, u64 *work, u64 *tbuf);
lm_modmap lm_map_q(uint32_t nl);
