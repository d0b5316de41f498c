// Internal definitions shared by the HIP translation units of liblumenos_hip.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <map>
#include <memory>
#include <mutex>
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
    u64 *d_key = nullptr;   // [L+K][beta][2][N] (lm_keyswitch.hip, ks_key_at), Montgomery form
    uint32_t *d_index = nullptr; // automorphism gather table, N entries: out[i] = in[index[i]]
    uint32_t *d_inv_index = nullptr; // its inverse: out[inv_index[p]] = in[p]
};

struct lumen_set {
    uint32_t count = 0;
    uint32_t nl = 0;
    // lane shard (multi-GPU Encode): the set holds coefficients [rank * N >> logw, (rank + 1) * N >> logw)
    // of every limb, i.e. limbs of N >> logw words.  0 for ordinary sets; only the lane entry points,
    // upload / download / fill and destroy accept anything else.
    uint32_t logw = 0;
    u64 *d = nullptr;
    size_t words = 0;
    bool owner = true;
    lumen_ctx *home = nullptr; // context whose pool the storage returns to
};

// Device tables that never change once loaded: shared by a context and its clones
// (lumen_ctx_clone = ServerBFV.CopyNew / Evaluator.ShallowCopy, fhe/bfv.go:56-58: the copies share
// parameters and keys and own their scratch).  `mu` guards the maps (a clone may build a cached plan or
// table while another one looks one up); the tables themselves are read-only on the device.
struct lm_shared {
    std::recursive_mutex mu;
    // twiddle tables, device: [L+K][N] tw_t
    tw_t *d_tw_fwd = nullptr;
    tw_t *d_tw_inv = nullptr;
    // plaintext field table (core.PrimeField)
    std::vector<uint64_t> roots;
    uint32_t fieldN = 0;
    tw_t *d_scal = nullptr; // [L][fieldN+1] centred twiddle scalars per limb
    std::map<uint64_t, lm_galois_key> gkeys;
    // derived tables owned by other translation units (key-switch constants, ciphertext-transform
    // plans, public/secret/ring-switch keys)
    std::map<std::string, std::shared_ptr<void>> ext;
    ~lm_shared();
};

// Tuning switches of the A/B tools ("Run-time switches").  The environment is read ONCE, in
// lumen_ctx_create (a clone copies its source's values): no getenv under LM_ENTER, so a setenv in the host
// process can neither race with nor change the kernels a running prover uses.  lumen_ctx_set_tuning is the
// in-process way for tests and tools.
struct lm_tuning {
    uint32_t ks_batch = 64;      // LUMEN_KS_BATCH: columns per key-switch batch
    uint32_t ks_lanes = 0;       // LUMEN_KS_LANES: 1 / 2 streams for the column batches of a key switch; 0 = by ring degree
    int32_t ks_fused_digits = -1; // LUMEN_KS_FUSED_DIGITS: digits packed inside k_intt_pack (-1: derive)
    uint32_t debug = 0;          // LUMEN_DEBUG
    // LUMEN_MODUP_TGROUP / LUMEN_MODDOWN_TGROUP: target limbs one XCD walks back to back in the work lists of the
    // two transform kernels of a key switch (lm_keyswitch.hip); 1 .. 31
    uint32_t modup_tgroup = 3, moddown_tgroup = 1; // (round 6, limb-major streams: extension 3 / 4 = 1.7732 / 1.7760 s per step at 2^14, 0.7868 / 0.7915 at 2^13; ModDown 1 / 2 / 3 = 1.7691 / 1.7726 / 1.7711 s per step; 2 / 4 / 6 / 12 = 1.8140 / 1.8258 / 1.8246 / 1.8371 on another box)
    // LUMEN_KS_PLACEMENT: candidate blocks per key-switch scratch buffer among which the first key switch of a context
    // picks by measurement (lm_keyswitch.hip, select_placement); 0 or 1 = take what hipMalloc returns
    uint32_t ks_placement = 6;
    // lumen_test_allow_shared_device_rccl (tests only; not reachable through lumen_ctx_set_tuning or the environment):
    // LUMEN_TRANSPORT_RCCL accepts ranks that share a device, for the test double tests/cpp/fake_rccl.cpp
    uint32_t rccl_shared_device = 0;
};

struct lumen_ctx {
    lm_tuning tune;
    // every entry point locks the context (LM_ENTER): concurrent calls on ONE context are serialised
    // here, not left to the caller (the reference runs the R and Z inner products on two goroutines,
    // fhe/ligero.go:231-242); recursive because entry points call one another
    std::recursive_mutex mu;
    std::shared_ptr<lm_shared> sh;
    int device = 0;
    hipStream_t stream = nullptr;  // where every entry point enqueues (may be swapped to stream2 internally)
    hipStream_t stream2 = nullptr; // second lane for independent column batches (key-switch pipeline)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipStream_t stream_aux = nullptr; // side jobs that overlap the main stream (leaf hashing)
    hipEvent_t ev_aux = nullptr;
    hipEvent_t ev_xdep = nullptr; // lumen_ctx_wait: "everything enqueued on this context so far"
    uint32_t aux_digests = 0;         // leaves of the lumen_leaf_digests_begin job in flight
    const u64 *aux_lo = nullptr, *aux_hi = nullptr; // storage that job is reading
    uint8_t *aux_host = nullptr;      // pinned staging of its digests
    size_t aux_host_cap = 0;
    // pinned staging for small host tables handed to asynchronous copies (plaintexts, index lists)
    void *stage_host = nullptr;
    size_t stage_cap = 0;
    hipEvent_t ev_stage = nullptr; // the last copy out of stage_host
    // two pinned bounce buffers for uploads / downloads of pageable host memory (lumen_set_upload)
    void *io_host[2] = {nullptr, nullptr};
    hipEvent_t ev_io[2] = {nullptr, nullptr};
    uint32_t logN = 0, N = 0, L = 0, K = 0;
    uint64_t T = 0;
    uint64_t mod[LM_MAX_LIMBS] = {0};
    uint64_t psi[LM_MAX_LIMBS] = {0};
    lm_mods mods; // device-side constants by value
    tw_t ninv[LM_MAX_LIMBS]; // N^-1 mod q_i
    // views of the shared tables (same names as before the split)
    tw_t *&d_tw_fwd;
    tw_t *&d_tw_inv;
    std::vector<uint64_t> &roots;
    uint32_t &fieldN;
    tw_t *&d_scal;
    std::map<uint64_t, lm_galois_key> &gkeys;
    std::map<std::string, std::shared_ptr<void>> &ext;
    uint64_t mul_counter = 0;
    // scratch
    std::map<std::string, std::pair<void *, size_t>> scratch;
    // freed set storage kept for reuse: a prover run allocates the same set sizes every time, and
    // mapping/unmapping tens of GB of HBM per call costs more than the kernels that fill it
    // A block may come back in STREAM ORDER (lm_set_release_async: the temporaries of a group call): `ready` is then
    // an event behind the last work that touches it, and whoever takes the block makes the context's streams wait
    // for it on the device -- the host never does.
    struct pool_block {
        void *p;
        hipEvent_t ready;
    };
    std::multimap<size_t, pool_block> pool;
    size_t pool_bytes = 0;
    // kernels whose dynamic-LDS limit has already been raised on this device
    std::set<const void *> lds_attr_done;
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
    explicit lumen_ctx(std::shared_ptr<lm_shared> s)
        : sh(std::move(s)), d_tw_fwd(sh->d_tw_fwd), d_tw_inv(sh->d_tw_inv), roots(sh->roots), fieldN(sh->fieldN),
          d_scal(sh->d_scal), gkeys(sh->gkeys), ext(sh->ext) {}
};

// words of one ciphertext of a set (lane shards are narrower)
static inline size_t lm_ctw(const lumen_ctx *ctx, const lumen_set *s);
#define LM_FULL_WIDTH(ctx, set, what) \
    LM_CHECK(ctx, (set)->logw == 0, "%s: lane-sharded set (1/%u of every limb) where a full-width set is required", what, 1u << (set)->logw)

// first statement of every entry point that takes a context: serialise callers and select the
// context's device for the calling thread (a fresh OS thread -- every cgo call may be one -- starts
// on device 0)
// ... and drop whatever error an earlier HIP call of this thread left behind (the host program's own
// calls, a clean-up path): hipGetLastError() after a launch must speak about that launch
#define LM_ENTER(ctx)                                             \
    std::lock_guard<std::recursive_mutex> lm_lock_((ctx)->mu);    \
    (void)hipSetDevice((ctx)->device);                            \
    (void)hipGetLastError()
// lookups and insertions in the shared maps (ext, gkeys, cached work lists)
#define LM_SHARED_LOCK(ctx) std::lock_guard<std::recursive_mutex> lm_shlock_((ctx)->sh->mu)

// shared derived tables by name: the returned handle keeps the table alive for the caller's scope even if
// another clone replaces the entry meanwhile
template <class T>
static inline std::shared_ptr<T> lm_ext_get(lumen_ctx *ctx, const std::string &key) {
    LM_SHARED_LOCK(ctx);
    auto it = ctx->ext.find(key);
    return it == ctx->ext.end() ? std::shared_ptr<T>() : std::static_pointer_cast<T>(it->second);
}
static inline void lm_ext_put(lumen_ctx *ctx, const std::string &key, std::shared_ptr<void> v) {
    LM_SHARED_LOCK(ctx);
    ctx->ext[key] = std::move(v);
}

static inline size_t lm_ctw(const lumen_ctx *ctx, const lumen_set *s) { return (size_t)2 * s->nl * (ctx->N >> s->logw); }

// waits for every stream of the context (main, second lane, side jobs)
void lm_sync_all(lumen_ctx *ctx);
// pinned host staging of at least `bytes`, safe to overwrite (the previous asynchronous copy out of it
// has completed); record ctx->ev_stage on the stream after enqueuing the next copy
void *lm_stage(lumen_ctx *ctx, size_t bytes);

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

// owns a freshly created set until the entry point hands it to the caller: every early return
// (LM_HIP / LM_CHECK) gives the storage back instead of leaking it
struct lm_set_guard {
    lumen_ctx *ctx;
    lumen_set *s;
    lm_set_guard(lumen_ctx *c, lumen_set *x) : ctx(c), s(x) {}
    ~lm_set_guard() {
        if (s) lumen_set_destroy(ctx, s);
    }
    lumen_set *release() {
        lumen_set *r = s;
        s = nullptr;
        return r;
    }
    lm_set_guard(const lm_set_guard &) = delete;
    lm_set_guard &operator=(const lm_set_guard &) = delete;
};

// gives a set back without blocking the host: its storage returns to the pool behind an event recorded on the
// context's streams (lm_ctx.hip).  For temporaries of calls that must not wait (lumen_group_encode / _gather).
void lm_set_release_async(lumen_ctx *ctx, lumen_set *set);

// scratch buffer that persists in the context and only grows
void *lm_scratch(lumen_ctx *ctx, const char *name, size_t bytes);
void lm_scratch_adopt(lumen_ctx *ctx, const char *name, void *p, size_t bytes);

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
// fhe.Encode with Enc(0) resident on the device; never blocks the host (lm_ctntt.hip)
int lm_encode_dev(lumen_ctx *ctx, const lumen_set *matrix, const u64 *dzero, uint32_t rho_inv, lumen_set **encoded);
// pooled timing events of a context (lm_ctx.hip)
hipEvent_t lm_ev_get(lumen_ctx *ctx);
