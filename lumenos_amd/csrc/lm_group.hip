// Several GPUs behind one call sequence (include/lumenos_hip.h "lumen_group").
//
// The reference is ONE process that owns the whole request (cmd/server/main.go:187-266) and fans its work out
// over goroutine pools (fhe/ligero.go:136-162, 231-242).  A lumen_group is that process's view of W = 2^k
// ranks, one lumen_ctx per GPU, with the exchange steps of the sharded Commit inside the library:
//   * two all-to-alls around the lane-sharded Encode (the ciphertext-axis transform never mixes lanes,
//     fhe/ntt.go:245-279), every travelling block a contiguous slice of a ct-major set;
//   * one all-gather of the S x 32-byte leaf digests (north_star's "single RCCL all-gather");
//   * the queried columns collected on rank 0 (309 level-1 ciphertexts).
// Transports: stream-ordered device copies pulled by the destination's stream (one device: hipMemcpyAsync;
// several devices of one process: hipMemcpyPeerAsync), or RCCL over xGMI -- grouped ncclSend / ncclRecv and
// ncclAllGather on the contexts' own streams.  RCCL is loaded at run time (dlopen): the library has no
// link-time dependency on it, a one-GPU host never loads it, and a process that already carries an RCCL
// (PyTorch's) shares that copy (same soname).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>

#include "lm_common.h"

namespace {

struct rccl_api {
    void *handle = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr; // optional: only used to tear down a poisoned communicator
    std::string why; // why loading failed
};

std::mutex g_rccl_mu;
rccl_api g_rccl;

// dlopen once per process; returns nullptr (with g_rccl.why set) when RCCL is not available
rccl_api *rccl_load() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.handle) return &g_rccl;
    void *h = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    if (!h) {
        const char *e = dlerror();
        g_rccl.why = std::string("dlopen(librccl.so.1) failed: ") + (e ? e : "unknown");
        return nullptr;
    }
    rccl_api a;
    a.handle = h;
    bool ok = true;
#define LM_SYM(field, sym)                                        \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, sym)); \
    if (!a.field) ok = false, a.why += std::string(" ") + sym;
    LM_SYM(GetVersion, "ncclGetVersion")
    LM_SYM(GetUniqueId, "ncclGetUniqueId")
    LM_SYM(CommInitRank, "ncclCommInitRank")
    LM_SYM(CommInitAll, "ncclCommInitAll")
    LM_SYM(CommDestroy, "ncclCommDestroy")
    LM_SYM(CommCount, "ncclCommCount")
    LM_SYM(Send, "ncclSend")
    LM_SYM(Recv, "ncclRecv")
    LM_SYM(AllGather, "ncclAllGather")
    LM_SYM(GroupStart, "ncclGroupStart")
    LM_SYM(GroupEnd, "ncclGroupEnd")
    LM_SYM(GetErrorString, "ncclGetErrorString")
#undef LM_SYM
    a.CommAbort = reinterpret_cast<decltype(a.CommAbort)>(dlsym(h, "ncclCommAbort"));
    if (!ok) {
        g_rccl.why = "librccl lacks:" + a.why;
        dlclose(h);
        return nullptr;
    }
    g_rccl = a;
    return &g_rccl;
}

struct stat_entry {
    double ms = 0;
    uint64_t bytes = 0, calls = 0;
};

} // namespace

struct lumen_group {
    std::recursive_mutex mu;
    uint32_t logw = 0, W = 1;
    std::vector<lumen_ctx *> ctx;   // local contexts, ascending global rank
    std::vector<uint32_t> rank;     // their global ranks
    int transport = LUMEN_TRANSPORT_COPY;
    bool multi_device = false;
    bool staged = false;            // copy transport across devices of which some pair has no peer access
    std::string note;               // how the transport was chosen (lumen_group_transport_note)
    rccl_api *rccl = nullptr;
    std::vector<ncclComm_t> comm;   // one per local context (RCCL)
    uint32_t rccl_ranks = 0;
    // dependency events, one pair per local context, created on its device
    std::vector<hipEvent_t> ev_ready, ev_done;
    // the all-gathered leaf digests: W * n_leaves_per_rank * 32 bytes in every local context's scratch
    std::vector<uint8_t *> d_digests;
    uint32_t n_per_rank = 0;
    // timing of the collectives
    struct pending {
        std::string name;
        uint64_t call;
        uint32_t local;
        hipEvent_t a, b;
        uint64_t bytes;
    };
    std::vector<pending> pend;
    std::map<std::string, stat_entry> stats;
    std::map<std::pair<std::string, uint64_t>, double> call_ms; // calls partly resolved (see stats_resolve)
    uint64_t call_seq = 0;
    // a collective failed between ncclGroupStart and ncclGroupEnd: the group that had to be closed launched sends /
    // receives without their counterparts, so the ranks' streams may never drain.  Every later collective is refused and
    // lumen_group_destroy aborts the communicators instead of waiting for those streams.
    bool poisoned = false;
    std::string poison_why;
};

namespace {

// locks the group and every local context (always in the same order) and leaves the calling thread's
// last-error state clean, like LM_ENTER does for one context
struct group_lock {
    std::unique_lock<std::recursive_mutex> g;
    std::vector<std::unique_lock<std::recursive_mutex>> c;
    explicit group_lock(lumen_group *grp) : g(grp->mu) {
        for (lumen_ctx *x : grp->ctx) c.emplace_back(x->mu);
        (void)hipGetLastError();
    }
};

#define G_HIP(call)                                                                                                  \
    do {                                                                                                             \
        hipError_t e_ = (call);                                                                                      \
        if (e_ != hipSuccess)                                                                                        \
            return lm_fail(nullptr, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);      \
    } while (0)
#define G_NCCL(g, call)                                                                                              \
    do {                                                                                                             \
        ncclResult_t r_ = (call);                                                                                    \
        if (r_ != ncclSuccess)                                                                                       \
            return lm_fail(nullptr, "%s failed: %s (%s:%d)", #call, (g)->rccl->GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

// ncclGroupStart ... ncclGroupEnd around the sends and receives of one collective.  An early return between the two
// (a failed ncclSend, hipSetDevice ...) must not leave the calling thread's RCCL group open: every later RCCL call of
// that thread -- this library's or the host's own (torch's nccl in bench.py's fallback) -- would be queued behind a
// GroupEnd that never comes.  The destructor closes what end() did not.
struct rccl_group_scope {
    lumen_group *g;
    bool open = false;
    explicit rccl_group_scope(lumen_group *grp) : g(grp) {}
    int begin() {
        G_NCCL(g, g->rccl->GroupStart());
        open = true;
        return 0;
    }
    int end() {
        open = false;
        G_NCCL(g, g->rccl->GroupEnd());
        return 0;
    }
    ~rccl_group_scope() {
        if (!open) return;
        (void)g->rccl->GroupEnd(); // the calling thread's RCCL group must not stay open (see above) ...
        g->poisoned = true;        // ... but what it launched is half a collective: nothing may be queued behind it
        g->poison_why = lm_global_err;
    }
};
#define G_USABLE(g, what)                                                                                                      \
    LM_CHECK(nullptr, !(g)->poisoned, "%s: an earlier collective of this group failed half-posted (%s): its streams may never "    \
             "drain -- destroy the group (and its contexts) and create new ones", what, (g)->poison_why.c_str())

int use(lumen_group *g, uint32_t i) {
    G_HIP(hipSetDevice(g->ctx[i]->device));
    return 0;
}

// timing bracket of one collective on local rank i's stream
void stats_resolve(lumen_group *g, bool only_done);
// (the caller has made rank i's device current: events are created on, and recorded from, the current device)
void time_begin(lumen_group *g, const char *name, uint32_t i, uint64_t bytes) {
    if (g->pend.size() > 512) stats_resolve(g, true); // a long run that never reads its statistics
    lumen_group::pending p;
    p.name = name, p.call = g->call_seq, p.local = i, p.bytes = bytes;
    p.a = lm_ev_get(g->ctx[i]);
    p.b = nullptr;
    if (!p.a || hipEventRecord(p.a, g->ctx[i]->stream) != hipSuccess) { // no measurement rather than a bad event in the pool
        (void)hipGetLastError();
        if (p.a) hipEventDestroy(p.a);
        return;
    }
    g->pend.push_back(p);
}
void time_end(lumen_group *g, uint32_t i) {
    for (auto it = g->pend.rbegin(); it != g->pend.rend(); ++it)
        if (it->local == i && it->call == g->call_seq && !it->b) {
            hipEvent_t b = lm_ev_get(g->ctx[i]);
            if (!b || hipEventRecord(b, g->ctx[i]->stream) != hipSuccess) {
                (void)hipGetLastError();
                if (b) hipEventDestroy(b);
                return; // stats_resolve drops a measurement without an end
            }
            it->b = b;
            return;
        }
}

// folds finished measurements into the statistics: per call the slowest local rank's time, bytes as sent by one
// rank.  only_done: leave what is still running on the device for later (no host block)
void stats_resolve(lumen_group *g, bool only_done = false) {
    int dev_before = -1; // the loop below walks over the local ranks' devices: leave the caller's device current
    (void)hipGetDevice(&dev_before);
    struct restore {
        int d;
        ~restore() {
            if (d >= 0) (void)hipSetDevice(d);
        }
    } restore_device{dev_before};
    std::vector<lumen_group::pending> later;
    std::map<std::pair<std::string, uint64_t>, std::pair<double, uint64_t>> per_call;
    for (auto &p : g->pend) {
        float ms = 0;
        (void)hipSetDevice(g->ctx[p.local]->device);
        if (only_done && p.b && hipEventQuery(p.b) != hipSuccess) {
            (void)hipGetLastError();
            later.push_back(p);
            continue;
        }
        if (only_done && !p.b && p.call == g->call_seq) { // the call that is being enqueued right now
            later.push_back(p);
            continue;
        }
        if (p.b && hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            auto &e = per_call[{p.name, p.call}];
            e.first = std::max(e.first, (double)ms);
            e.second = std::max(e.second, p.bytes);
        } else {
            (void)hipGetLastError();
        }
        g->ctx[p.local]->ev_pool.push_back(p.a);
        if (p.b) g->ctx[p.local]->ev_pool.push_back(p.b);
    }
    g->pend.swap(later);
    // a call's local ranks may resolve in different rounds: the call counts once, its time is the maximum seen
    for (auto &kv : per_call) {
        auto &s = g->stats[kv.first.first];
        auto seen = g->call_ms.find(kv.first);
        if (seen == g->call_ms.end()) {
            s.ms += kv.second.first, s.bytes += kv.second.second, s.calls += 1;
            g->call_ms[kv.first] = kv.second.first;
        } else if (kv.second.first > seen->second) {
            s.ms += kv.second.first - seen->second;
            seen->second = kv.second.first;
        }
    }
    if (g->pend.empty()) g->call_ms.clear();
}

// "everything enqueued so far on every local rank" -> ev_ready; then dst's stream waits for all of them
int ready_all(lumen_group *g) {
    for (uint32_t i = 0; i < g->ctx.size(); i++) {
        if (use(g, i)) return 1;
        G_HIP(hipEventRecord(g->ev_ready[i], g->ctx[i]->stream));
    }
    return 0;
}
int wait_ready(lumen_group *g, uint32_t j) {
    for (uint32_t i = 0; i < g->ctx.size(); i++)
        if (i != j) G_HIP(hipStreamWaitEvent(g->ctx[j]->stream, g->ev_ready[i], 0));
    return 0;
}
// the sources may reuse what was read once every destination has pulled its part
int done_all(lumen_group *g, const std::vector<uint32_t> &dsts) {
    for (uint32_t i = 0; i < g->ctx.size(); i++) {
        if (use(g, i)) return 1;
        for (uint32_t j : dsts)
            if (i != j) G_HIP(hipStreamWaitEvent(g->ctx[i]->stream, g->ev_done[j], 0));
    }
    return 0;
}

int copy_between(lumen_group *g, uint32_t dst_local, void *dst, uint32_t src_local, const void *src, size_t bytes) {
    lumen_ctx *d = g->ctx[dst_local], *s = g->ctx[src_local];
    if (!bytes) return 0;
    if (d->device == s->device)
        G_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, d->stream));
    else
        G_HIP(hipMemcpyPeerAsync(dst, d->device, src, s->device, bytes, d->stream));
    return 0;
}

bool same_params(const lumen_ctx *a, const lumen_ctx *b) {
    return a->logN == b->logN && a->L == b->L && a->K == b->K && a->T == b->T &&
           !memcmp(a->mod, b->mod, sizeof(uint64_t) * (a->L + a->K));
}

int group_finish_create(lumen_group *g) {
    const uint32_t n = (uint32_t)g->ctx.size();
    g->ev_ready.assign(n, nullptr), g->ev_done.assign(n, nullptr), g->d_digests.assign(n, nullptr);
    for (uint32_t i = 0; i < n; i++) {
        if (use(g, i)) return 1;
        G_HIP(hipEventCreateWithFlags(&g->ev_ready[i], hipEventDisableTiming));
        G_HIP(hipEventCreateWithFlags(&g->ev_done[i], hipEventDisableTiming));
    }
    return 0;
}

} // namespace

extern "C" void lumen_group_destroy(lumen_group *g) {
    if (!g) return;
    {
        group_lock lk(g);
        if (!g->poisoned) {
            for (uint32_t i = 0; i < g->ctx.size(); i++) {
                (void)hipSetDevice(g->ctx[i]->device);
                lm_sync_all(g->ctx[i]);
            }
            stats_resolve(g);
        } else {
            // half a collective sits on the ranks' streams: waiting for them may never return.  Abort the communicators
            // (that is what releases kernels blocked on a peer) and drop the timing events unread.
            for (auto &p : g->pend) {
                if (p.a) hipEventDestroy(p.a);
                if (p.b) hipEventDestroy(p.b);
            }
            g->pend.clear();
        }
        for (uint32_t i = 0; i < g->comm.size(); i++)
            if (g->comm[i]) {
                (void)hipSetDevice(g->ctx[i]->device);
                if (g->poisoned && g->rccl->CommAbort) g->rccl->CommAbort(g->comm[i]);
                else g->rccl->CommDestroy(g->comm[i]);
            }
        for (hipEvent_t e : g->ev_ready)
            if (e) hipEventDestroy(e);
        for (hipEvent_t e : g->ev_done)
            if (e) hipEventDestroy(e);
    }
    delete g;
}

// RCCL communicators for every rank of a one-process group; on failure `why` says what refused and nothing is left behind
static bool rccl_init_all(lumen_group *g, std::string &why) {
    g->rccl = rccl_load();
    if (!g->rccl) {
        why = g_rccl.why;
        return false;
    }
    const uint32_t W = g->W;
    std::vector<int> devs;
    for (lumen_ctx *c : g->ctx) devs.push_back(c->device);
    g->comm.assign(W, nullptr);
    const ncclResult_t r = g->rccl->CommInitAll(g->comm.data(), (int)W, devs.data());
    if (r != ncclSuccess) {
        why = std::string("ncclCommInitAll over ") + std::to_string(W) + " devices failed: " + g->rccl->GetErrorString(r) +
              " (hosts whose driver only does dmabuf IPC need HSA_ENABLE_IPC_MODE_LEGACY=0 in the process environment; "
              "NCCL_DEBUG=WARN prints RCCL's own reason)";
        g->comm.clear();
        return false;
    }
    int cnt = 0;
    const ncclResult_t rc = g->rccl->CommCount(g->comm[0], &cnt);
    if (rc != ncclSuccess || (uint32_t)cnt != W) {
        why = "the RCCL communicator reports " + std::to_string(cnt) + " ranks, expected " + std::to_string(W);
        for (uint32_t i = 0; i < W; i++)
            if (g->comm[i]) {
                (void)hipSetDevice(g->ctx[i]->device);
                g->rccl->CommDestroy(g->comm[i]);
            }
        g->comm.clear();
        return false;
    }
    g->rccl_ranks = (uint32_t)cnt;
    int ver = 0;
    (void)g->rccl->GetVersion(&ver);
    g->note = "rccl: librccl version " + std::to_string(ver) + ", ncclCommInitAll over " + std::to_string(W) + " devices";
    return true;
}

// copy transport across several devices: peer access where the hardware offers it (xGMI: hipMemcpyPeerAsync then moves
// device to device); a pair without it still works -- the runtime stages such copies through host memory -- but the
// group says so ("copy-staged") instead of passing for a device-to-device exchange
static int peer_setup(lumen_group *g) {
    uint32_t pairs = 0, direct = 0;
    for (uint32_t i = 0; i < g->ctx.size(); i++)
        for (uint32_t j = 0; j < g->ctx.size(); j++) {
            const int di = g->ctx[i]->device, dj = g->ctx[j]->device;
            if (di == dj) continue;
            pairs++;
            int can = 0;
            G_HIP(hipDeviceCanAccessPeer(&can, di, dj));
            if (!can) continue;
            G_HIP(hipSetDevice(di));
            const hipError_t e = hipDeviceEnablePeerAccess(dj, 0);
            if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
            else if (e != hipSuccess)
                return lm_fail(nullptr, "lumen_group_create: hipDeviceEnablePeerAccess(%d -> %d) failed: %s", di, dj, hipGetErrorString(e));
            direct++;
        }
    g->staged = direct < pairs;
    if (pairs) g->note += (g->note.empty() ? "" : "; ") + std::string("peer access on ") + std::to_string(direct) + " of " +
                          std::to_string(pairs) + " device pairs";
    return 0;
}

extern "C" int lumen_group_create(lumen_ctx *const *ctxs, uint32_t log_world, uint32_t transport, lumen_group **out) {
    LM_CHECK(nullptr, ctxs && out, "lumen_group_create: NULL argument");
    *out = nullptr;
    LM_CHECK(nullptr, log_world <= 6, "lumen_group_create: log_world %u out of range [0, 6]", log_world);
    LM_CHECK(nullptr, transport <= LUMEN_TRANSPORT_RCCL, "lumen_group_create: unknown transport %u", transport);
    const uint32_t W = 1u << log_world;
    std::set<int> devices;
    for (uint32_t r = 0; r < W; r++) {
        LM_CHECK(nullptr, ctxs[r], "lumen_group_create: context of rank %u is NULL", r);
        for (uint32_t q = 0; q < r; q++) LM_CHECK(nullptr, ctxs[q] != ctxs[r], "lumen_group_create: ranks %u and %u are the same context", q, r);
        LM_CHECK(nullptr, same_params(ctxs[0], ctxs[r]), "lumen_group_create: rank %u has other parameters than rank 0", r);
        devices.insert(ctxs[r]->device);
    }
    LM_CHECK(nullptr, W == 1 || (ctxs[0]->N >> log_world) >= 64,
             "lumen_group_create: a lane shard of 1/%u of N = %u is narrower than 64 coefficients", W, ctxs[0]->N);
    const bool distinct = devices.size() == W;
    const bool automatic = transport == LUMEN_TRANSPORT_AUTO;
    // (the shared-device exception is a test switch: real RCCL refuses such a communicator itself)
    const bool rccl_possible = distinct || ctxs[0]->tune.rccl_shared_device;
    if (automatic) transport = (rccl_possible && W > 1) ? LUMEN_TRANSPORT_RCCL : LUMEN_TRANSPORT_COPY;
    LM_CHECK(nullptr, transport != LUMEN_TRANSPORT_RCCL || rccl_possible,
             "lumen_group_create: RCCL needs every rank on its own device (%zu devices for %u ranks); use LUMEN_TRANSPORT_COPY", devices.size(), W);
    std::unique_ptr<lumen_group, void (*)(lumen_group *)> guard(new lumen_group(), lumen_group_destroy);
    lumen_group *g = guard.get();
    g->logw = log_world, g->W = W, g->transport = (int)transport, g->multi_device = devices.size() > 1;
    for (uint32_t r = 0; r < W; r++) g->ctx.push_back(ctxs[r]), g->rank.push_back(r);
    group_lock lk(g);
    if (group_finish_create(g)) return 1;
    if (transport == LUMEN_TRANSPORT_RCCL) {
        std::string why;
        if (!rccl_init_all(g, why)) {
            // asked for by name: an error.  Chosen by LUMEN_TRANSPORT_AUTO: the W devices of one process can always
            // exchange by (peer) copies -- a host without librccl, or whose RCCL refuses to initialise, still runs
            LM_CHECK(nullptr, automatic, "lumen_group_create: RCCL transport unavailable: %s", why.c_str());
            (void)hipGetLastError();
            g->transport = LUMEN_TRANSPORT_COPY, g->rccl = nullptr, g->rccl_ranks = 0;
            g->note = "auto: fell back to device copies, RCCL unavailable: " + why;
        }
    }
    if (g->transport == LUMEN_TRANSPORT_COPY && g->multi_device && peer_setup(g)) return 1;
    if (g->note.empty()) g->note = g->transport == LUMEN_TRANSPORT_COPY ? "stream-ordered copies on one device" : "";
    *out = guard.release();
    return 0;
}

extern "C" int lumen_group_unique_id(uint8_t id[128]) {
    LM_CHECK(nullptr, id, "lumen_group_unique_id: NULL argument");
    rccl_api *r = rccl_load();
    LM_CHECK(nullptr, r, "lumen_group_unique_id: RCCL unavailable: %s", g_rccl.why.c_str());
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId u;
    ncclResult_t rc = r->GetUniqueId(&u);
    LM_CHECK(nullptr, rc == ncclSuccess, "ncclGetUniqueId failed: %s", r->GetErrorString(rc));
    memcpy(id, &u, 128);
    return 0;
}

extern "C" int lumen_group_create_rank(lumen_ctx *ctx, uint32_t rank, uint32_t log_world, const uint8_t id[128],
                                       lumen_group **out) {
    LM_CHECK(nullptr, ctx && id && out, "lumen_group_create_rank: NULL argument");
    *out = nullptr;
    LM_CHECK(nullptr, log_world <= 6 && rank < (1u << log_world), "lumen_group_create_rank: rank %u of 2^%u", rank, log_world);
    const uint32_t W = 1u << log_world;
    LM_CHECK(nullptr, W == 1 || (ctx->N >> log_world) >= 64,
             "lumen_group_create_rank: a lane shard of 1/%u of N = %u is narrower than 64 coefficients", W, ctx->N);
    std::unique_ptr<lumen_group, void (*)(lumen_group *)> guard(new lumen_group(), lumen_group_destroy);
    lumen_group *g = guard.get();
    g->logw = log_world, g->W = W, g->transport = LUMEN_TRANSPORT_RCCL;
    g->ctx.push_back(ctx), g->rank.push_back(rank);
    group_lock lk(g);
    if (group_finish_create(g)) return 1;
    g->rccl = rccl_load();
    LM_CHECK(nullptr, g->rccl, "lumen_group_create_rank: RCCL unavailable: %s", g_rccl.why.c_str());
    ncclUniqueId u;
    memcpy(&u, id, 128);
    g->comm.assign(1, nullptr);
    if (use(g, 0)) return 1;
    {
        const ncclResult_t r = g->rccl->CommInitRank(&g->comm[0], (int)W, u, (int)rank);
        LM_CHECK(nullptr, r == ncclSuccess,
                 "ncclCommInitRank (rank %u of %u on device %d) failed: %s -- RCCL needs every rank on its own device, and "
                 "on hosts whose driver only does dmabuf IPC the process environment must carry "
                 "HSA_ENABLE_IPC_MODE_LEGACY=0; NCCL_DEBUG=WARN prints RCCL's own reason", rank, W, ctx->device,
                 g->rccl->GetErrorString(r));
    }
    int cnt = 0;
    G_NCCL(g, g->rccl->CommCount(g->comm[0], &cnt));
    g->rccl_ranks = (uint32_t)cnt;
    LM_CHECK(nullptr, g->rccl_ranks == W, "lumen_group_create_rank: the RCCL communicator reports %u ranks, expected %u", g->rccl_ranks, W);
    int ver = 0;
    (void)g->rccl->GetVersion(&ver);
    g->note = "rccl: librccl version " + std::to_string(ver) + ", ncclCommInitRank " + std::to_string(rank) + " of " + std::to_string(W);
    *out = guard.release();
    return 0;
}

extern "C" uint32_t lumen_group_world(const lumen_group *g) { return g ? g->W : 0; }
extern "C" uint32_t lumen_group_local(const lumen_group *g) { return g ? (uint32_t)g->ctx.size() : 0; }
extern "C" uint32_t lumen_group_rccl_ranks(const lumen_group *g) { return g ? g->rccl_ranks : 0; }
extern "C" const char *lumen_group_transport(const lumen_group *g) {
    if (!g) return "";
    if (g->transport == LUMEN_TRANSPORT_RCCL) return "rccl";
    return !g->multi_device ? "copy" : (g->staged ? "copy-staged" : "copy-peer");
}
extern "C" const char *lumen_group_transport_note(const lumen_group *g) { return g ? g->note.c_str() : ""; }

extern "C" int lumen_group_sync(lumen_group *g) {
    LM_CHECK(nullptr, g, "lumen_group_sync: NULL group");
    group_lock lk(g);
    G_USABLE(g, "lumen_group_sync");
    for (uint32_t i = 0; i < g->ctx.size(); i++) {
        if (use(g, i)) return 1;
        G_HIP(hipStreamSynchronize(g->ctx[i]->stream));
    }
    return 0;
}

// ---- all-to-all on contiguous blocks.  send_ptr[i] / recv_ptr[i]: base of local rank i's W blocks of blk bytes.
static int all_to_all_raw(lumen_group *g, const std::vector<const u64 *> &send, const std::vector<u64 *> &recv, size_t blk_words,
                          const char *stat_name) {
    const uint32_t n = (uint32_t)g->ctx.size(), W = g->W;
    G_USABLE(g, stat_name);
    g->call_seq++;
    const uint64_t sent = (uint64_t)blk_words * 8 * (W - 1);
    if (g->transport == LUMEN_TRANSPORT_RCCL) {
        for (uint32_t i = 0; i < n; i++) {
            if (use(g, i)) return 1;
            time_begin(g, stat_name, i, sent);
        }
        rccl_group_scope grp(g);
        if (grp.begin()) return 1;
        for (uint32_t i = 0; i < n; i++) {
            if (use(g, i)) return 1;
            for (uint32_t p = 0; p < W; p++) {
                G_NCCL(g, g->rccl->Send(send[i] + (size_t)p * blk_words, blk_words, ncclUint64, (int)p, g->comm[i], g->ctx[i]->stream));
                G_NCCL(g, g->rccl->Recv(recv[i] + (size_t)p * blk_words, blk_words, ncclUint64, (int)p, g->comm[i], g->ctx[i]->stream));
            }
        }
        if (grp.end()) return 1;
        for (uint32_t i = 0; i < n; i++) {
            if (use(g, i)) return 1;
            time_end(g, i);
        }
        return 0;
    }
    // copy transport: every rank is local (n == W, local index == rank)
    for (uint32_t i = 0; i < n; i++) {
        if (use(g, i)) return 1;
        time_begin(g, stat_name, i, sent);
    }
    if (ready_all(g)) return 1;
    std::vector<uint32_t> all;
    for (uint32_t j = 0; j < n; j++) {
        if (use(g, j) || wait_ready(g, j)) return 1;
        for (uint32_t k = 0; k < n; k++) {
            const uint32_t i = (j + k) % n; // start with the rank's own block, then spread over the peers
            if (copy_between(g, j, recv[j] + (size_t)i * blk_words, i, send[i] + (size_t)j * blk_words, blk_words * 8)) return 1;
        }
        G_HIP(hipEventRecord(g->ev_done[j], g->ctx[j]->stream));
        all.push_back(j);
    }
    if (done_all(g, all)) return 1;
    for (uint32_t i = 0; i < n; i++) {
        if (use(g, i)) return 1;
        time_end(g, i);
    }
    return 0;
}

static int all_to_all_sets(lumen_group *g, const lumen_set *const *send, lumen_set *const *recv, const char *stat_name) {
    group_lock lk(g);
    const uint32_t n = (uint32_t)g->ctx.size(), W = g->W;
    std::vector<const u64 *> sp;
    std::vector<u64 *> rp;
    for (uint32_t i = 0; i < n; i++) {
        LM_CHECK(nullptr, send[i] && recv[i], "lumen_group_all_to_all: set %u is NULL", i);
        LM_CHECK(nullptr, send[i]->words == recv[i]->words && send[i]->words == send[0]->words,
                 "lumen_group_all_to_all: send and receive sets differ in size (%zu / %zu words at local rank %u, %zu at 0)",
                 send[i]->words, recv[i]->words, i, send[0]->words);
        LM_CHECK(nullptr, send[i]->count % W == 0 && recv[i]->count % W == 0,
                 "lumen_group_all_to_all: %u / %u ciphertexts are not %u equal blocks", send[i]->count, recv[i]->count, W);
        LM_CHECK(nullptr, (const void *)send[i]->d != (const void *)recv[i]->d, "lumen_group_all_to_all: in-place exchange");
        sp.push_back(send[i]->d), rp.push_back(recv[i]->d);
    }
    if (!send[0]->words) return 0;
    return all_to_all_raw(g, sp, rp, send[0]->words / W, stat_name);
}

extern "C" int lumen_group_all_to_all(lumen_group *g, const lumen_set *const *send, lumen_set *const *recv) {
    LM_CHECK(nullptr, g && send && recv, "lumen_group_all_to_all: NULL argument");
    return all_to_all_sets(g, send, recv, "all_to_all");
}

// ---- fhe.Encode over column-sharded input (include/lumenos_hip.h)
namespace {
// temporaries of a group call: handed back when the call ends, in STREAM ORDER -- their storage returns to the
// owning context's pool behind an event on its stream (lm_set_release_async), so the call returns with its work
// enqueued and the host goes on to enqueue the rescale and the leaf hashing of every rank (lumen_set_destroy would
// wait for every rank's Encode here).  Other ranks' reads of a temporary are ordered before that event: the
// copy transport makes the source's stream wait for every destination (done_all), RCCL sends from the owner's stream.
// ONLY when the call marks itself successful (commit()): on an error path the exchange may have stopped between some
// destinations' copies and done_all(), and then the owner's stream has NOT waited for the other ranks' reads -- an
// event on it would not cover them and the next taker of the block could overwrite it mid-copy.  There every local
// rank is drained first and the blocks go back the blocking way.
struct set_bin {
    lumen_group *g;
    std::vector<std::pair<lumen_ctx *, lumen_set *>> v;
    bool ok = false;
    explicit set_bin(lumen_group *grp) : g(grp) {}
    lumen_set *keep(lumen_ctx *c, lumen_set *s) {
        v.emplace_back(c, s);
        return s;
    }
    void commit() { ok = true; }
    ~set_bin() {
        if (!ok && !v.empty() && !g->poisoned)
            for (lumen_ctx *c : g->ctx) {
                (void)hipSetDevice(c->device);
                lm_sync_all(c);
            }
        for (auto &p : v) {
            (void)hipSetDevice(p.first->device);
            if (ok) {
                lm_set_release_async(p.first, p.second);
            } else if (!g->poisoned) {
                lumen_set_destroy(p.first, p.second);
            } else {
                // half a collective may sit on the streams for good: neither a stream wait nor hipFree (an implicit device
                // synchronisation) may be issued -- the block is leaked with the group that has to be torn down anyway
                p.second->owner = false;
                lumen_set_destroy(nullptr, p.second);
            }
        }
    }
};
} // namespace

extern "C" int lumen_group_encode(lumen_group *g, const lumen_set *const *matrix, const uint64_t *zero_ct,
                                  uint32_t rho_inv, lumen_set **encoded) {
    LM_CHECK(nullptr, g && matrix && zero_ct && encoded, "lumen_group_encode: NULL argument");
    group_lock lk(g);
    const uint32_t n = (uint32_t)g->ctx.size(), W = g->W, logw = g->logw;
    for (uint32_t i = 0; i < n; i++) {
        encoded[i] = nullptr;
        LM_CHECK(nullptr, matrix[i], "lumen_group_encode: matrix block %u is NULL", i);
        LM_CHECK(nullptr, matrix[i]->logw == 0, "lumen_group_encode: block %u is a lane shard, full-width columns are required", i);
        LM_CHECK(nullptr, matrix[i]->count == matrix[0]->count && matrix[i]->nl == matrix[0]->nl && matrix[i]->count > 0,
                 "lumen_group_encode: block %u holds %u columns of %u limbs, block 0 %u of %u", i, matrix[i]->count, matrix[i]->nl,
                 matrix[0]->count, matrix[0]->nl);
    }
    if (W == 1) return lumen_encode(g->ctx[0], matrix[0], zero_ct, rho_inv, &encoded[0]);
    const uint32_t own = matrix[0]->count, cols = own * W, S = cols * rho_inv, nl = matrix[0]->nl;
    const uint32_t N = g->ctx[0]->N, Nw = N >> logw;
    // every rank's slice of the ONE Enc(0): staged through the context's pinned buffer, no host block
    std::vector<u64 *> dzero(n);
    for (uint32_t i = 0; i < n; i++) {
        lumen_ctx *c = g->ctx[i];
        if (use(g, i)) return 1;
        const size_t words = (size_t)2 * nl * Nw;
        u64 *h = (u64 *)lm_stage(c, words * 8);
        dzero[i] = (u64 *)lm_scratch(c, "zero_ct", words * 8);
        if (!h || !dzero[i]) return 1;
        for (uint32_t row = 0; row < 2 * nl; row++)
            memcpy(h + (size_t)row * Nw, zero_ct + (size_t)row * N + (size_t)g->rank[i] * Nw, (size_t)Nw * 8);
        G_HIP(hipMemcpyAsync(dzero[i], h, words * 8, hipMemcpyHostToDevice, c->stream));
        G_HIP(hipEventRecord(c->ev_stage, c->stream));
    }
    set_bin bin(g);
    std::vector<const lumen_set *> a(n);
    std::vector<lumen_set *> b(n);
    // own columns -> W lane blocks -> all-to-all -> the rank's lane shard of ALL columns
    for (uint32_t i = 0; i < n; i++) {
        lumen_set *blk = nullptr, *lanes = nullptr;
        if (use(g, i) || lumen_lanes_split(g->ctx[i], matrix[i], logw, &blk)) return 1;
        bin.keep(g->ctx[i], blk);
        if (lumen_set_create_lanes(g->ctx[i], cols, nl, logw, &lanes)) return 1;
        bin.keep(g->ctx[i], lanes);
        a[i] = blk, b[i] = lanes;
    }
    if (all_to_all_sets(g, a.data(), b.data(), "all_to_all_1")) return 1;
    // Encode on the lane shard, then block h of every shard -> rank h
    for (uint32_t i = 0; i < n; i++) {
        lumen_set *enc = nullptr, *recv = nullptr;
        if (use(g, i) || lm_encode_dev(g->ctx[i], b[i], dzero[i], rho_inv, &enc)) return 1;
        bin.keep(g->ctx[i], enc);
        if (lumen_set_create_lanes(g->ctx[i], S, nl, logw, &recv)) return 1;
        bin.keep(g->ctx[i], recv);
        a[i] = enc, b[i] = recv;
    }
    if (all_to_all_sets(g, a.data(), b.data(), "all_to_all_2")) return 1;
    std::vector<lumen_set *> mine(n, nullptr);
    for (uint32_t i = 0; i < n; i++) {
        if (use(g, i) || lumen_lanes_assemble(g->ctx[i], b[i], &mine[i])) {
            for (uint32_t k = 0; k < i; k++) lumen_set_destroy(g->ctx[k], mine[k]);
            return 1;
        }
    }
    for (uint32_t i = 0; i < n; i++) encoded[i] = mine[i];
    bin.commit();
    return 0; // `bin` gives the temporaries back in stream order: nothing here waits for the device
}

// ---- Commit's exchange: all-gather of the leaf digests
extern "C" int lumen_group_all_gather_digests(lumen_group *g) {
    LM_CHECK(nullptr, g, "lumen_group_all_gather_digests: NULL group");
    group_lock lk(g);
    G_USABLE(g, "lumen_group_all_gather_digests");
    const uint32_t n = (uint32_t)g->ctx.size(), W = g->W;
    uint32_t per = 0;
    std::vector<const uint8_t *> src(n);
    for (uint32_t i = 0; i < n; i++) {
        lumen_ctx *c = g->ctx[i];
        LM_CHECK(nullptr, c->aux_digests, "lumen_group_all_gather_digests: local rank %u has no lumen_leaf_digests_begin job in flight", i);
        if (i == 0) per = c->aux_digests;
        LM_CHECK(nullptr, c->aux_digests == per, "lumen_group_all_gather_digests: local rank %u hashed %u leaves, rank 0 %u", i, c->aux_digests, per);
    }
    g->call_seq++;
    const size_t part = (size_t)per * 32;
    for (uint32_t i = 0; i < n; i++) {
        lumen_ctx *c = g->ctx[i];
        if (use(g, i)) return 1;
        auto it = c->scratch.find("digests_async");
        LM_CHECK(nullptr, it != c->scratch.end() && it->second.first, "digest buffer missing");
        src[i] = (const uint8_t *)it->second.first;
        g->d_digests[i] = (uint8_t *)lm_scratch(c, "group_digests", part * W);
        if (!g->d_digests[i]) return 1;
        // the job ends on the device: the main stream waits for the side stream, the host does not
        c->aux_digests = 0;
        c->aux_lo = c->aux_hi = nullptr;
        G_HIP(hipEventRecord(c->ev_aux, c->stream_aux));
        G_HIP(hipStreamWaitEvent(c->stream, c->ev_aux, 0));
        time_begin(g, "all_gather", i, (uint64_t)part * (W - 1));
    }
    g->n_per_rank = per;
    if (g->transport == LUMEN_TRANSPORT_RCCL) {
        rccl_group_scope grp(g);
        if (grp.begin()) return 1;
        for (uint32_t i = 0; i < n; i++) {
            if (use(g, i)) return 1;
            G_NCCL(g, g->rccl->AllGather(src[i], g->d_digests[i], part, ncclUint8, g->comm[i], g->ctx[i]->stream));
        }
        if (grp.end()) return 1;
    } else {
        if (ready_all(g)) return 1;
        std::vector<uint32_t> all;
        for (uint32_t j = 0; j < n; j++) {
            if (use(g, j) || wait_ready(g, j)) return 1;
            for (uint32_t i = 0; i < n; i++)
                if (copy_between(g, j, g->d_digests[j] + (size_t)i * part, i, src[i], part)) return 1;
            G_HIP(hipEventRecord(g->ev_done[j], g->ctx[j]->stream));
            all.push_back(j);
        }
        if (done_all(g, all)) return 1;
    }
    for (uint32_t i = 0; i < n; i++) {
        if (use(g, i)) return 1;
        time_end(g, i);
    }
    return 0;
}

extern "C" int lumen_group_merkle_root(lumen_group *g, uint8_t root[32]) {
    LM_CHECK(nullptr, g && root, "lumen_group_merkle_root: NULL argument");
    group_lock lk(g);
    LM_CHECK(nullptr, g->n_per_rank && g->d_digests[0], "lumen_group_merkle_root: no gathered digests (lumen_group_all_gather_digests first)");
    if (use(g, 0)) return 1;
    return lumen_merkle_root_device(g->ctx[0], g->d_digests[0], g->n_per_rank * g->W, root);
}

extern "C" int lumen_group_digests(lumen_group *g, uint8_t *digests, size_t cap, uint32_t *n_leaves) {
    LM_CHECK(nullptr, g && digests, "lumen_group_digests: NULL argument");
    group_lock lk(g);
    LM_CHECK(nullptr, g->n_per_rank && g->d_digests[0], "lumen_group_digests: no gathered digests (lumen_group_all_gather_digests first)");
    const size_t bytes = (size_t)g->n_per_rank * g->W * 32;
    LM_CHECK(nullptr, cap >= bytes, "lumen_group_digests: buffer of %zu bytes, %zu needed", cap, bytes);
    if (use(g, 0)) return 1;
    G_HIP(hipMemcpyAsync(digests, g->d_digests[0], bytes, hipMemcpyDeviceToHost, g->ctx[0]->stream));
    G_HIP(hipStreamSynchronize(g->ctx[0]->stream));
    if (n_leaves) *n_leaves = g->n_per_rank * g->W;
    return 0;
}

// ---- the query loop over column-sharded leaves
extern "C" int lumen_group_gather(lumen_group *g, const lumen_set *const *src, const uint32_t *idx, uint32_t nq,
                                  lumen_set **out) {
    LM_CHECK(nullptr, g && src && out && (idx || !nq), "lumen_group_gather: NULL argument");
    *out = nullptr;
    group_lock lk(g);
    G_USABLE(g, "lumen_group_gather");
    const uint32_t n = (uint32_t)g->ctx.size(), W = g->W;
    // One process per GPU: every process derives the send / receive plan from the (n, idx[]) IT was given, and a rank that
    // returned early on bad arguments of its own would leave its peers inside RCCL for good.  So in that form nothing
    // returns before the ranks have AGREED: what is wrong with the local arguments is only noted (`bad`), every rank
    // all-gathers { fingerprint of (n, idx[]), "my arguments are valid" } -- a fixed-size exchange that cannot mismatch --
    // and then all ranks fail together or none does.  (This makes the call block the host in that form; with all ranks
    // in one process there is one idx and the checks return at once.)
    const bool per_rank = n < W && g->transport == LUMEN_TRANSPORT_RCCL;
    std::string bad;
    auto note_bad = [&](const char *fmt, ...) {
        if (!bad.empty()) return;
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        bad = buf;
    };
    for (uint32_t i = 0; i < n; i++) {
        if (!src[i]) note_bad("lumen_group_gather: block %u is NULL", i);
        else if (!src[0] || src[i]->count != src[0]->count || src[i]->nl != src[0]->nl || src[i]->logw != 0)
            note_bad("lumen_group_gather: block %u differs in shape from block 0", i);
    }
    const bool have_root = g->rank[0] == 0;
    if (W == 1) {
        LM_CHECK(nullptr, bad.empty(), "%s", bad.c_str());
        return lumen_gather(g->ctx[0], src[0], idx, nq, out);
    }
    const uint32_t per = bad.empty() ? src[0]->count : 0, nl = bad.empty() ? src[0]->nl : 1;
    const size_t ctw = (size_t)2 * nl * g->ctx[0]->N;
    // who owns what: queries of rank p, in query order
    std::vector<std::vector<uint32_t>> local_idx(W);
    std::vector<uint32_t> perm(nq), off(W + 1, 0);
    for (uint32_t k = 0; k < nq && bad.empty(); k++) {
        if (!per || idx[k] / per >= W) note_bad("lumen_group_gather: column %u out of range (%u x %u columns)", idx[k], W, per);
        else local_idx[idx[k] / per].push_back(idx[k] % per);
    }
    if (bad.empty()) {
        for (uint32_t p = 0; p < W; p++) off[p + 1] = off[p] + (uint32_t)local_idx[p].size();
        std::vector<uint32_t> seen(W, 0);
        for (uint32_t k = 0; k < nq; k++) {
            const uint32_t p = idx[k] / per;
            perm[k] = off[p] + seen[p]++;
        }
    }
    if (per_rank) {
        uint64_t fp = 1469598103934665603ull ^ nq;
        for (uint32_t k = 0; k < nq; k++) fp = (fp ^ idx[k]) * 1099511628211ull;
        std::vector<u64 *> dbuf(n);
        for (uint32_t i = 0; i < n; i++) {
            lumen_ctx *c = g->ctx[i];
            if (use(g, i)) return 1;
            dbuf[i] = (u64 *)lm_scratch(c, "gather_agree", 16 * (size_t)(W + 1));
            u64 *h = (u64 *)lm_stage(c, 16);
            if (!dbuf[i] || !h) return 1;
            h[0] = fp, h[1] = bad.empty() ? 1 : 0;
            G_HIP(hipMemcpyAsync(dbuf[i] + 2 * W, h, 16, hipMemcpyHostToDevice, c->stream));
            G_HIP(hipEventRecord(c->ev_stage, c->stream));
        }
        {
            rccl_group_scope grp(g);
            if (grp.begin()) return 1;
            for (uint32_t i = 0; i < n; i++) {
                if (use(g, i)) return 1;
                G_NCCL(g, g->rccl->AllGather(dbuf[i] + 2 * W, dbuf[i], 2, ncclUint64, g->comm[i], g->ctx[i]->stream));
            }
            if (grp.end()) return 1;
        }
        std::vector<uint64_t> seen(2 * (size_t)W);
        for (uint32_t i = 0; i < n; i++) {
            if (use(g, i)) return 1;
            G_HIP(hipMemcpyAsync(seen.data(), dbuf[i], 16 * (size_t)W, hipMemcpyDeviceToHost, g->ctx[i]->stream));
            G_HIP(hipStreamSynchronize(g->ctx[i]->stream));
            LM_CHECK(nullptr, bad.empty(), "%s", bad.c_str()); // (the peers see this rank's flag and fail with it)
            for (uint32_t p = 0; p < W; p++)
                LM_CHECK(nullptr, seen[2 * p + 1] == 1, "lumen_group_gather: rank %u rejected its arguments (see its own message): no "
                         "rank sends", p);
            for (uint32_t p = 0; p < W; p++)
                LM_CHECK(nullptr, seen[2 * p] == fp, "lumen_group_gather: rank %u was given other query indices than rank %u (every rank "
                         "must pass the same n and idx[])", p, g->rank[i]);
        }
    }
    LM_CHECK(nullptr, bad.empty(), "%s", bad.c_str());
    g->call_seq++;
    set_bin bin(g);
    std::vector<lumen_set *> q(n, nullptr);
    for (uint32_t i = 0; i < n; i++) {
        const auto &li = local_idx[g->rank[i]];
        if (li.empty()) continue;
        if (use(g, i) || lumen_gather(g->ctx[i], src[i], li.data(), (uint32_t)li.size(), &q[i])) return 1;
        bin.keep(g->ctx[i], q[i]);
    }
    lumen_set *stage = nullptr;
    if (have_root) {
        if (use(g, 0) || lumen_set_create(g->ctx[0], nq, nl, &stage)) return 1;
        bin.keep(g->ctx[0], stage);
    }
    for (uint32_t i = 0; i < n; i++) {
        if (use(g, i)) return 1;
        const uint64_t sent = g->rank[i] == 0 ? 0 : (uint64_t)local_idx[g->rank[i]].size() * ctw * 8;
        time_begin(g, "gather_to_root", i, sent);
    }
    if (g->transport == LUMEN_TRANSPORT_RCCL) {
        if (have_root && q[0]) { // rank 0's own columns: a device copy
            if (use(g, 0)) return 1;
            G_HIP(hipMemcpyAsync(stage->d, q[0]->d, q[0]->words * 8, hipMemcpyDeviceToDevice, g->ctx[0]->stream));
        }
        rccl_group_scope grp(g);
        if (grp.begin()) return 1;
        for (uint32_t i = 0; i < n; i++) {
            if (use(g, i)) return 1;
            if (g->rank[i] != 0 && q[i])
                G_NCCL(g, g->rccl->Send(q[i]->d, q[i]->words, ncclUint64, 0, g->comm[i], g->ctx[i]->stream));
        }
        if (have_root) {
            if (use(g, 0)) return 1;
            for (uint32_t p = 1; p < W; p++)
                if (off[p + 1] > off[p])
                    G_NCCL(g, g->rccl->Recv(stage->d + (size_t)off[p] * ctw, (size_t)(off[p + 1] - off[p]) * ctw, ncclUint64, (int)p,
                                            g->comm[0], g->ctx[0]->stream));
        }
        if (grp.end()) return 1;
    } else {
        if (ready_all(g)) return 1;
        if (use(g, 0) || wait_ready(g, 0)) return 1;
        for (uint32_t i = 0; i < n; i++)
            if (q[i] && copy_between(g, 0, stage->d + (size_t)off[i] * ctw, i, q[i]->d, q[i]->words * 8)) return 1;
        G_HIP(hipEventRecord(g->ev_done[0], g->ctx[0]->stream));
        if (done_all(g, {0})) return 1;
    }
    for (uint32_t i = 0; i < n; i++) {
        if (use(g, i)) return 1;
        time_end(g, i);
    }
    if (have_root) {
        if (use(g, 0)) return 1;
        if (lumen_gather(g->ctx[0], stage, perm.data(), nq, out)) return 1;
    }
    bin.commit();
    return 0;
}

// ---- host <-> ranks: every local rank's transfer is enqueued before any is waited for, so the W PCIe links of a
// process that owns W GPUs carry their blocks at the same time (one lumen_set_upload per rank would finish the
// first DMA before it starts the second: 12.9 GB over one link after the other)
bool lm_host_is_pinned(const void *p);
int lm_h2d(lumen_ctx *ctx, void *dev, const void *host, size_t bytes);
int lm_d2h(lumen_ctx *ctx, void *host, const void *dev, size_t bytes, bool wait);

static int group_io(lumen_group *g, lumen_set *const *sets, uint64_t *const *hosts, bool up, const char *what) {
    LM_CHECK(nullptr, g && sets && hosts, "%s: NULL argument", what);
    group_lock lk(g);
    const uint32_t n = (uint32_t)g->ctx.size();
    std::vector<uint32_t> later; // pageable buffers go through the contexts' bounce buffers, one rank after the other
    for (uint32_t i = 0; i < n; i++) {
        LM_CHECK(nullptr, sets[i] && (hosts[i] || !sets[i]->words), "%s: set or host buffer %u is NULL", what, i);
        if (!sets[i]->words) continue;
        if (!lm_host_is_pinned(hosts[i])) {
            later.push_back(i);
            continue;
        }
        if (use(g, i)) return 1;
        const size_t bytes = sets[i]->words * 8;
        if (up) G_HIP(hipMemcpyAsync(sets[i]->d, hosts[i], bytes, hipMemcpyHostToDevice, g->ctx[i]->stream));
        else G_HIP(hipMemcpyAsync(hosts[i], sets[i]->d, bytes, hipMemcpyDeviceToHost, g->ctx[i]->stream));
    }
    for (uint32_t i : later) {
        if (use(g, i)) return 1;
        const size_t bytes = sets[i]->words * 8;
        if (up ? lm_h2d(g->ctx[i], sets[i]->d, hosts[i], bytes) : lm_d2h(g->ctx[i], hosts[i], sets[i]->d, bytes, true)) return 1;
    }
    for (uint32_t i = 0; i < n; i++) { // the host buffers are the caller's: return when they may be reused / hold the data
        if (use(g, i)) return 1;
        G_HIP(hipStreamSynchronize(g->ctx[i]->stream));
    }
    return 0;
}

extern "C" int lumen_group_upload(lumen_group *g, lumen_set *const *sets, const uint64_t *const *hosts) {
    return group_io(g, sets, const_cast<uint64_t *const *>(hosts), true, "lumen_group_upload");
}
extern "C" int lumen_group_download(lumen_group *g, const lumen_set *const *sets, uint64_t *const *hosts) {
    return group_io(g, const_cast<lumen_set *const *>(sets), hosts, false, "lumen_group_download");
}

extern "C" int lumen_group_stats(lumen_group *g, const char *name, double *ms, uint64_t *bytes, uint64_t *calls) {
    LM_CHECK(nullptr, g && name, "lumen_group_stats: NULL argument");
    group_lock lk(g);
    stats_resolve(g);
    stat_entry e;
    auto it = g->stats.find(name);
    if (it != g->stats.end()) e = it->second;
    if (ms) *ms = e.ms;
    if (bytes) *bytes = e.bytes;
    if (calls) *calls = e.calls;
    return 0;
}

extern "C" int lumen_group_stats_reset(lumen_group *g) {
    LM_CHECK(nullptr, g, "lumen_group_stats_reset: NULL group");
    group_lock lk(g);
    stats_resolve(g);
    g->stats.clear();
    return 0;
}
