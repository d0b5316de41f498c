// TEST DOUBLE -- never shipped, never linked into the product.  Built by tests/test_group_rccl.py into
// tests/cpp/fake_rccl/librccl.so.1 and put first on LD_LIBRARY_PATH of a child process that has no other RCCL
// loaded, so that lumenos_amd/csrc/lm_group.hip's dlopen("librccl.so.1") finds THIS file.
//
// Why: a one-GPU box cannot run real RCCL with more than one rank (RCCL refuses two ranks on a device), so the RCCL
// branches of lm_group.hip -- grouped ncclSend / ncclRecv with per-peer pointer arithmetic (send-to-self included),
// ncclAllGather, gather-to-root with per-peer offsets, ncclCommInitAll and ncclCommInitRank -- would otherwise only
// ever execute with a world of one.  This file implements the 12 entry points rccl_load() resolves with NCCL's
// documented point-to-point semantics, inside one process:
//   * ncclSend / ncclRecv match by (source rank, destination rank) in posting order, per communicator world;
//   * a pair must agree on the byte count (real NCCL would hang or corrupt: here it is ncclInvalidArgument, and a
//     receive nobody sends to times out with ncclSystemError instead of hanging the test);
//   * everything is stream-ordered: the copy runs on the receiver's stream behind an event of the sender's stream,
//     and the sender's stream waits for the copy before it goes on (its buffer may be reused afterwards);
//   * ncclGroupStart / End defer the calls of the thread; nothing is matched before the outermost GroupEnd, so
//     sends and receives of one group may be posted in any order, to any peer, self included;
//   * ncclCommInitRank blocks until all ranks of the id have joined (ranks = threads of this process);
//   * ncclAllGather = a send of the same buffer to every rank + a receive from every rank.
// ncclGetVersion answers 99999 so that a test can tell which library the product loaded.
// FAKE_RCCL_FAIL_INIT=1 in the environment makes ncclCommInitAll / ncclCommInitRank fail with ncclInvalidUsage (what
// real RCCL answers when it cannot set up its transports), for the LUMEN_TRANSPORT_AUTO fall-back test.
// fake_rccl_fail_send_after(n) (a test calls it through ctypes) makes one later ncclSend fail: the poisoned-group path.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace {

constexpr int kFakeVersion = 99999;
constexpr int kTimeoutSeconds = 60;

struct letter { // one posted send
    const void *src;
    size_t bytes;
    int src_dev;
    hipEvent_t ready; // recorded on the sender's stream when the send was posted
    hipEvent_t done = nullptr; // recorded on the receiver's stream behind the copy
    bool taken = false, copied = false, failed = false;
};

struct world {
    int n = 0, joined = 0, alive = 0;
    std::map<std::pair<int, int>, std::deque<std::shared_ptr<letter>>> box; // (src, dst) -> posted, not yet taken
    std::vector<hipEvent_t> garbage; // events destroyed with the last communicator (their streams are idle then)
};

std::mutex g_mu;
std::condition_variable g_cv;
std::map<std::string, std::shared_ptr<world>> g_worlds; // by unique id
uint64_t g_next_id = 1;

} // namespace

struct ncclComm {
    std::shared_ptr<world> w;
    int rank = 0, dev = 0;
};

namespace {

struct op {
    enum kind_t { SEND, RECV, ALLGATHER } kind;
    ncclComm *c;
    const void *src;
    void *dst;
    size_t bytes;
    int peer;
    hipStream_t stream;
};

thread_local int t_depth = 0;
thread_local std::vector<op> t_ops;

size_t type_size(ncclDataType_t t) {
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

bool wait_until(std::unique_lock<std::mutex> &lk, const std::function<bool()> &pred) {
    return g_cv.wait_for(lk, std::chrono::seconds(kTimeoutSeconds), pred);
}

struct sent {
    std::shared_ptr<letter> l;
    ncclComm *c;
    hipStream_t stream;
};

ncclResult_t post_send(ncclComm *c, int peer, const void *src, size_t bytes, hipStream_t stream, std::vector<sent> &mine) {
    if (peer < 0 || peer >= c->w->n) return ncclInvalidArgument;
    auto l = std::make_shared<letter>();
    l->src = src, l->bytes = bytes, l->src_dev = c->dev;
    if (hipSetDevice(c->dev) != hipSuccess) return ncclUnhandledCudaError;
    if (hipEventCreateWithFlags(&l->ready, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
    if (hipEventRecord(l->ready, stream) != hipSuccess) return ncclUnhandledCudaError;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        c->w->box[{c->rank, peer}].push_back(l);
        c->w->garbage.push_back(l->ready);
    }
    g_cv.notify_all();
    mine.push_back({l, c, stream});
    return ncclSuccess;
}

ncclResult_t take_recv(ncclComm *c, int peer, void *dst, size_t bytes, hipStream_t stream) {
    if (peer < 0 || peer >= c->w->n) return ncclInvalidArgument;
    std::shared_ptr<letter> l;
    {
        std::unique_lock<std::mutex> lk(g_mu);
        auto &q = c->w->box[{peer, c->rank}];
        if (!wait_until(lk, [&] { return !q.empty(); })) return ncclSystemError; // nobody sends: real NCCL hangs here
        l = q.front();
        q.pop_front();
        l->taken = true;
    }
    ncclResult_t rc = ncclSuccess;
    hipEvent_t done = nullptr;
    if (l->bytes != bytes) rc = ncclInvalidArgument; // the two ranks disagree on the size of this block
    else if (hipSetDevice(c->dev) != hipSuccess || hipStreamWaitEvent(stream, l->ready, 0) != hipSuccess) rc = ncclUnhandledCudaError;
    else {
        hipError_t e = hipSuccess;
        if (bytes)
            e = l->src_dev == c->dev ? hipMemcpyAsync(dst, l->src, bytes, hipMemcpyDeviceToDevice, stream)
                                     : hipMemcpyPeerAsync(dst, c->dev, l->src, l->src_dev, bytes, stream);
        if (e != hipSuccess || hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess ||
            hipEventRecord(done, stream) != hipSuccess)
            rc = ncclUnhandledCudaError;
    }
    {
        std::lock_guard<std::mutex> lk(g_mu);
        l->done = done, l->copied = true, l->failed = rc != ncclSuccess;
        if (done) c->w->garbage.push_back(done);
    }
    g_cv.notify_all();
    return rc;
}

// the outermost ncclGroupEnd (or a call outside any group): post every send, then match every receive, then make
// every sender's stream wait for its receivers
ncclResult_t run(std::vector<op> ops) {
    std::vector<sent> mine;
    ncclResult_t rc = ncclSuccess;
    for (const op &o : ops) {
        if (rc != ncclSuccess) break;
        if (o.kind == op::SEND) rc = post_send(o.c, o.peer, o.src, o.bytes, o.stream, mine);
        else if (o.kind == op::ALLGATHER)
            for (int p = 0; p < o.c->w->n && rc == ncclSuccess; p++) rc = post_send(o.c, p, o.src, o.bytes, o.stream, mine);
    }
    for (const op &o : ops) {
        if (rc != ncclSuccess) break;
        if (o.kind == op::RECV) rc = take_recv(o.c, o.peer, o.dst, o.bytes, o.stream);
        else if (o.kind == op::ALLGATHER)
            for (int p = 0; p < o.c->w->n && rc == ncclSuccess; p++)
                rc = take_recv(o.c, p, (char *)o.dst + (size_t)p * o.bytes, o.bytes, o.stream);
    }
    for (sent &s : mine) {
        std::unique_lock<std::mutex> lk(g_mu);
        if (!wait_until(lk, [&] { return s.l->copied; })) {
            if (rc == ncclSuccess) rc = ncclSystemError; // nobody received what this rank sent
            continue;
        }
        hipEvent_t done = s.l->done;
        const bool failed = s.l->failed;
        lk.unlock();
        if (failed && rc == ncclSuccess) rc = ncclInvalidArgument;
        if (done && (hipSetDevice(s.c->dev) != hipSuccess || hipStreamWaitEvent(s.stream, done, 0) != hipSuccess) && rc == ncclSuccess)
            rc = ncclUnhandledCudaError;
    }
    return rc;
}

ncclResult_t submit(const op &o) {
    if (!o.c || !o.c->w) return ncclInvalidArgument;
    if (t_depth > 0) {
        t_ops.push_back(o);
        return ncclSuccess;
    }
    int dev = -1;
    (void)hipGetDevice(&dev);
    const ncclResult_t rc = run({o});
    if (dev >= 0) (void)hipSetDevice(dev);
    return rc;
}

} // namespace

extern "C" {

ncclResult_t ncclGetVersion(int *version) {
    if (!version) return ncclInvalidArgument;
    *version = kFakeVersion;
    return ncclSuccess;
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof(*id));
    std::lock_guard<std::mutex> lk(g_mu);
    snprintf(id->internal, sizeof(id->internal), "fake-rccl-world-%llu", (unsigned long long)g_next_id++);
    return ncclSuccess;
}

static bool fail_init() {
    const char *e = getenv("FAKE_RCCL_FAIL_INIT");
    return e && *e && *e != '0';
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    if (fail_init()) return ncclInvalidUsage;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return ncclUnhandledCudaError;
    const std::string key(id.internal, strnlen(id.internal, sizeof(id.internal)));
    std::unique_lock<std::mutex> lk(g_mu);
    auto &w = g_worlds[key];
    if (!w) w = std::make_shared<world>(), w->n = nranks;
    if (w->n != nranks) return ncclInvalidArgument;
    std::shared_ptr<world> mine = w;
    mine->joined++, mine->alive++;
    g_cv.notify_all();
    if (!wait_until(lk, [&] { return mine->joined >= mine->n; })) { // a rank never arrived: real NCCL waits for good
        mine->joined--, mine->alive--;
        return ncclSystemError;
    }
    ncclComm *c = new ncclComm();
    c->w = mine, c->rank = rank, c->dev = dev;
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t *comm, int ndev, const int *devlist) {
    if (!comm || ndev < 1) return ncclInvalidArgument;
    if (fail_init()) return ncclInvalidUsage;
    auto w = std::make_shared<world>();
    w->n = w->joined = w->alive = ndev;
    for (int i = 0; i < ndev; i++) {
        ncclComm *c = new ncclComm();
        c->w = w, c->rank = i, c->dev = devlist ? devlist[i] : i;
        comm[i] = c;
    }
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    if (!comm) return ncclSuccess;
    std::vector<hipEvent_t> garbage;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (--comm->w->alive == 0) {
            garbage.swap(comm->w->garbage);
            for (auto it = g_worlds.begin(); it != g_worlds.end();)
                it = it->second == comm->w ? g_worlds.erase(it) : std::next(it);
        }
    }
    for (hipEvent_t e : garbage) (void)hipEventDestroy(e);
    delete comm;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int *count) {
    if (!comm || !count) return ncclInvalidArgument;
    *count = comm->w->n;
    return ncclSuccess;
}

// fault injection for the tests (not an nccl* symbol): the n-th ncclSend from now on (0 = the next one) answers
// ncclInternalError without posting anything -- what lm_group.hip must survive between ncclGroupStart and ncclGroupEnd
static std::atomic<int> g_fail_send_after{-1};
void fake_rccl_fail_send_after(int n) { g_fail_send_after.store(n); }

ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    int left = g_fail_send_after.load();
    while (left >= 0 && !g_fail_send_after.compare_exchange_weak(left, left - 1)) {
    }
    if (left == 0) return ncclInternalError;
    return submit({op::SEND, comm, sendbuff, nullptr, count * type_size(datatype), peer, stream});
}

ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    return submit({op::RECV, comm, nullptr, recvbuff, count * type_size(datatype), peer, stream});
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm,
                           hipStream_t stream) {
    return submit({op::ALLGATHER, comm, sendbuff, recvbuff, sendcount * type_size(datatype), -1, stream});
}

ncclResult_t ncclGroupStart() {
    t_depth++;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth > 0) return ncclSuccess;
    std::vector<op> ops;
    ops.swap(t_ops);
    int dev = -1;
    (void)hipGetDevice(&dev);
    const ncclResult_t rc = run(std::move(ops));
    if (dev >= 0) (void)hipSetDevice(dev);
    return rc;
}

const char *ncclGetErrorString(ncclResult_t result) {
    switch (result) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "fake rccl: a HIP call failed";
    case ncclSystemError: return "fake rccl: timed out waiting for the peer (unmatched send / receive, or a rank that never joined)";
    case ncclInvalidArgument: return "fake rccl: invalid argument (a send and its receive disagree on the size, or a bad rank)";
    case ncclInvalidUsage: return "fake rccl: invalid usage";
    default: return "fake rccl: error";
    }
}

} // extern "C"
