// Where the time of a forward limb transform goes, phase by phase, under production conditions: the
// plain forward kernel (k_limb_ntt<14, false>: one 1024-thread workgroup per transform, 144 KB of LDS)
// re-stated pass by pass with a clock stamp per wave at every phase boundary, run over 12288 limbs
// (1.6 GB, 48 rounds of the 256 CUs).  Prints, averaged over workgroups: the length of every phase,
// the spread between the first and the last wave of a workgroup, and the gap on a CU between the last
// stamp of one workgroup and the first stamp of the next (dispatch + kernel-argument loads).
// Build: hipcc -O3 --offload-arch=gfx950 -I lumenos_amd/csrc tools/ubench_phases.hip -o tools/ubench_phases
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>

#include "lm_ntt_dev.h"

#define NSTAMP 8
struct rec_t {
    unsigned long long t[NSTAMP];
    unsigned hw, xcc;
};

// the constant 100 MHz clock (s_memrealtime): the shader cycle counters of the eight XCDs are not aligned
__device__ __forceinline__ unsigned long long now() { return wall_clock64(); }
#define STAMP(k)                                                                 \
    do {                                                                         \
        if (rec && (tid & 63) == 0) rec[(size_t)item * 16 + (tid >> 6)].t[k] = now(); \
    } while (0)

// first pass split in two: butterflies in registers, then the LDS write -- so that a persistent
// workgroup can put its barrier between them and a wave that is done with the previous transform loads
// and computes while the slower waves finish
template <int LOGN, int R, class Loader>
__device__ __forceinline__ void fwd_first_regs(u64 *e, const tw_t *tw, const lm_qc &c, uint32_t tid, Loader &ld) {
    constexpr uint32_t log_tl = LOGN - R;
    lm_twset<R, true> T;
    T.load(tw, 0, 0);
#pragma unroll
    for (int k = 0; k < (1 << R); k++) e[k] = ld(tid + ((uint32_t)k << log_tl));
    lm_fwd_stages<R, true>(e, T, c);
}
template <int LOGN, int R>
__device__ __forceinline__ void fwd_first_write(u64 *s, const u64 *e, uint32_t tid) {
    constexpr uint32_t log_tl = LOGN - R;
#pragma unroll
    for (int k = 0; k < (1 << R); k++) s[LM_PAD(tid + ((uint32_t)k << log_tl))] = e[k];
}

// wave-local pass of R stages starting at stage S0 on one work item, twiddles given
template <int LOGN, int R, int S0, bool UW>
__device__ __forceinline__ void mid_item(u64 *s, const lm_twset<R, UW> &T, const lm_qc &c, uint32_t w) {
    constexpr uint32_t log_tl = LOGN - S0 - R;
    const uint32_t blk = w >> log_tl, off = w & ((1u << log_tl) - 1);
    const uint32_t base = (blk << (log_tl + R)) + off;
    u64 e[1 << R];
#pragma unroll
    for (int k = 0; k < (1 << R); k++) e[k] = s[LM_PAD(base + ((uint32_t)k << log_tl))];
    lm_fwd_stages<R, UW>(e, T, c);
#pragma unroll
    for (int k = 0; k < (1 << R); k++) s[LM_PAD(base + ((uint32_t)k << log_tl))] = e[k];
}

template <int LOGN, bool PERSIST, bool OVERLAP = false, bool PREFETCH = false, bool TWPIPE = false>
__global__ __launch_bounds__(1024) void k_fwd(const u64 *src, u64 *dst, uint32_t nlimbs, u64 q, u64 qinv64,
                                              const tw_t *__restrict__ tw, rec_t *rec) {
    extern __shared__ __attribute__((aligned(16))) u64 sm[];
    constexpr uint32_t N = 1u << LOGN;
    const uint32_t tid0 = threadIdx.x;
    lm_qc c;
    c.q = q, c.nq = 0 - q, c.q3 = 3 * q, c.qinv64 = qinv64;
    u64 pf[16]; // PREFETCH: the next transform's coefficients, requested while the current one is in its wave-local passes
    if (PREFETCH) {
#pragma unroll
        for (int k = 0; k < 16; k++) pf[k] = src[(size_t)blockIdx.x * N + tid0 + ((uint32_t)k << (LOGN - 4))];
    }
    for (uint32_t item = blockIdx.x; item < nlimbs; item += PERSIST ? gridDim.x : nlimbs) {
        // per-iteration copy of the lane index the compiler cannot see through: otherwise every LDS
        // address and twiddle index of the transform is hoisted out of the loop and spilled
        uint32_t tid = tid0;
        asm volatile("" : "+v"(tid));
        STAMP(0);
        const u64 *p = src + (size_t)item * N;
        u64 *o = dst + (size_t)item * N;
        auto ld = [&](uint32_t i) { return p[i]; };
        auto st = [&](uint32_t i0, const u64 *v, int count) {
            u64 r[8];
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (k < count) r[k] = lm_reduce_s(v[k], c.q, c.nq, c.qinv64);
            lm_store_run(o, i0, r, count);
        };
        if (PREFETCH) {
            u64 e[16];
            lm_twset<4, true> T;
            T.load(tw, 0, 0);
#pragma unroll
            for (int k = 0; k < 16; k++) e[k] = pf[k];
            lm_fwd_stages<4, true>(e, T, c);
            __syncthreads(); // the previous item's last pass still reads the LDS
            fwd_first_write<LOGN, 4>(sm, e, tid);
        } else if (OVERLAP) {
            u64 e[16];
            fwd_first_regs<LOGN, 4>(e, tw, c, tid, ld);
            __syncthreads(); // the previous item's last pass still reads the LDS
            fwd_first_write<LOGN, 4>(sm, e, tid);
        } else {
            if (PERSIST) __syncthreads();
            lm_fwd_first<LOGN, 4, true>(sm, tw, c, tid, ld);
        }
        STAMP(1);
        if (TWPIPE) {
            // every twiddle set is requested a whole work item (or a barrier wait) before its butterflies:
            // pass 1's (wave-uniform, SGPRs) before the barrier, pass 2's two per-lane sets under pass 1,
            // pass 3's under pass 2 -- in the registers pass 2 frees item by item
            using D4 = lm_deal<LOGN, 4>;
            using D3 = lm_deal<LOGN, 3>;
            lm_twset<4, true> T1;
            T1.load(tw, 4, D4::local(tid, 0) >> (LOGN - 8));
            __syncthreads();
            STAMP(2);
            lm_twset<3, false> A, B;
            A.load(tw, 8, D3::local(tid, 0) >> (LOGN - 11));
            B.load(tw, 8, D3::local(tid, 1) >> (LOGN - 11));
            mid_item<LOGN, 4, 4, true>(sm, T1, c, D4::local(tid, 0));
            lm_wave_sync();
            STAMP(3);
            mid_item<LOGN, 3, 8, false>(sm, A, c, D3::local(tid, 0));
            A.load(tw, 11, D3::local(tid, 0));
            mid_item<LOGN, 3, 8, false>(sm, B, c, D3::local(tid, 1));
            B.load(tw, 11, D3::local(tid, 1));
            lm_wave_sync();
            STAMP(4);
#pragma unroll
            for (uint32_t m = 0; m < 2; m++) {
                const uint32_t base = D3::local(tid, m) << 3;
                u64 e[8];
#pragma unroll
                for (int k = 0; k < 8; k++) e[k] = sm[LM_PAD(base + k)];
                lm_fwd_stages<3, false>(e, m ? B : A, c);
                st(base, e, 8);
            }
            STAMP(5);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            STAMP(6);
            if ((tid & 63) == 0 && rec) {
                rec_t &r = rec[(size_t)item * 16 + (tid >> 6)];
                r.hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
                r.xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));
            }
            continue;
        }
        __syncthreads();
        STAMP(2);
        if (PREFETCH && item + gridDim.x < nlimbs) {
            const u64 *pn = src + (size_t)(item + gridDim.x) * N;
#pragma unroll
            for (int k = 0; k < 16; k++) pf[k] = pn[tid + ((uint32_t)k << (LOGN - 4))];
        }
#ifdef UB_SETPRIO // waves that lag get the VALU first: the waves of a SIMD finish together instead of one by one
        __builtin_amdgcn_s_setprio(3);
#endif
        lm_fwd_mid<LOGN, 4, 4>(sm, tw, c, tid);
        lm_wave_sync();
        STAMP(3);
#ifdef UB_SETPRIO
        __builtin_amdgcn_s_setprio(2);
#endif
        lm_fwd_mid<LOGN, 3, 8>(sm, tw, c, tid);
        lm_wave_sync();
        STAMP(4);
#ifdef UB_SETPRIO
        __builtin_amdgcn_s_setprio(1);
#endif
        lm_fwd_last<LOGN, 3>(sm, tw, c, tid, st);
#ifdef UB_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        STAMP(5);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(6);
        if ((tid & 63) == 0 && rec) {
            rec_t &r = rec[(size_t)item * 16 + (tid >> 6)];
            r.hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_REG_HW_ID, all 32 bits
            r.xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); // HW_REG_XCC_ID
        }
    }
}

static u64 mulmod(u64 a, u64 b, u64 q) { return (u64)((unsigned __int128)a * b % q); }
static u64 powmod(u64 a, u64 e, u64 q) {
    u64 r = 1;
    for (; e; e >>= 1, a = mulmod(a, a, q))
        if (e & 1) r = mulmod(r, a, q);
    return r;
}

template <bool PERSIST, bool OVERLAP = false, bool PREFETCH = false, bool TWPIPE = false>
static void run(const char *name, const u64 *src, u64 *dst, uint32_t nlimbs, u64 q, const tw_t *tw, rec_t *drec,
                uint32_t grid) {
    const size_t lds = lm_fwd_lds(14);
    hipFuncSetAttribute((const void *)k_fwd<14, PERSIST, OVERLAP, PREFETCH, TWPIPE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    k_fwd<14, PERSIST, OVERLAP, PREFETCH, TWPIPE><<<grid, 1024, lds>>>(src, dst, nlimbs, q, ~0ull / q, tw, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k_fwd<14, PERSIST, OVERLAP, PREFETCH, TWPIPE><<<grid, 1024, lds>>>(src, dst, nlimbs, q, ~0ull / q, tw, nullptr);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms0;
    hipEventElapsedTime(&ms0, a, b);
    hipEventRecord(a);
    k_fwd<14, PERSIST, OVERLAP, PREFETCH, TWPIPE><<<grid, 1024, lds>>>(src, dst, nlimbs, q, ~0ull / q, tw, drec);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    std::vector<rec_t> rec((size_t)nlimbs * 16);
    hipMemcpy(rec.data(), drec, rec.size() * sizeof(rec_t), hipMemcpyDeviceToHost);
    // clock rate of the cycle counter from the span of all stamps against the event time
    unsigned long long tmin = ~0ull, tmax = 0;
    for (auto &r : rec) tmin = std::min(tmin, r.t[0]), tmax = std::max(tmax, r.t[6]);
    const double mhz = 100.0;
    printf("   (span of all stamps: %.3f ms)\n", (double)(tmax - tmin) / mhz / 1e3);
    printf("== %s: %.3f ms untraced, %.3f ms traced, %u limbs -> %.2f M transforms/s; counter %.0f MHz\n", name, ms0, ms,
           nlimbs, nlimbs / ms0 / 1e3, mhz);
    const char *ph[6] = {"load + pass 0 (4 stages) + LDS write", "barrier", "pass 1 (4 stages, LDS r/w)",
                         "pass 2 (3 stages, LDS r/w)", "pass 3 (3 stages) + reduce + store issue", "store drain"};
    double sum[6] = {0}, wg_total = 0, spread_in = 0, spread_out = 0;
    for (uint32_t i = 0; i < nlimbs; i++) {
        unsigned long long first = ~0ull, last = 0, s0 = ~0ull, s1 = 0;
        for (int w = 0; w < 16; w++) {
            const rec_t &r = rec[(size_t)i * 16 + w];
            for (int k = 0; k < 6; k++) sum[k] += (double)(r.t[k + 1] - r.t[k]);
            first = std::min(first, r.t[0]), last = std::max(last, r.t[6]);
            s0 = std::min(s0, r.t[6]), s1 = std::max(s1, r.t[0]);
        }
        wg_total += (double)(last - first);
        spread_in += (double)(s1 - first), spread_out += (double)(last - s0);
    }
    const double us = 1.0 / mhz;
    for (int k = 0; k < 6; k++) printf("   %-44s %7.2f us (mean over waves)\n", ph[k], sum[k] / (nlimbs * 16.0) * us);
    printf("   workgroup first stamp -> last stamp            %7.2f us; wave start spread %.2f us, end spread %.2f us\n",
           wg_total / nlimbs * us, spread_in / nlimbs * us, spread_out / nlimbs * us);
    if (!PERSIST) {
        // gap between consecutive workgroups of one CU: key = (xcc, se, cu) from HW_ID
        std::map<unsigned, std::vector<std::pair<unsigned long long, unsigned long long>>> by_cu;
        for (uint32_t i = 0; i < nlimbs; i++) {
            unsigned long long first = ~0ull, last = 0;
            for (int w = 0; w < 16; w++) {
                const rec_t &r = rec[(size_t)i * 16 + w];
                first = std::min(first, r.t[0]), last = std::max(last, r.t[6]);
            }
            const rec_t &r = rec[(size_t)i * 16];
            const unsigned cu = (r.hw >> 8) & 0xF, sh = (r.hw >> 12) & 1, se = (r.hw >> 13) & 0x7;
            by_cu[(r.xcc & 0xF) << 16 | se << 8 | sh << 4 | cu].push_back({first, last});
        }
        double gap = 0;
        size_t n = 0;
        for (auto &kv : by_cu) {
            auto &v = kv.second;
            std::sort(v.begin(), v.end());
            for (size_t k = 1; k < v.size(); k++)
                if (v[k].first > v[k - 1].second) gap += (double)(v[k].first - v[k - 1].second), n++;
        }
        printf("   %zu CUs seen; gap between workgroups on a CU       %7.2f us (mean of %zu)\n", by_cu.size(),
               n ? gap / n * us : 0.0, n);
    }
}

int main() {
    const uint32_t N = 1 << 14, nlimbs = 12288;
    // a 58-bit NTT prime: search downwards from 2^58 for q = 1 mod 2N
    u64 q = (1ull << 58) + 1;
    auto is_prime = [](u64 n) {
        for (u64 a : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
            if (n % a == 0) return n == a;
            u64 d = n - 1;
            int r = 0;
            while (!(d & 1)) d >>= 1, r++;
            u64 x = powmod(a, d, n);
            if (x == 1 || x == n - 1) continue;
            bool comp = true;
            for (int i = 1; i < r && comp; i++) {
                x = mulmod(x, x, n);
                if (x == n - 1) comp = false;
            }
            if (comp) return false;
        }
        return true;
    };
    do q -= 2 * N; while (!is_prime(q));
    // any table of valid (w, floor(w * 2^64 / q)) pairs exercises the same instructions
    std::vector<tw_t> tw(N);
    u64 g = 3;
    for (uint32_t i = 0; i < N; i++) {
        g = mulmod(g, 0x9e3779b97f4a7c15ull % q, q);
        tw[i].w = g;
        tw[i].wp = (u64)(((unsigned __int128)g << 64) / q);
    }
    u64 *src, *dst;
    tw_t *dtw;
    rec_t *drec;
    hipMalloc(&src, (size_t)nlimbs * N * 8);
    hipMalloc(&dst, (size_t)nlimbs * N * 8);
    hipMalloc(&dtw, N * sizeof(tw_t));
    hipMalloc(&drec, (size_t)nlimbs * 16 * sizeof(rec_t));
    hipMemset(src, 0x11, (size_t)nlimbs * N * 8);
    hipMemcpy(dtw, tw.data(), N * sizeof(tw_t), hipMemcpyHostToDevice);
    run<false>("one workgroup per transform", src, dst, nlimbs, q, dtw, drec, nlimbs);
    run<true>("persistent, 256 workgroups", src, dst, nlimbs, q, dtw, drec, 256);
    run<true, true>("persistent, 256 workgroups, barrier between pass-0 butterflies and LDS write", src, dst, nlimbs, q, dtw,
                    drec, 256);
    run<true, false, true>("persistent, 256 workgroups, next transform's loads issued after the barrier", src, dst, nlimbs, q,
                           dtw, drec, 256);
    run<false, false, false, true>("one workgroup per transform, twiddles requested a work item ahead across passes", src, dst,
                                   nlimbs, q, dtw, drec, nlimbs);
    run<false>("one workgroup per transform (again)", src, dst, nlimbs, q, dtw, drec, nlimbs);
    return 0;
}
