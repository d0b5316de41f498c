// What the butterfly arithmetic of the limb NTT costs on its own: the forward stages of a pass
// (lm_fwd_stages<4>: 32 butterflies on 16 register-resident coefficients, hand-scheduled Shoup chain)
// in a loop, no LDS traffic, no global memory -- at the transform's own occupancy (1024 threads and
// 144 KB of LDS per workgroup: one workgroup, 4 waves per SIMD) and at 2 / 1 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 -I lumenos_amd/csrc tools/ubench_bfly.hip -o tools/ubench_bfly
#include <cstdio>

#include "lm_ntt_dev.h"

template <bool UW>
__global__ __launch_bounds__(1024) void k_bfly(u64 *out, const tw_t *tw, u64 q, int iters) {
    extern __shared__ u64 sm[];
    lm_qc c;
    c.q = q, c.nq = 0 - q, c.q3 = 3 * q, c.qinv64 = ~0ull / q;
    u64 e[16];
    for (int k = 0; k < 16; k++) e[k] = (u64)threadIdx.x * 0x9e3779b97f4a7c15ull + k + blockIdx.x;
    lm_twset<4, UW> T;
    T.load(tw, 0, UW ? 0 : threadIdx.x & 63);
    for (int it = 0; it < iters; it++) {
        lm_fwd_stages<4, UW>(e, T, c);
#pragma unroll
        for (int k = 0; k < 16; k++) e[k] = lm_keep(e[k]);
    }
    u64 acc = 0;
    for (int k = 0; k < 16; k++) acc ^= e[k];
    if (acc == 0x1234567) sm[threadIdx.x] = acc; // keep the LDS allocation alive
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <bool UW>
static void run(const char *name, int threads, size_t lds, u64 *out, const tw_t *tw) {
    const int iters = 400, blocks = 256 * 4;
    hipFuncSetAttribute((const void *)k_bfly<UW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    k_bfly<UW><<<blocks, threads, lds>>>(out, tw, (1ull << 58) - 27, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k_bfly<UW><<<blocks, threads, lds>>>(out, tw, (1ull << 58) - 27, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double wave_bfly = (double)blocks * threads / 64 * iters * 32;
    // SIMD-cycles spent per wave-butterfly: 1024 SIMDs; clock unknown -> report ns and cycles at 2.4 GHz
    const double ns = ms * 1e6 * 1024 / wave_bfly;
    printf("%-46s %8.3f ms  %7.2f ns/wave-butterfly/SIMD = %6.1f cycles @2.4GHz (15 VALU instructions)\n", name, ms, ns,
           ns * 2.4);
}

int main() {
    u64 *out;
    tw_t *tw;
    hipMalloc(&out, (size_t)256 * 4 * 1024 * 8);
    hipMalloc(&tw, 4096 * sizeof(tw_t));
    hipMemset(tw, 0x5a, 4096 * sizeof(tw_t));
    run<true>("uniform twiddles, 16 waves/CU (1 WG of 1024)", 1024, 144 * 1024, out, tw);
    run<false>("per-lane twiddles, 16 waves/CU (1 WG of 1024)", 1024, 144 * 1024, out, tw);
    run<true>("uniform twiddles, 8 waves/CU (1 WG of 512)", 512, 144 * 1024, out, tw);
    run<true>("uniform twiddles, 4 waves/CU (1 WG of 256)", 256, 144 * 1024, out, tw);
    run<true>("uniform twiddles, 32 waves/CU? (2 WG of 1024, regs permitting)", 1024, 64 * 1024, out, tw);
    return 0;
}
