// Integer-VALU micro-benchmark for gfx950: what a 64-bit modular multiply costs.
// Build: hipcc -O3 --offload-arch=gfx950 ubench.hip -o ubench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;

#define ITER 2048
#define ILP 8

template <int OP>
__global__ void k(u64 *out, u64 seed) {
    u64 x[ILP];
    u32 y[ILP];
    double f[ILP];
    for (int i = 0; i < ILP; i++) {
        x[i] = seed * (threadIdx.x + 1 + i) + blockIdx.x;
        y[i] = (u32)x[i];
        f[i] = (double)x[i];
    }
    const u64 w = seed | 1, wp = seed * 3 + 7, q = (1ull << 58) - 27;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            if (OP == 0) y[i] = y[i] * (u32)w + 1;                          // v_mul_lo_u32 (+add)
            if (OP == 1) y[i] = __umulhi(y[i], (u32)w) + y[i];               // v_mul_hi_u32
            if (OP == 2) x[i] = (u64)(u32)x[i] * (u32)w + x[i];              // v_mad_u64_u32
            if (OP == 3) x[i] = x[i] * w + 1;                                // 64-bit mullo
            if (OP == 4) x[i] = __umul64hi(x[i], wp) + x[i];                 // 64-bit mulhi
            if (OP == 5) x[i] = x[i] * w - __umul64hi(x[i], wp) * q;         // Shoup lazy modmul
            if (OP == 6) f[i] = fma(f[i], 1.0000001, 0.5);                   // v_fma_f64
            if (OP == 7) y[i] = y[i] + (u32)w;                               // v_add_u32 baseline
            if (OP == 8) y[i] = __mul24(y[i], (u32)w) + 1;                   // v_mul_u32_u24 / mad_u32_u24
            if (OP == 9) x[i] = x[i] + w;                                    // 64-bit add
        }
    }
    u64 acc = 0;
    for (int i = 0; i < ILP; i++) acc += x[i] + y[i] + (u64)f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int OP>
double run(const char *name, u64 *d) {
    const int blocks = 256 * 8, threads = 256;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    k<OP><<<blocks, threads>>>(d, 12345);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; r++) k<OP><<<blocks, threads>>>(d, 12345 + r);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    double ops = 5.0 * blocks * threads * (double)ITER * ILP;
    double rate = ops / (ms * 1e-3);
    // cycles per wave-instruction per SIMD at 2.4 GHz: 1024 SIMDs
    double cyc = 1024.0 * 2.4e9 / (rate / 64.0);
    printf("%-28s %8.3f ms  %10.3e lane-ops/s  ~%6.2f cyc/wave-op/SIMD (at 2.4GHz)\n", name, ms / 5, rate, cyc);
    return rate;
}

int main() {
    u64 *d;
    hipMalloc(&d, 256 * 8 * 256 * 8);
    run<7>("v_add_u32", d);
    run<9>("add u64", d);
    run<0>("v_mul_lo_u32+add", d);
    run<1>("v_mul_hi_u32+add", d);
    run<8>("mul24+add", d);
    run<2>("v_mad_u64_u32", d);
    run<3>("mullo64+add", d);
    run<4>("mulhi64+add", d);
    run<5>("shoup lazy modmul", d);
    run<6>("v_fma_f64", d);
    return 0;
}
