// Latency vs throughput of v_mad_u64_u32 on gfx950: dependent chains of multiply-adds with ILP
// independent chains per wave and W waves per SIMD.  Prints cycles per instruction per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 ubench_lat.hip -o ubench_lat
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
typedef unsigned int u32;
#define ITER 4096

template <int ILP, int KIND>
__global__ void k(u64 *out, u32 m, long long *clk) {
    u64 x[4];
    for (int i = 0; i < 4; i++) x[i] = (u64)m * (threadIdx.x + 1 + i) + blockIdx.x;
    u32 y = m | 1;
    long long t0 = wall_clock64();
    long long c0 = clock64();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < ILP; i++) {
                if (KIND == 0) // dependent through the 64-bit addend
                    asm volatile("v_mad_u64_u32 %0, s[96:97], %1, %1, %0" : "+v"(x[i]) : "v"(y) : "s96", "s97");
                if (KIND == 1) // dependent through the 32-bit multiplicand (low word of the previous result)
                    asm volatile("v_mad_u64_u32 %0, s[96:97], %1, %2, %0"
                                 : "+v"(x[i])
                                 : "v"((u32)x[i]), "v"(y)
                                 : "s96", "s97");
                if (KIND == 2) // 32-bit add chain
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(y) : "v"((u32)x[i]));
                if (KIND == 3) // v_lshl_add_u64 chain
                    asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(x[i]) : "v"(x[(i + 1) & 3]));
            }
        }
    }
    long long c1 = clock64();
    long long t1 = wall_clock64();
    u64 acc = y;
    for (int i = 0; i < 4; i++) acc += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        clk[0] = c1 - c0;
        clk[1] = t1 - t0;
    }
}

template <int ILP, int KIND>
void run(const char *name, int waves_per_simd, u64 *d, long long *dclk) {
    const int blocks = 256, threads = 256 * waves_per_simd; // 4 SIMDs per CU
    k<ILP, KIND><<<blocks, threads>>>(d, 12345, dclk);
    hipDeviceSynchronize();
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipEventRecord(a);
    k<ILP, KIND><<<blocks, threads>>>(d, 54321, dclk);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    long long h[2];
    hipMemcpy(h, dclk, 16, hipMemcpyDeviceToHost);
    const double insts_per_wave = (double)ITER * 8 * ILP;
    // shader clock from s_memtime ticks (clock64) per instruction; wall_clock64 runs at 100 MHz
    const double secs = h[1] / 100e6;
    printf("%-34s waves/SIMD=%d ILP=%d  %7.3f ms  clock64 ticks/inst/wave = %6.2f  ns/inst/SIMD = %6.3f\n", name,
           waves_per_simd, ILP, ms, h[0] / insts_per_wave, secs * 1e9 / (insts_per_wave * waves_per_simd));
}

int main() {
    u64 *d;
    long long *dclk;
    hipMalloc(&d, 256 * 1024 * 8);
    hipMalloc(&dclk, 16);
    for (int w = 1; w <= 4; w *= 2) {
        run<1, 0>("mad chain via addend", w, d, dclk);
        run<2, 0>("mad chain via addend", w, d, dclk);
        run<4, 0>("mad chain via addend", w, d, dclk);
        run<1, 1>("mad chain via multiplicand", w, d, dclk);
        run<2, 1>("mad chain via multiplicand", w, d, dclk);
        run<4, 1>("mad chain via multiplicand", w, d, dclk);
        run<1, 2>("v_add_u32 chain", w, d, dclk);
        run<1, 3>("v_lshl_add_u64 chain", w, d, dclk);
        run<2, 3>("v_lshl_add_u64 chain", w, d, dclk);
    }
    return 0;
}
