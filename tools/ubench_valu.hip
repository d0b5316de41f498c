// Issue cost of the integer VALU instructions the modular arithmetic can be built from, on gfx950:
// 64 back-to-back copies of one instruction over 8 independent accumulators, 4 waves per SIMD (one
// 1024-thread workgroup per CU, as the limb transforms run), no memory traffic.  Prints SIMD cycles per
// wave-instruction at an assumed 2.4 GHz.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o tools/ubench_valu
#include <hip/hip_runtime.h>

#include <cstdio>
typedef unsigned long long u64;
typedef unsigned int u32;

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

// one kernel per instruction: acc[k] are 64-bit, a/b 32-bit lane values, s a uniform
#define KERNEL(NAME, ASM)                                                                          \
    __global__ __launch_bounds__(1024) void NAME(u64 *out, u32 sv, int iters) {                    \
        u64 acc[8];                                                                                \
        u32 a = threadIdx.x * 2654435761u + 1, b = threadIdx.x * 40503u + 7;                       \
        double fa = (double)a, fb = 1.0 + 1e-9 * b;                                                \
        (void)fa, (void)fb;                                                                        \
        for (int k = 0; k < 8; k++) acc[k] = (u64)threadIdx.x * 0x9e3779b97f4a7c15ull + k;         \
        for (int it = 0; it < iters; it++) {                                                       \
            REP64(ASM)                                                                             \
        }                                                                                          \
        u64 r = 0;                                                                                 \
        for (int k = 0; k < 8; k++) r ^= acc[k];                                                   \
        out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;                                    \
    }

#define A_MAD64(k) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b) : "s20", "s21");
#define A_MAD64S(k) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(acc[k]) : "v"(a), "s"(sv) : "s20", "s21");
#define A_MAD64Z(k) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, 0" : "=v"(acc[k]) : "v"(a), "v"(b) : "s20", "s21");
#define A_MAD64ONE(k) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, 1, %0" : "+v"(acc[k]) : "v"(a) : "s20", "s21");
#define A_MULLO(k) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(((u32 *)&acc[k])[0]) : "v"(((u32 *)&acc[k])[0]), "v"(b));
#define A_MULHI(k) asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(((u32 *)&acc[k])[0]) : "v"(((u32 *)&acc[k])[0]), "v"(b));
#define A_ADD32(k) asm volatile("v_add_u32 %0, %1, %2" : "=v"(((u32 *)&acc[k])[0]) : "v"(((u32 *)&acc[k])[0]), "v"(b));
#define A_ADD3(k) asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(((u32 *)&acc[k])[0]) : "v"(((u32 *)&acc[k])[0]), "v"(b), "v"(a));
#define A_ADDCO(k) asm volatile("v_add_co_u32_e32 %0, vcc, %1, %2" : "=v"(((u32 *)&acc[k])[0]) : "v"(((u32 *)&acc[k])[0]), "v"(b) : "vcc");
#define A_ADDC(k) asm volatile("v_addc_co_u32_e32 %0, vcc, %1, %2, vcc" : "=v"(((u32 *)&acc[k])[0]) : "v"(((u32 *)&acc[k])[0]), "v"(b) : "vcc");
#define A_LSHLADD64(k) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(acc[k]) : "v"(acc[(k + 1) & 7]));
#define A_MAD24(k) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(((u32 *)&acc[k])[0]) : "v"(a), "v"(b));
#define A_MUL24(k) asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(((u32 *)&acc[k])[0]) : "v"(((u32 *)&acc[k])[0]), "v"(b));
#define A_MULHI24(k) asm volatile("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(((u32 *)&acc[k])[0]) : "v"(((u32 *)&acc[k])[0]), "v"(b));
#define A_MADU32(k) asm volatile("v_mad_u32_u16 %0, %1, %2, %0" : "+v"(((u32 *)&acc[k])[0]) : "v"(a), "v"(b));
#define A_FMA64(k) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(*(double *)&acc[k]) : "v"(fa), "v"(fb));
#define A_MULF64(k) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(*(double *)&acc[k]) : "v"(fb));
#define A_ADDF64(k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(*(double *)&acc[k]) : "v"(fb));
#define A_FMA32(k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(((float *)&acc[k])[0]) : "v"(a), "v"(b));
#define A_PKFMA32(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(acc[k]) : "v"(acc[(k + 1) & 7]));
#define A_LSHR64(k) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(acc[k]));
#define A_MOV(k) asm volatile("v_mov_b32 %0, %1" : "=v"(((u32 *)&acc[k])[0]) : "v"(((u32 *)&acc[(k + 1) & 7])[1]));
#define A_CNDMASK(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(((u32 *)&acc[k])[0]) : "v"(b) : "vcc");
#define A_CMP64(k) asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(acc[k]), "v"(acc[(k + 1) & 7]) : "vcc");
#define A_SUB64(k) asm volatile("v_sub_co_u32_e32 %0, vcc, %0, %1" : "+v"(((u32 *)&acc[k])[0]) : "v"(b) : "vcc");
#define A_DOT4(k) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(((u32 *)&acc[k])[0]) : "v"(a), "v"(b));
#define A_CND64(k) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[22:23]" : "+v"(((u32 *)&acc[k])[0]) : "v"(b) : "s22", "s23");
#define A_CMPCND(k) asm volatile("v_cmp_lt_u32_e32 vcc, %1, %0\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(((u32 *)&acc[k])[0]) : "v"(b) : "vcc");
#define A_SUBCND(k) asm volatile("v_sub_co_u32_e32 %0, vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(((u32 *)&acc[k])[0]) : "v"(b) : "vcc");
#define A_CSUB64(k) asm volatile("v_sub_co_u32_e32 v80, vcc, %0, %2\n\tv_subb_co_u32_e32 v81, vcc, %1, %3, vcc\n\tv_cndmask_b32_e32 %0, v80, %0, vcc\n\tv_cndmask_b32_e32 %1, v81, %1, vcc" : "+v"(((u32 *)&acc[k])[0]), "+v"(((u32 *)&acc[k])[1]) : "v"(a), "v"(b) : "vcc", "v80", "v81");
#define A_CSUB64M(k) asm volatile("v_sub_co_u32_e32 v80, vcc, %0, %2\n\tv_subb_co_u32_e32 v81, vcc, %1, %3, vcc\n\tv_ashrrev_i32_e32 v82, 31, v81\n\tv_and_b32_e32 v83, v82, %2\n\tv_and_b32_e32 v82, v82, %3\n\tv_add_co_u32_e32 %0, vcc, v80, v83\n\tv_addc_co_u32_e32 %1, vcc, v81, v82, vcc" : "+v"(((u32 *)&acc[k])[0]), "+v"(((u32 *)&acc[k])[1]) : "v"(a), "v"(b) : "vcc", "v80", "v81", "v82", "v83");
#define A_AND(k) asm volatile("v_and_b32_e32 %0, %0, %1" : "+v"(((u32 *)&acc[k])[0]) : "v"(b));
#define A_ASHR(k) asm volatile("v_ashrrev_i32_e32 %0, 3, %0" : "+v"(((u32 *)&acc[k])[0]));
#define A_MIN(k) asm volatile("v_min_u32_e32 %0, %0, %1" : "+v"(((u32 *)&acc[k])[0]) : "v"(b));
#define A_MADI64(k) asm volatile("v_mad_i64_i32 %0, s[20:21], %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b) : "s20", "s21");

KERNEL(k_mad64, A_MAD64)
KERNEL(k_mad64s, A_MAD64S)
KERNEL(k_mad64z, A_MAD64Z)
KERNEL(k_mad64one, A_MAD64ONE)
KERNEL(k_madi64, A_MADI64)
KERNEL(k_mullo, A_MULLO)
KERNEL(k_mulhi, A_MULHI)
KERNEL(k_add32, A_ADD32)
KERNEL(k_add3, A_ADD3)
KERNEL(k_addco, A_ADDCO)
KERNEL(k_addc, A_ADDC)
KERNEL(k_lshladd64, A_LSHLADD64)
KERNEL(k_mad24, A_MAD24)
KERNEL(k_mul24, A_MUL24)
KERNEL(k_mulhi24, A_MULHI24)
KERNEL(k_madu16, A_MADU32)
KERNEL(k_fma64, A_FMA64)
KERNEL(k_mulf64, A_MULF64)
KERNEL(k_addf64, A_ADDF64)
KERNEL(k_fma32, A_FMA32)
KERNEL(k_pkfma32, A_PKFMA32)
KERNEL(k_lshr64, A_LSHR64)
KERNEL(k_mov, A_MOV)
KERNEL(k_cndmask, A_CNDMASK)
KERNEL(k_cmp64, A_CMP64)
KERNEL(k_dot4, A_DOT4)
KERNEL(k_cnd64, A_CND64)
KERNEL(k_cmpcnd, A_CMPCND)
KERNEL(k_subcnd, A_SUBCND)
KERNEL(k_csub64, A_CSUB64)
KERNEL(k_csub64m, A_CSUB64M)
KERNEL(k_and, A_AND)
KERNEL(k_ashr, A_ASHR)
KERNEL(k_min, A_MIN)

typedef void (*kern_t)(u64 *, u32, int);
static void run(const char *name, kern_t k, int threads, u64 *out) {
    const int iters = 2000, blocks = 256;
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, 12345u, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, 12345u, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double wave_instr = (double)blocks * threads / 64 * iters * 64;
    const double ns = ms * 1e6 * 1024 / wave_instr; // 1024 SIMDs
    printf("%-28s %4d thr/WG  %8.3f ms  %6.2f ns = %5.1f cycles @2.4GHz per wave-instruction per SIMD\n", name, threads, ms,
           ns, ns * 2.4);
}

int main() {
    u64 *out;
    hipMalloc(&out, (size_t)256 * 1024 * 8);
#define RUN(K) run(#K, K, 1024, out);
    RUN(k_mad64) RUN(k_mad64s) RUN(k_mad64z) RUN(k_mad64one) RUN(k_madi64) RUN(k_mullo) RUN(k_mulhi) RUN(k_add32) RUN(k_add3)
    RUN(k_addco) RUN(k_addc) RUN(k_lshladd64) RUN(k_mad24) RUN(k_mul24) RUN(k_mulhi24) RUN(k_madu16) RUN(k_fma64)
    RUN(k_mulf64) RUN(k_addf64) RUN(k_fma32) RUN(k_pkfma32) RUN(k_lshr64) RUN(k_mov) RUN(k_cndmask) RUN(k_cnd64) RUN(k_cmpcnd) RUN(k_subcnd) RUN(k_csub64) RUN(k_csub64m) RUN(k_and) RUN(k_ashr) RUN(k_min) RUN(k_cmp64) RUN(k_dot4)
    run("k_mad64 (1 wave/SIMD)", k_mad64, 256, out);
    run("k_add32 (1 wave/SIMD)", k_add32, 256, out);
    run("k_fma64 (1 wave/SIMD)", k_fma64, 256, out);
    return 0;
}
