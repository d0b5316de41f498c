// Can the gadget product be accumulated with 64-bit atomic adds from the extension kernel's store phase?
// Throughput of no-return global atomic adds (u64), coalesced (lane-consecutive) and in the 64-byte lane
// stride a run-of-8 storer would produce, over a 235 MB accumulator (64 columns x 28 limbs x 128 KB),
// against plain stores of the same pattern.  6 passes = the six digits accumulating into the same words.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_atomic.hip -o tools/ubench_atomic
#include <hip/hip_runtime.h>

#include <cstdio>
typedef unsigned long long u64;

template <int MODE> // 0: store, 1: atomic add, coalesced; 2: store, 3: atomic add, 8-word runs per lane
__global__ __launch_bounds__(256) void k(u64 *p, size_t n, u64 v) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    if (MODE < 2) {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
            if (MODE == 0) p[i] = v + i;
            else atomicAdd(p + i, v + i);
        }
    } else {
        for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += stride * 8) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (MODE == 2) p[i + k] = v + i;
                else atomicAdd(p + i + k, v + i);
            }
        }
    }
}

template <int MODE>
static void run(const char *name, u64 *p, size_t n) {
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    k<MODE><<<4096, 256>>>(p, n, 1);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int pass = 0; pass < 6; pass++) k<MODE><<<4096, 256>>>(p, n, pass);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%-44s %7.3f ms for 6 passes over %zu MB: %6.2f TB/s of 8-byte updates\n", name, ms, n * 8 >> 20,
           6.0 * n * 8 / ms / 1e9);
}

int main() {
    const size_t n = (size_t)64 * 28 * 16384;
    u64 *p;
    hipMalloc(&p, n * 8);
    hipMemset(p, 0, n * 8);
    run<0>("plain store, lane-consecutive", p, n);
    run<1>("atomic add (no return), lane-consecutive", p, n);
    run<2>("plain store, runs of 8 words per lane", p, n);
    run<3>("atomic add (no return), runs of 8 per lane", p, n);
    return 0;
}
