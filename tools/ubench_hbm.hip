// Achievable HBM bandwidth on this box: streaming read (sum), write (fill) and copy of 2 GiB with
// 16-byte accesses.  Build: hipcc -O3 --offload-arch=gfx950 ubench_hbm.hip -o ubench_hbm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;

__global__ void k_read(const ulonglong2 *p, size_t n, u64 *out) {
    u64 acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const ulonglong2 v = p[i];
        acc += v.x ^ v.y;
    }
    if (acc == 0x1234567) out[0] = acc;
}
__global__ void k_write(ulonglong2 *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_ulonglong2(i, i);
}
__global__ void k_copy(const ulonglong2 *a, ulonglong2 *b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        b[i] = a[i];
}
int main() {
    const size_t bytes = (size_t)2 << 30, n = bytes / 16;
    ulonglong2 *a, *b;
    u64 *o;
    (void)hipMalloc(&a, bytes);
    (void)hipMalloc(&b, bytes);
    (void)hipMalloc(&o, 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int grid : {256 * 8, 256 * 16, 256 * 32}) {
        for (int kind = 0; kind < 3; kind++) {
            float best = 1e9;
            for (int r = 0; r < 6; r++) {
                (void)hipEventRecord(e0);
                if (kind == 0) k_read<<<grid, 256>>>(a, n, o);
                if (kind == 1) k_write<<<grid, 256>>>(b, n);
                if (kind == 2) k_copy<<<grid, 256>>>(a, b, n);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (r && ms < best) best = ms;
            }
            const double gb = (kind == 2 ? 2.0 : 1.0) * bytes / 1e9;
            printf("grid=%5d %-5s %.3f ms  %.2f TB/s\n", grid, kind == 0 ? "read" : kind == 1 ? "write" : "copy", best,
                   gb / best);
        }
    }
    return 0;
}
