// Does streaming bandwidth depend on WHICH block of HBM is streamed?  NB separately allocated blocks of 512 MiB
// (twice the Infinity Cache); per block: read and write bandwidth; for the first 5: copy bandwidth of every ordered
// pair.  Round 6, the "ks_mac lottery": a kernel's time is a deterministic function of where its buffers sit
// (profiles/r06_exp_ks_mac_placement.txt); this asks whether one stream alone already shows it.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_place.hip -o tools/ubench_place
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
int main(int argc, char **argv) {
    const int NB = argc > 1 ? atoi(argv[1]) : 10;
    const size_t bytes = (size_t)512 << 20, n = bytes / 16;
    ulonglong2 *blk[64];
    u64 *o;
    for (int i = 0; i < NB; i++)
        if (hipMalloc(&blk[i], bytes) != hipSuccess) return 1;
    (void)hipMalloc(&o, 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int grid = 2048;
    auto time = [&](int kind, int a, int b) {
        float best = 1e9;
        for (int r = 0; r < 6; r++) {
            (void)hipEventRecord(e0);
            if (kind == 0) k_read<<<grid, 256>>>(blk[a], n, o);
            if (kind == 1) k_write<<<grid, 256>>>(blk[a], n);
            if (kind == 2) k_copy<<<grid, 256>>>(blk[a], blk[b], n);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (r && ms < best) best = ms;
        }
        return (kind == 2 ? 2.0 : 1.0) * bytes / 1e9 / best;
    };
    for (int i = 0; i < NB; i++) time(1, i, 0); // touch
    for (int pass = 0; pass < 2; pass++)
        for (int i = 0; i < NB; i++)
            printf("pass %d block %2d %p  read %.3f TB/s  write %.3f TB/s\n", pass, i, (void *)blk[i], time(0, i, 0), time(1, i, 0));
    const int M = NB < 5 ? NB : 5;
    for (int i = 0; i < M; i++) {
        printf("copy from %d:", i);
        for (int j = 0; j < M; j++) printf(" %s%.3f", i == j ? "*" : "", i == j ? 0.0 : time(2, i, j));
        printf("\n");
    }
    return 0;
}
