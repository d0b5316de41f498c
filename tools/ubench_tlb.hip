// Does address translation limit scattered access, and does the VA alignment / mapping of the region change it?
// Every wave reads random 1 KB chunks (64 lanes x 16 B) from a region of R bytes; R = 64 MB .. 32 GB; the region is
// (a) one hipMalloc, (b) hipMemAddressReserve aligned to 1 GB + hipMemCreate + hipMemMap (one physical handle).
// Round 6: looking for the mechanism behind the placement dependence of k_ks_mac (profiles/r06_exp_ks_mac_placement.txt).
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_tlb.hip -o tools/ubench_tlb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;

__global__ void k_scatter(const ulonglong2 *p, u64 chunks, int iters, u64 *out) {
    const u64 wave = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    u64 acc = 0, s = wave * 0x9e3779b97f4a7c15ull + 12345;
    for (int i = 0; i < iters; i++) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const u64 c = (s >> 17) % chunks;
        const ulonglong2 v = p[c * 64 + lane];
        acc += v.x ^ v.y;
    }
    if (acc == 0x1234567) out[0] = acc;
}
int main() {
    u64 *o;
    (void)hipMalloc(&o, 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int grid = 8192, iters = 256;
    for (size_t mb : {64ul, 512ul, 2048ul, 8192ul, 32768ul}) {
        const size_t bytes = mb << 20;
        for (int mode = 0; mode < 2; mode++) {
            void *ptr = nullptr;
            hipMemGenericAllocationHandle_t h;
            if (mode == 0) {
                if (hipMalloc(&ptr, bytes) != hipSuccess) { printf("hipMalloc %zu MB failed\n", mb); continue; }
            } else {
                hipMemAllocationProp prop = {};
                prop.type = hipMemAllocationTypePinned;
                prop.location.type = hipMemLocationTypeDevice;
                prop.location.id = 0;
                size_t gran = 0;
                (void)hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
                if (hipMemAddressReserve(&ptr, bytes, (size_t)1 << 30, nullptr, 0) != hipSuccess) { printf("reserve failed\n"); continue; }
                if (hipMemCreate(&h, bytes, &prop, 0) != hipSuccess) { printf("hipMemCreate %zu MB failed (granularity %zu)\n", mb, gran); continue; }
                if (hipMemMap(ptr, bytes, 0, h, 0) != hipSuccess) { printf("map failed\n"); continue; }
                hipMemAccessDesc acc = {};
                acc.location = prop.location;
                acc.flags = hipMemAccessFlagsProtReadWrite;
                if (hipMemSetAccess(ptr, bytes, &acc, 1) != hipSuccess) { printf("set access failed\n"); continue; }
            }
            (void)hipMemset(ptr, 1, bytes);
            float best = 1e9;
            for (int r = 0; r < 4; r++) {
                (void)hipEventRecord(e0);
                k_scatter<<<grid, 256>>>((const ulonglong2 *)ptr, bytes / 1024, iters, o);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (r && ms < best) best = ms;
            }
            const double gb = (double)grid * 4 * iters * 1024 / 1e9;
            printf("region %6zu MB  %-28s %p  %.3f ms  %.2f TB/s\n", mb, mode ? "VMM, VA aligned to 1 GB" : "hipMalloc", ptr, best, gb / best);
            if (mode == 0) (void)hipFree(ptr);
            else {
                (void)hipMemUnmap(ptr, bytes);
                (void)hipMemRelease(h);
                (void)hipMemAddressFree(ptr, bytes);
            }
        }
    }
    return 0;
}
