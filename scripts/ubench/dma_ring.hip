// Microbenchmark 2: contiguous blocked traffic mix (A: shared-by-4 HBM stream, B: 384 KB query tile),
// chunk = CI DMA instr per wave (CI=8: 64 KB/chunk/CU, CI=4: 32 KB), ring of NSLOT LDS slots,
// INFL chunks kept in flight across the barrier.
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ inline void glds16(const char* g, char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <int CI, int NSLOT, int INFL>
__global__ __launch_bounds__(512) void ring_kernel(const char* src, size_t region_bytes, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t g = blockIdx.x;
    if ((gridDim.x & 7u) == 0) g = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const char* pa = src + (size_t)(g / 4) * region_bytes;
    const char* pq = src + ((size_t)5 << 30) + (size_t)(g % 4) * ((size_t)3 << 17);
    constexpr int CB = CI * 8 * 1024;   // bytes per chunk per WG
    constexpr int HB = CB / 2;          // half for A, half for B
    int issued = 0;
    auto issue = [&](int it) {
        char* dst = smem + (it % NSLOT) * CB;
#pragma unroll
        for (int j = 0; j < CI; ++j) {
            const size_t o = (size_t)wave * (HB / 8) + (size_t)(j % (CI / 2)) * 1024 + lane * 16;
            if (j < CI / 2) glds16(pa + (((size_t)it * HB + o) % region_bytes), dst + wave * (HB / 8) + (j % (CI / 2)) * 1024);
            else glds16(pq + (((size_t)(it % (393216 / HB)) * HB + o)), dst + HB + wave * (HB / 8) + (j % (CI / 2)) * 1024);
        }
        ++issued;
    };
    for (int p = 0; p < INFL && p < iters; ++p) issue(p);
    for (int it = 0; it < iters; ++it) {
        if (issued < iters) issue(issued);
        // wait until chunk `it` has landed: chunks issued after it = issued - it - 1
        const int later = issued - it - 1;
        if (later >= 3) { if (CI == 8) asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
        else if (later == 2) { if (CI == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        else if (later == 1) { if (CI == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (iters < 0) sink[0] = smem[tid];
}
template <int CI, int NSLOT, int INFL>
void run(const char* d, float* sink, const char* name) {
    auto k = ring_kernel<CI, NSLOT, INFL>;
    const int lds = NSLOT * CI * 8 * 1024;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int W = 256; const int iters = 240 * 8 / CI; float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(W), dim3(512), lds, 0, d, (size_t)24 << 20, iters, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    double bytes = (double)W * iters * CI * 8 * 1024;
    printf("%-44s lds=%3dKB %8.3f ms %7.2f TB/s %6.1f GB/s/CU %6.2f us/64KB\n", name, lds / 1024, best,
           bytes / best / 1e9, bytes / best / 1e6 / W, best * 1e3 / iters * 8 / CI);
}
int main() {
    char* d; float* sink; hipMalloc(&d, (size_t)6 << 30); hipMemset(d, 1, (size_t)6 << 30); hipMalloc(&sink, 4);
    run<8, 2, 1>(d, sink, "64KB chunks, 2 slots, 1 in flight (v3)");
    run<8, 2, 2>(d, sink, "64KB chunks, 2 slots, 2 in flight (no compute slot)");
    run<4, 4, 1>(d, sink, "32KB chunks, 4 slots, 1 in flight");
    run<4, 4, 2>(d, sink, "32KB chunks, 4 slots, 2 in flight");
    run<4, 4, 3>(d, sink, "32KB chunks, 4 slots, 3 in flight");
    run<2, 8, 3>(d, sink, "16KB chunks, 8 slots, 3 in flight");
    run<2, 8, 5>(d, sink, "16KB chunks, 8 slots, 5 in flight");
    run<2, 8, 7>(d, sink, "16KB chunks, 8 slots, 7 in flight");
    printf("status: %s\n", hipGetErrorString(hipGetLastError()));
}
