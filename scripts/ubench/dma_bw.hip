// Microbenchmark: global->LDS DMA (global_load_lds_dwordx4) fill rate per CU.
// 256 WGs x 512 threads (1 per CU), each "chunk" = 64 KB into LDS = 8 DMA instr per wave.
//   mode 0: every WG reads the same 1.5 MB region (L2 resident, like the query operand)
//   mode 1: each WG streams its own region (HBM)
//   mode 2: groups of 4 WGs (same XCD) stream the same region (like the corpus operand)
//   pitch : 0 = contiguous 64 KB per chunk; else row pitch in bytes for 128-B rows (512 rows/chunk)
//   depth : chunks kept in flight (1 or 2; LDS 2 x 64 KB)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__device__ inline void glds16(const char* g, char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int DEPTH, bool PLAIN>
__global__ __launch_bounds__(512) void dma_kernel(const char* src, size_t region_bytes, int mode, int pitch,
                                                  int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t g = blockIdx.x;
    if ((gridDim.x & 7u) == 0) g = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    size_t base = 0;
    if (mode == 1) base = (size_t)g * region_bytes;
    if (mode == 2 || mode == 3) base = (size_t)(g / 4) * region_bytes;
    const char* p = src + base;
    const char* pq = src + ((size_t)5 << 30) + (size_t)(g % 4) * ((size_t)3 << 17);  // 4 query tiles of 384 KB
    const size_t chunk_stride = pitch ? 128 : 65536;  // next chunk: next 128-B column of the rows, or next 64 KB
    float acc = 0.f;
    uint4 r[8];
    for (int it = 0; it < iters + DEPTH - 1; ++it) {
        if (it < iters) {
            size_t off;
            const int buf = it % 2;
            for (int j = 0; j < 8; ++j) {
                const int row = wave * 64 + j * 8 + (lane >> 3);  // 512 rows of 128 B per chunk
                if (pitch) {
                    const int chunks_per_row = pitch / 128;
                    const size_t tile = (size_t)(it / chunks_per_row), kc = it % chunks_per_row;
                    off = (tile * 512 + row) * (size_t)pitch + kc * 128 + (lane & 7) * 16;
                } else {
                    off = (size_t)it * 65536 + (size_t)row * 128 + (lane & 7) * 16;
                }
                off %= region_bytes;
                if (mode == 3) {
                    // A half: 256 rows from the shared-by-4 stream; B half: 256 rows from the 384 KB query tile
                    const int row2 = wave * 32 + (j & 3) * 8 + (lane >> 3);
                    size_t o2;
                    if (pitch) o2 = ((size_t)(it / 12) * 256 + row2) * 1536 + (size_t)(it % 12) * 128 + (lane & 7) * 16;
                    else o2 = (size_t)it * 32768 + (size_t)row2 * 128 + (lane & 7) * 16;
                    if (j < 4) glds16(p + (o2 % region_bytes), smem + buf * 65536 + (wave * 64 + j * 8) * 128);
                    else {
                        size_t oq = pitch ? (size_t)row2 * 1536 + (size_t)(it % 12) * 128 + (lane & 7) * 16
                                          : (size_t)(it % 12) * 32768 + (size_t)row2 * 128 + (lane & 7) * 16;
                        glds16(pq + oq, smem + buf * 65536 + (wave * 64 + j * 8) * 128);
                    }
                    continue;
                }
                if (PLAIN) r[j] = *(const uint4*)(p + off);
                else glds16(p + off, smem + buf * 65536 + (wave * 64 + j * 8) * 128);
            }
        }
        if (DEPTH == 2 && it < iters && it + 1 >= DEPTH) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (PLAIN) { for (int j = 0; j < 8; ++j) acc += __uint_as_float(r[j].x); }
        __builtin_amdgcn_s_barrier();
    }
    if (acc == 123.456f) sink[0] = acc + smem[tid];
}

int main(int argc, char** argv) {
    const size_t total = (size_t)6 << 30;
    char* d; float* sink;
    hipMalloc(&d, total); hipMemset(d, 1, total); hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct Cfg { int mode, pitch, depth, plain; size_t region; const char* name; };
    std::vector<Cfg> cfgs = {
        {0, 0, 1, 0, (size_t)3 << 19, "L2-shared 1.5MB contiguous depth1"},
        {0, 0, 2, 0, (size_t)3 << 19, "L2-shared 1.5MB contiguous depth2"},
        {0, 1536, 1, 0, (size_t)3 << 19, "L2-shared 1.5MB pitch1536 depth1"},
        {1, 0, 1, 0, (size_t)16 << 20, "HBM per-WG 16MB contiguous depth1"},
        {1, 0, 2, 0, (size_t)16 << 20, "HBM per-WG 16MB contiguous depth2"},
        {1, 1536, 1, 0, (size_t)16 << 20, "HBM per-WG 16MB pitch1536 depth1"},
        {1, 1536, 2, 0, (size_t)16 << 20, "HBM per-WG 16MB pitch1536 depth2"},
        {2, 0, 1, 0, (size_t)24 << 20, "HBM shared-by-4 24MB contiguous depth1"},
        {2, 1536, 1, 0, (size_t)24 << 20, "HBM shared-by-4 24MB pitch1536 depth1"},
        {2, 1536, 2, 0, (size_t)24 << 20, "HBM shared-by-4 24MB pitch1536 depth2"},
        {3, 1536, 1, 0, (size_t)24 << 20, "MIX (A shared-by-4 HBM + B query tile) pitch1536 d1"},
        {3, 1536, 2, 0, (size_t)24 << 20, "MIX pitch1536 depth2"},
        {3, 0, 1, 0, (size_t)24 << 20, "MIX contiguous depth1"},
        {3, 0, 2, 0, (size_t)24 << 20, "MIX contiguous depth2"},
        {0, 0, 1, 1, (size_t)3 << 19, "PLAIN loads L2-shared contiguous depth1"},
        {1, 0, 1, 1, (size_t)16 << 20, "PLAIN loads HBM per-WG contiguous depth1"},
    };
    hipFuncSetAttribute((const void*)dma_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void*)dma_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void*)dma_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    const int iters = 240, W = 256;
    for (auto& c : cfgs) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (c.plain) hipLaunchKernelGGL((dma_kernel<1, true>), dim3(W), dim3(512), 131072, 0, d, c.region, c.mode, c.pitch, iters, sink);
            else if (c.depth == 2) hipLaunchKernelGGL((dma_kernel<2, false>), dim3(W), dim3(512), 131072, 0, d, c.region, c.mode, c.pitch, iters, sink);
            else hipLaunchKernelGGL((dma_kernel<1, false>), dim3(W), dim3(512), 131072, 0, d, c.region, c.mode, c.pitch, iters, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        double bytes = (double)W * iters * 65536;
        printf("%-48s %8.3f ms  %7.2f TB/s  %6.1f GB/s/CU  %6.2f us/chunk\n", c.name, best, bytes / best / 1e9,
               bytes / best / 1e6 / W, best * 1e3 / iters);
    }
    hipError_t e = hipGetLastError();
    printf("status: %s\n", hipGetErrorString(e));
    return 0;
}
