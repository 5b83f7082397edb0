// Microbenchmark (one MI355X): issue rate of v_accvgpr_read_b32 beside / without MFMAs - what bounds the epilogue of the
// one-wave-per-SIMD fp8 coarse kernel (256 accumulators in AGPRs, 16 reads per 32 x 32 block in front of one 64-cycle MFMA).
//   mode 0: N x 16 v_accvgpr_read_b32 (independent)                      -> cycles per read
//   mode 1: N x 16 v_max3_f32 on VGPRs                                  -> cycles per VALU op (reference)
//   mode 2: N x [1 MFMA 32x32x64 f8f6f4 (zero operands)]                 -> cycles per MFMA
//   mode 3: N x [16 reads + 1 MFMA]   mode 4: N x [8 reads + 1 MFMA]   mode 5: N x [16 reads + 13 v_max3 + 1 MFMA]
// one workgroup of 256 threads (one wave per SIMD) per CU; cycles from s_memtime around the loop, per wave.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/accread.hip -o scripts/ubench/accread && scripts/ubench/accread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k(int iters, unsigned long long* out, float* sink) {
    f32x16 acc, acc1;   // the MFMAs run on acc; the reads take acc1 (independent registers: what the kernel's epilogue does -
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f, acc1[i] = (float)i;   // it reads a block whose last MFMA is 16 instructions old)
    i32x8 za = {0, 0, 0, 0, 0, 0, 0, 0};
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = (float)(threadIdx.x + i);
    float w[16];
    for (int i = 0; i < 16; ++i) w[i] = (float)(threadIdx.x * 3 + i);
    float m = 0.0f;
    asm volatile("" : "+a"(acc), "+a"(acc1));
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 3 || MODE == 4 || MODE == 5) {
            const int nr = MODE == 4 ? 8 : 16;
#pragma unroll
            for (int i = 0; i < nr; ++i) {
                float t;
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(acc1[i]));
                v[i] = t;
            }
        }
        if (MODE == 1 || MODE == 5) {
            const int nm = MODE == 5 ? 13 : 16;
#pragma unroll
            for (int i = 0; i < nm; ++i) {   // independent ops (sources from the previous iteration's registers of another group)
                float t;
                asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(t) : "v"(w[i]), "v"(w[(i + 1) & 15]), "v"(w[(i + 2) & 15]));
                v[i] = t;
            }
        }
        if (MODE >= 2) {
            acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(za, za, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            asm volatile("" : "+a"(acc));
        }
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    for (int i = 0; i < 16; ++i) m += v[i];
    m += acc[0] + acc1[3] + w[5];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[MODE] = t1 - t0;
    if (m == 12345.678f) sink[0] = m;
}

int main() {
    unsigned long long* out;
    float* sink;
    hipMalloc(&out, 64);
    hipMalloc(&sink, 4);
    hipMemset(out, 0, 64);
    const int iters = 20000, grid = 256;
    hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, iters, out, sink);
    hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, iters, out, sink);
    hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, iters, out, sink);
    hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, iters, out, sink);
    hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, iters, out, sink);
    hipLaunchKernelGGL(k<5>, dim3(grid), dim3(256), 0, 0, iters, out, sink);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
    // s_memtime / readcyclecounter ticks at a fixed 100 MHz on this part: report ticks per iteration and ratios
    const char* names[] = {"16 accvgpr_read", "16 v_max3", "1 mfma 32x32x64 f8", "16 reads + mfma", "8 reads + mfma", "16 reads + 13 max3 + mfma"};
    for (int i = 0; i < 6; ++i)
        printf("mode %d (%s): %llu ticks / %d iters = %.4f ticks per iteration (x %.2f of the bare MFMA)\n", i, names[i], h[i], iters,
               (double)h[i] / iters, h[2] ? (double)h[i] / (double)h[2] : 0.0);
    return 0;
}
