// kernels_prep.h — ingest: f32 rows -> storage dtype (RNE) + per-row norms.
// HBM-bound streaming kernels: one wave per row, 16-byte loads and stores.
// Replaces the host-side work of store_embeddings / upload_vectors
// (crates/codegraph-core/src/traits.rs:13; crates/codegraph-vector/src/gpu.rs:221-246)
// and the normalise step of parallel_normalize_vectors (simd_ops.rs:386-419) — the
// corpus is NOT rewritten to unit length (chunk-mean rows are left
// un-normalised by the reference), the inverse norm is kept beside it instead.
#pragma once
#include "common.h"

namespace cgv {

// in: [n][D] f32 (n rows to append). out: the index' row storage (f32: row-major [.][ld];
// bf16/fp16/fp8: blocked layout B32, see common.h), written at absolute rows row0 + r, columns
// zero padded to ld (a whole number of 64-byte chunks).
// fp8 rows are stored scaled by their own power of two (rexp[r], common.h); norm / invn are those of the
// de-scaled stored values (code * 2^-e), the domain the scaled MFMA of the coarse pass accumulates in.
// norm[r] = sqrt(sum of squares of the ROUNDED values) (any order; used only by the
// coarse pass), invn[r] = 1/norm or 0. nonfinite: bit 0 set if any input is NaN/Inf, bit 1 if an fp8 row's
// magnitude is outside the supported range.
// 16 storage bytes from EPP = 16 / esize consecutive f32 values (RNE)
template <int DT>
__device__ inline uint4 pack_piece(const float* x) {
    uint4 o;
    if (DT == DT_F32) {
        o = make_uint4(__float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3]));
    } else if (DT == DT_BF16) {
        o.x = (uint32_t)f32_to_bf16_rne(x[0]) | ((uint32_t)f32_to_bf16_rne(x[1]) << 16);
        o.y = (uint32_t)f32_to_bf16_rne(x[2]) | ((uint32_t)f32_to_bf16_rne(x[3]) << 16);
        o.z = (uint32_t)f32_to_bf16_rne(x[4]) | ((uint32_t)f32_to_bf16_rne(x[5]) << 16);
        o.w = (uint32_t)f32_to_bf16_rne(x[6]) | ((uint32_t)f32_to_bf16_rne(x[7]) << 16);
    } else if (DT == DT_FP16) {
        o.x = (uint32_t)f32_to_f16_rne(x[0]) | ((uint32_t)f32_to_f16_rne(x[1]) << 16);
        o.y = (uint32_t)f32_to_f16_rne(x[2]) | ((uint32_t)f32_to_f16_rne(x[3]) << 16);
        o.z = (uint32_t)f32_to_f16_rne(x[4]) | ((uint32_t)f32_to_f16_rne(x[5]) << 16);
        o.w = (uint32_t)f32_to_f16_rne(x[6]) | ((uint32_t)f32_to_f16_rne(x[7]) << 16);
    } else {
        uint32_t w[4];
#pragma unroll
        for (int d = 0; d < 4; ++d)
            w[d] = (uint32_t)f32_to_e4m3_rne(x[4 * d]) | ((uint32_t)f32_to_e4m3_rne(x[4 * d + 1]) << 8) |
                   ((uint32_t)f32_to_e4m3_rne(x[4 * d + 2]) << 16) | ((uint32_t)f32_to_e4m3_rne(x[4 * d + 3]) << 24);
        o = make_uint4(w[0], w[1], w[2], w[3]);
    }
    return o;
}

// One wave per row; a lane converts whole 16-byte OUTPUT pieces (4 f32 / 8 two-byte / 16 fp8 elements): 16-byte loads
// of the f32 input (the query batch may sit in pinned HOST memory and travel over PCIe right here - wide requests
// matter), one 16-byte store per piece straight into the blocked layout (round 2 stored 2 bytes per lane).
template <int DT>
__global__ __launch_bounds__(256) void prep_rows_kernel(const float* __restrict__ in, uint64_t n,
                                                        uint32_t D, uint32_t ld, uint64_t row0,
                                                        char* __restrict__ out,
                                                        float* __restrict__ norm,
                                                        float* __restrict__ invn,
                                                        int8_t* __restrict__ rexp,
                                                        uint32_t* __restrict__ nonfinite,
                                                        uint32_t* __restrict__ zero0,
                                                        uint32_t* __restrict__ zero1) {
    constexpr int EPP = 16 / Elem<DT>::bytes;
    const int lane = threadIdx.x & 63;
    const uint64_t row = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    if (lane == 0) {  // query batches: per-query state cleared here instead of by two extra memset launches
        if (zero0) zero0[row0 + row] = 0u;
        if (zero1) zero1[row0 + row] = 0u;
    }
    const float* src = in + row * (uint64_t)D;
    const bool vec = (D & 3u) == 0 && (((uintptr_t)in) & 15u) == 0;  // every row starts 16-byte aligned
    const uint32_t pieces = ld / EPP;
    auto load_piece = [&](uint32_t pc, float* x) {
        const uint32_t i0 = pc * EPP;
        if (vec && i0 + EPP <= D) {
#pragma unroll
            for (int v = 0; v < EPP / 4; ++v) {
                const float4 f = *(const float4*)(src + i0 + 4 * v);
                x[4 * v] = f.x;
                x[4 * v + 1] = f.y;
                x[4 * v + 2] = f.z;
                x[4 * v + 3] = f.w;
            }
        } else {
#pragma unroll
            for (int t = 0; t < EPP; ++t) x[t] = (i0 + t < D) ? src[i0 + t] : 0.0f;
        }
    };
    int e = 0;
    if (DT == DT_FP8) {  // per-row power-of-two scale: amax * 2^e <= 448 (common.h)
        float amax = 0.0f;
        for (uint32_t pc = lane; pc < pieces; pc += 64) {
            float x[EPP];
            load_piece(pc, x);
#pragma unroll
            for (int t = 0; t < EPP; ++t) amax = fmaxf(amax, fabsf(x[t]));
        }
        for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
        e = fp8_row_exponent(amax);
        if (lane == 0) rexp[row0 + row] = (int8_t)e;
    }
    float ss = 0.0f, amx = 0.0f;
    int bad = 0;
    if (DT == DT_FP8 && (e < FP8_EXP_MIN || e > FP8_EXP_MAX)) bad = 2;  // magnitude outside the supported range (common.h)
    const uint64_t R = row0 + row;
    char* obase = (DT == DT_F32) ? out + R * (uint64_t)ld * 4 : out + blocked_row_base(R, ld, kchunk_of(DT));
    const uint32_t key = blocked_row_key(R);
    for (uint32_t pc = lane; pc < pieces; pc += 64) {
        float x[EPP];
        load_piece(pc, x);
#pragma unroll
        for (int t = 0; t < EPP; ++t) {
            if (!(fabsf(x[t]) <= 3.402823466e38f)) bad |= 1;  // NaN or Inf
            amx = fmaxf(amx, fabsf(x[t]));
            if (DT == DT_FP8) x[t] = ldexpf(x[t], e);          // exact
            const float xr = Elem<DT>::round_trip(x[t]);
            ss = fmaf(xr, xr, ss);
        }
        *(uint4*)(obase + ((DT == DT_F32) ? (uint64_t)pc * 16 : blocked_piece_off(pc, key))) = pack_piece<DT>(x);
    }
    for (int off = 32; off > 0; off >>= 1) {
        ss += __shfl_xor(ss, off, 64);
        amx = fmaxf(amx, __shfl_xor(amx, off, 64));
    }
    // The MFMA coarse pass and its error bound assume no underflow / overflow in the squared norms and in the products
    // (DESIGN.md §5.3): a row whose largest magnitude lies outside [2^-40, 2^40] (and is not all zero) is answered by the
    // exact scan instead - bit 2 of the flag word (corpus: the index leaves the fast path; queries: zero1[row] = 1
    // sends THAT query through the exact scan, see rescore_body's `overflow`).
    const bool wide = amx != 0.0f && (amx < 9.094947017729282e-13f || amx > 1.099511627776e12f) && DT != DT_FP8;
    if (lane == 0) {
        if (wide && zero1) zero1[row0 + row] = 1u;
        const float nr = sqrtf(ss);
        // fp8: the coarse kernel applies the rows' scales inside the MFMA (kernels_coarse_fp8.h), so its accumulators -
        // and therefore these norms - live in the DE-SCALED domain: norm * 2^-e, (1 / norm) * 2^e, both exact
        norm[row0 + row] = (DT == DT_FP8) ? ldexpf(nr, -e) : nr;
        invn[row0 + row] = nr > 0.0f ? ((DT == DT_FP8) ? ldexpf(1.0f / nr, e) : 1.0f / nr) : 0.0f;
    }
    const uint32_t bits = (__any(bad & 1) ? 1u : 0u) | (__any(bad & 2) ? 2u : 0u) | (wide ? 4u : 0u);
    if (bits && lane == 0) atomicOr(nonfinite, bits);
}

// f32 index with a bf16 SHADOW for the coarse pass (CGV_DTYPE_F32_SHADOW): second ingest pass over the
// same f32 input. Writes the bf16-rounded copy in the blocked layout, the shadow's norm / inverse norm
// (the coarse pass works on the shadow), and the rounding residual of each row:
//   abs = |x - x^|,  rel = abs / min(|x|, |x^|)      (x^ = RNE(x) to bf16)
// which bound how far a coarse score can be from the exact f32 score (kernels_select.h). Per-row values
// go to res[2*row..] when res != NULL (queries); the maxima over all rows are folded into
// res_max[0] (rel) and res_max[1] (abs) as non-negative float bits (corpus).
static __global__ __launch_bounds__(256) void shadow_rows_kernel(const float* __restrict__ in, uint64_t n, uint32_t D,
                                                          uint32_t lds, uint64_t row0, char* __restrict__ out,
                                                          float* __restrict__ norm, float* __restrict__ invn,
                                                          float* __restrict__ res, uint32_t* __restrict__ res_max) {
    const int lane = threadIdx.x & 63;
    const uint64_t row = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float* src = in + row * (uint64_t)D;
    const bool vec = (D & 3u) == 0 && (((uintptr_t)in) & 15u) == 0;
    const uint64_t R = row0 + row;
    char* obase = out + blocked_row_base(R, lds, kchunk_of(DT_BF16));
    const uint32_t key = blocked_row_key(R);
    float sh = 0.0f, sx = 0.0f, sd = 0.0f;
    for (uint32_t pc = lane; pc < lds / 8; pc += 64) {  // one 16-byte piece (8 bf16) per lane and step
        float x[8];
        const uint32_t i0 = pc * 8;
        if (vec && i0 + 8 <= D) {
            const float4 f0 = *(const float4*)(src + i0), f1 = *(const float4*)(src + i0 + 4);
            x[0] = f0.x; x[1] = f0.y; x[2] = f0.z; x[3] = f0.w;
            x[4] = f1.x; x[5] = f1.y; x[6] = f1.z; x[7] = f1.w;
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) x[t] = (i0 + t < D) ? src[i0 + t] : 0.0f;
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float xr = Elem<DT_BF16>::round_trip(x[t]), d = x[t] - xr;
            sh = fmaf(xr, xr, sh);
            sx = fmaf(x[t], x[t], sx);
            sd = fmaf(d, d, sd);
        }
        *(uint4*)(obase + blocked_piece_off(pc, key)) = pack_piece<DT_BF16>(x);
    }
    for (int off = 32; off > 0; off >>= 1) {
        sh += __shfl_xor(sh, off, 64);
        sx += __shfl_xor(sx, off, 64);
        sd += __shfl_xor(sd, off, 64);
    }
    if (lane == 0) {
        const float nh = sqrtf(sh), nx = sqrtf(sx), ra = sqrtf(sd);
        const float mn = fminf(nh, nx);
        const float rr = mn > 0.0f ? ra / mn : 0.0f;  // zero rows never become candidates (score 0 via the exact rule)
        norm[row0 + row] = nh;
        invn[row0 + row] = nh > 0.0f ? 1.0f / nh : 0.0f;
        if (res) {
            res[2 * (row0 + row)] = rr;
            res[2 * (row0 + row) + 1] = ra;
        }
        if (res_max && rr == rr && ra == ra) {  // NaN/Inf rows are rejected by the first pass
            // one pair of words for the whole corpus: look before raising it (an unconditional atomicMax per row
            // serialises at ~12 ns each - 188 GB/s of ingest on a 1M-row f32 + shadow index, r03e)
            if (__float_as_uint(rr) > __hip_atomic_load(res_max, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                atomicMax(res_max, __float_as_uint(rr));
            if (__float_as_uint(ra) > __hip_atomic_load(res_max + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                atomicMax(res_max + 1, __float_as_uint(ra));
        }
    }
}

// Stored row -> f32 (get_embedding); fp8 rows are returned de-scaled (value * 2^-e).
template <int DT>
__global__ void gather_row_kernel(const char* __restrict__ rows, uint64_t R, uint32_t D, uint32_t ld,
                                  const int8_t* __restrict__ rexp, float* __restrict__ out) {
    R += blockIdx.x;  // one block per row: rows [R, R + gridDim.x) -> out[gridDim.x][D]
    out += (uint64_t)blockIdx.x * D;
    const Row<DT> r = make_row<DT>(rows, R, ld);
    const int e = (DT == DT_FP8 && rexp) ? (int)rexp[R] : 0;
    for (uint32_t i = threadIdx.x; i < D; i += blockDim.x) out[i] = ldexpf(r.at(i), -e);
}

// Per aligned 32-row block: min and max row norm over the valid rows (used by the
// conservative fast filter of the coarse kernel's epilogue).
static __global__ __launch_bounds__(256) void block_norm_stats_kernel(const float* __restrict__ norm,
                                                               uint64_t n, uint64_t blk0,
                                                               uint64_t blk1,
                                                               float* __restrict__ blk_min,
                                                               float* __restrict__ blk_max) {
    uint64_t b = blk0 + (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= blk1) return;
    float mn = 3.402823466e38f, mx = 0.0f;
    for (int i = 0; i < 32; ++i) {
        uint64_t r = b * 32 + i;
        if (r < n) {
            float v = norm[r];
            mn = fminf(mn, v);
            mx = fmaxf(mx, v);
        }
    }
    blk_min[b] = mn;
    blk_max[b] = mx;
}

// max over all rows of norm (dot-product error bound); single block, grid-stride.
static __global__ __launch_bounds__(1024) void max_norm_kernel(const float* __restrict__ norm, uint64_t n0,
                                                        uint64_t n1, float* __restrict__ out_max) {
    __shared__ float red[16];
    float m = 0.0f;
    for (uint64_t i = n0 + threadIdx.x; i < n1; i += 1024) m = fmaxf(m, norm[i]);
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float r = *out_max;
        for (int i = 0; i < 16; ++i) r = fmaxf(r, red[i]);
        *out_max = r;
    }
}

// COARSE_TOP2 with more than 16 queries: the spread copy of the (single) query tile - query j's 64-byte row chunks go to row
// top2_col_of(j) of `dst` (same blocked layout, kc_count blocks of 16 KiB). One thread per 16-byte piece.
static __global__ void top2_spread_queries_kernel(const char* __restrict__ src, char* __restrict__ dst, uint32_t nq, uint32_t kc_count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq * kc_count * 4u) return;
    const uint32_t j = i / (kc_count * 4u), rem = i % (kc_count * 4u), kc = rem >> 2, pc = rem & 3u;
    const uint4 v = *(const uint4*)(src + (uint64_t)kc * BLOCK_BYTES + (uint64_t)j * CHUNK_BYTES + pc * 16u);
    *(uint4*)(dst + (uint64_t)kc * BLOCK_BYTES + (uint64_t)top2_col_of(j) * CHUNK_BYTES + pc * 16u) = v;
}

static __global__ void fill_f32_kernel(float* p, float v, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

}  // namespace cgv
