// common.h — shared device helpers for libcgvec_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cgv {

constexpr int DT_F32 = 0, DT_BF16 = 1, DT_FP16 = 2, DT_FP8 = 3;
constexpr int METRIC_COSINE = 0, METRIC_DOT = 1;

// ---- sortable keys: larger key == better (score desc, row asc) -----------------
__host__ __device__ inline uint32_t f2ord(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float ord2f(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}
__host__ __device__ inline uint64_t make_key(float s, uint32_t row) {
    s = s + 0.0f;  // -0.0 -> +0.0 so that equal scores tie-break on the row id
    return ((uint64_t)f2ord(s) << 32) | (uint64_t)(uint32_t)(~row);
}
__host__ __device__ inline float key_score(uint64_t k) { return ord2f((uint32_t)(k >> 32)); }
__host__ __device__ inline uint32_t key_row(uint64_t k) { return ~(uint32_t)k; }

// ---- storage dtype conversion (round-to-nearest-even) ---------------------------
__host__ __device__ inline uint16_t f32_to_bf16_rne(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__host__ __device__ inline float bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}
__device__ inline uint16_t f32_to_f16_rne(float f) {
    _Float16 h = (_Float16)f;  // v_cvt_f16_f32: RNE, overflow -> inf
    uint16_t b;
    __builtin_memcpy(&b, &h, 2);
    return b;
}
__device__ inline float f16_to_f32(uint16_t b) {
    _Float16 h;
    __builtin_memcpy(&h, &b, 2);
    return (float)h;
}

// ---- storage layouts ---------------------------------------------------------------
// f32 corpora: plain row-major [rows][ld] (exact path only; the reference's own layout).
// bf16/fp16 corpora: BLOCKED layout "B64", the unit the coarse kernel streams:
//   rows are grouped in tiles of 256; K in chunks of 64 elements; block (tile t, chunk kc)
//   is 32 KiB contiguous at byte ((t*KC + kc) << 15), KC = ld/64. Inside a block, row r
//   (0..255) owns 128 B at r*128 and its 16-byte slot p holds elements 8c..8c+7 of the
//   chunk with c = p ^ ((r>>1)&7). That is byte-for-byte the LDS image the MFMA kernel
//   wants (bank-conflict-free ds_read_b128), so the global->LDS DMA is a linear 1-KiB copy
//   per wave instruction: measured 64 KB / 1.05 us per CU vs 1.98 us for 128-B pieces at a
//   1536-B row pitch (scripts/ubench/dma_ring.hip, dma_bw.hip).
constexpr uint32_t TILE_ROWS = 256, KCHUNK = 64, BLOCK_BYTES = TILE_ROWS * KCHUNK * 2;

__host__ __device__ inline uint64_t blocked_row_base(uint64_t R, uint32_t ld) {
    return ((R >> 8) * (uint64_t)(ld / KCHUNK)) * BLOCK_BYTES + (R & 255u) * 128u;
}
__host__ __device__ inline uint32_t blocked_row_key(uint64_t R) { return (uint32_t)((R & 255u) >> 1) & 7u; }
// byte offset of element i of a row, relative to blocked_row_base
__host__ __device__ inline uint64_t blocked_elem_off(uint32_t i, uint32_t key) {
    return (uint64_t)(i >> 6) * BLOCK_BYTES + ((((i >> 3) & 7u) ^ key) << 4) + ((i & 7u) << 1);
}

template <int DT>
struct Elem;
template <>
struct Elem<DT_F32> {
    static constexpr int bytes = 4;
    static __device__ inline float cvt_load(const char* p) { return *(const float*)p; }
    static __device__ inline void cvt_store(char* p, float x) { *(float*)p = x; }
    static __device__ inline float round_trip(float x) { return x; }
};
template <>
struct Elem<DT_BF16> {
    static constexpr int bytes = 2;
    static __device__ inline float cvt_load(const char* p) { return bf16_to_f32(*(const uint16_t*)p); }
    static __device__ inline void cvt_store(char* p, float x) { *(uint16_t*)p = f32_to_bf16_rne(x); }
    static __device__ inline float round_trip(float x) { return bf16_to_f32(f32_to_bf16_rne(x)); }
};
template <>
struct Elem<DT_FP16> {
    static constexpr int bytes = 2;
    static __device__ inline float cvt_load(const char* p) { return f16_to_f32(*(const uint16_t*)p); }
    static __device__ inline void cvt_store(char* p, float x) { *(uint16_t*)p = f32_to_f16_rne(x); }
    static __device__ inline float round_trip(float x) { return f16_to_f32(f32_to_f16_rne(x)); }
};

// One stored row (corpus or query) seen as a sequence of f32 values.
template <int DT>
struct Row {
    const char* p;
    uint32_t key;
    __device__ inline float at(uint32_t i) const {
        if (DT == DT_F32) return Elem<DT>::cvt_load(p + (uint64_t)i * 4);
        return Elem<DT>::cvt_load(p + blocked_elem_off(i, key));
    }
};
template <int DT>
__device__ inline Row<DT> make_row(const char* base, uint64_t R, uint32_t ld) {
    Row<DT> r;
    if (DT == DT_F32) {
        r.p = base + R * (uint64_t)ld * 4;
        r.key = 0;
    } else {
        r.p = base + blocked_row_base(R, ld);
        r.key = blocked_row_key(R);
    }
    return r;
}
template <int DT>
__device__ inline char* elem_ptr(char* base, uint64_t R, uint32_t ld, uint32_t i) {
    if (DT == DT_F32) return base + (R * (uint64_t)ld + i) * 4;
    return base + blocked_row_base(R, ld) + blocked_elem_off(i, blocked_row_key(R));
}

// ---- the reference's exact f32 arithmetic ---------------------------------------
// One similarity is evaluated by a group of 8 consecutive lanes; lane l of the group
// plays AVX2 lane l of the reference's __m256 accumulators
// (crates/codegraph-vector/src/simd_ops.rs:15-78): it sums elements i == l (mod 8)
// sequentially with fused FMA; the group then reproduces horizontal_sum_avx2
// (simd_ops.rs:227-242): ((l0+l4)+(l1+l5))+((l2+l6)+(l3+l7)); lane 0 adds the
// scalar tail (D mod 8, separate mul/add rounding) and finishes
// dp / sqrt(na_sq*nb_sq). The library is compiled with -ffp-contract=off so only the
// explicit fmaf() calls fuse; sqrt and divide are IEEE correctly rounded
// (-fhip-fp32-correctly-rounded-divide-sqrt).
// The result is valid on lane 0 of each group.
__device__ inline float group8_hsum(float v) {
    float x = v + __shfl_down(v, 4, 8);   // lanes 0..3: l0+l4, l1+l5, l2+l6, l3+l7
    float y = x + __shfl_down(x, 1, 8);   // lane 0: (l0+l4)+(l1+l5); lane 2: (l2+l6)+(l3+l7)
    float z = y + __shfl_down(y, 2, 8);   // lane 0: full tree
    return z;
}

template <class RQ, class RC>
__device__ inline float exact_cosine_group8(const RQ& q, const RC& c, uint32_t D, int l) {
    float result = 0.0f;
    if (D >= 32) {  // adaptive_cosine_similarity dispatch, simd_ops.rs:281-295
        float dp = 0.0f, na = 0.0f, nb = 0.0f;
        const uint32_t chunks = D / 8;
        for (uint32_t j = 0; j < chunks; ++j) {
            float x = q.at(8 * j + l);
            float y = c.at(8 * j + l);
            dp = fmaf(x, y, dp);
            na = fmaf(x, x, na);
            nb = fmaf(y, y, nb);
        }
        dp = group8_hsum(dp);
        na = group8_hsum(na);
        nb = group8_hsum(nb);
        if (l == 0) {
            float dr = 0.0f, ar = 0.0f, br = 0.0f;
            for (uint32_t i = chunks * 8; i < D; ++i) {
                float x = q.at(i), y = c.at(i);
                dr = dr + x * y;
                ar = ar + x * x;
                br = br + y * y;
            }
            float fd = dp + dr, fa = na + ar, fb = nb + br;
            float np = sqrtf(fa * fb);
            result = (np == 0.0f) ? 0.0f : fd / np;
        }
    } else if (l == 0) {  // cosine_similarity_scalar, simd_ops.rs:257-278
        float dp = 0.0f, na = 0.0f, nb = 0.0f;
        for (uint32_t i = 0; i < D; ++i) {
            float x = q.at(i), y = c.at(i);
            dp = dp + x * y;
            na = na + x * x;
            nb = nb + y * y;
        }
        float np = sqrtf(na * nb);
        result = (np == 0.0f) ? 0.0f : dp / np;
    }
    return result;
}

template <class RQ, class RC>
__device__ inline float exact_dot_group8(const RQ& q, const RC& c, uint32_t D, int l) {
    // dot_product_avx2, simd_ops.rs:149-183
    float dp = 0.0f;
    const uint32_t chunks = D / 8;
    for (uint32_t j = 0; j < chunks; ++j) dp = fmaf(q.at(8 * j + l), c.at(8 * j + l), dp);
    dp = group8_hsum(dp);
    float result = 0.0f;
    if (l == 0) {
        float r = 0.0f;
        for (uint32_t i = chunks * 8; i < D; ++i) r = r + q.at(i) * c.at(i);
        result = (D == 0) ? 0.0f : dp + r;
    }
    return result;
}

template <class RQ, class RC>
__device__ inline float exact_score_group8(int metric, const RQ& q, const RC& c, uint32_t D, int l) {
    return metric == METRIC_DOT ? exact_dot_group8(q, c, D, l) : exact_cosine_group8(q, c, D, l);
}

// l2_distance_avx2, simd_ops.rs:105-143 (8 lanes: diff, fused square-accumulate; h-sum; scalar
// tail with separate rounding; sqrt of the total)
template <class RQ, class RC>
__device__ inline float exact_l2_group8(const RQ& q, const RC& c, uint32_t D, int l) {
    float acc = 0.0f;
    const uint32_t chunks = D / 8;
    for (uint32_t j = 0; j < chunks; ++j) {
        const float d = q.at(8 * j + l) - c.at(8 * j + l);
        acc = fmaf(d, d, acc);
    }
    acc = group8_hsum(acc);
    float result = 0.0f;
    if (l == 0) {
        float r = 0.0f;
        for (uint32_t i = chunks * 8; i < D; ++i) {
            const float d = q.at(i) - c.at(i);
            r = r + d * d;
        }
        result = (D == 0) ? 0.0f : sqrtf(acc + r);
    }
    return result;
}

// The reference's scalar formula (sequential f32 sums, no FMA): search.rs:519-533,
// optimization.rs:404-418, gpu.rs:324-338. One lane walks the whole row.
// distance = false: dot/(|a||b|), 0 on a zero norm; true: 1 - that, +inf on a zero norm.
template <class RQ, class RC>
__device__ inline float exact_cosine_seq(const RQ& q, const RC& c, uint32_t D, bool distance) {
    float dot = 0.0f, na = 0.0f, nb = 0.0f;
    for (uint32_t i = 0; i < D; ++i) {
        const float x = q.at(i), y = c.at(i);
        dot = dot + x * y;
        na = na + x * x;
        nb = nb + y * y;
    }
    const float norm_a = sqrtf(na), norm_b = sqrtf(nb);
    if (norm_a == 0.0f || norm_b == 0.0f) return distance ? INFINITY : 0.0f;
    const float cs = dot / (norm_a * norm_b);
    return distance ? 1.0f - cs : cs;
}

constexpr int OP_COSINE = 0, OP_DOT = 1, OP_L2 = 2, OP_COSINE_SEQ = 3, OP_COSINE_DISTANCE_SEQ = 4,
              OP_NEG_COSINE_DISTANCE_SEQ = 5;  // 5: -(distance), so that "larger is better" orders by distance asc

template <class RQ, class RC>
__device__ inline float exact_op_group8(int op, const RQ& q, const RC& c, uint32_t D, int l) {
    switch (op) {
        case OP_DOT: return exact_dot_group8(q, c, D, l);
        case OP_L2: return exact_l2_group8(q, c, D, l);
        case OP_COSINE_SEQ: return l == 0 ? exact_cosine_seq(q, c, D, false) : 0.0f;
        case OP_COSINE_DISTANCE_SEQ: return l == 0 ? exact_cosine_seq(q, c, D, true) : 0.0f;
        case OP_NEG_COSINE_DISTANCE_SEQ: return l == 0 ? -exact_cosine_seq(q, c, D, true) : 0.0f;
        default: return exact_cosine_group8(q, c, D, l);
    }
}

// A row staged in LDS as linear [ld] storage-dtype elements.
template <int DT>
struct LdsRow {
    const char* p;
    __device__ inline float at(uint32_t i) const { return Elem<DT>::cvt_load(p + (size_t)i * Elem<DT>::bytes); }
};

// Wave-wide (64 lanes) unsigned max, result uniform. DPP row operations + 4 readlanes:
// ~20 issue slots, versus ~1.4k cycles for a 6-step ds_bpermute (__shfl_xor) butterfly.
__device__ inline uint32_t wave_max_u32(uint32_t x) {
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
    x = t > x ? t : x;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    x = t > x ? t : x;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x141, 0xF, 0xF, false);  // row_half_mirror
    x = t > x ? t : x;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x140, 0xF, 0xF, false);  // row_mirror
    x = t > x ? t : x;
    const uint32_t a = __builtin_amdgcn_readlane(x, 0), b = __builtin_amdgcn_readlane(x, 16);
    const uint32_t c = __builtin_amdgcn_readlane(x, 32), d = __builtin_amdgcn_readlane(x, 48);
    const uint32_t ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}
// 64-bit max as two 32-bit phases (high word, then low word among the lanes that hold it).
__device__ inline uint64_t wave_max_u64(uint64_t k) {
    const uint32_t hi = (uint32_t)(k >> 32), lo = (uint32_t)k;
    const uint32_t mh = wave_max_u32(hi);
    const uint32_t ml = wave_max_u32(hi == mh ? lo : 0u);
    return ((uint64_t)mh << 32) | ml;
}

// In-LDS bitonic sort, descending, P a power of two, NT threads.
template <int NT>
__device__ inline void bitonic_sort_desc(uint64_t* keys, uint32_t P, int tid) {
    for (uint32_t k2 = 2; k2 <= P; k2 <<= 1) {
        for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
            for (uint32_t i = tid; i < P; i += NT) {
                uint32_t ixj = i ^ j;
                if (ixj > i) {
                    uint64_t a = keys[i], b = keys[ixj];
                    bool desc_blk = ((i & k2) == 0);
                    if (desc_blk ? (a < b) : (a > b)) {
                        keys[i] = b;
                        keys[ixj] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
}

__host__ __device__ inline uint32_t next_pow2(uint32_t x) {
    uint32_t p = 1;
    while (p < x) p <<= 1;
    return p;
}

}  // namespace cgv
