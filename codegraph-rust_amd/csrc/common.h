// common.h — shared device helpers for libcgvec_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cgv {

constexpr int DT_F32 = 0, DT_BF16 = 1, DT_FP16 = 2, DT_FP8 = 3;
constexpr int METRIC_COSINE = 0, METRIC_DOT = 1, METRIC_COSINE_SEQ = 2, METRIC_COSINE_SCALAR = 3;

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

// ---- local row -> reported id ----------------------------------------------------
// One device index reports id = base + local row. Under a block-cyclic sharded handle
// (cgv_sharded_*, sharded.hip) global rows are dealt to the shards in chunks of `chunk` rows:
// global row r lives on shard (r / chunk) % nshards at local row (r / chunk / nshards) * chunk + r % chunk.
struct IdMap {
    uint64_t base;
    uint32_t chunk, nshards, shard, pad_;
};
__host__ __device__ inline uint64_t map_id(const IdMap& m, uint32_t local) {
    if (m.nshards <= 1u) return m.base + local;
    return m.base + ((uint64_t)(local / m.chunk) * m.nshards + m.shard) * m.chunk + (local % m.chunk);
}

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

// OCP FP8 E4M3FN (gfx950's fp8): bias 7, no inf, NaN = S.1111.111, max 448. RNE, finite inputs
// beyond the range saturate to +-448.
__device__ inline uint8_t f32_to_e4m3_rne(float f) {
    const uint32_t u = __float_as_uint(f);
    const uint8_t sign = (uint8_t)((u >> 24) & 0x80u);
    const uint32_t a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return (uint8_t)(sign | 0x7fu);
    const float af = __uint_as_float(a);
    if (af >= 464.0f) return (uint8_t)(sign | 0x7eu);
    if (af < 0.0009765625f) return sign;  // below half of the smallest subnormal 2^-9
    int ex = (int)(a >> 23) - 127;
    if (ex < -6) ex = -6;                                                 // subnormal quantum 2^-9
    const float q = __uint_as_float((uint32_t)(ex - 3 + 127) << 23);      // quantum 2^(ex-3)
    float v = rintf(af / q) * q;                                          // RNE; /q and *q are exact
    if (v > 448.0f) v = 448.0f;
    if (v == 0.0f) return sign;
    const uint32_t vu = __float_as_uint(v);
    const int e2 = (int)(vu >> 23) - 127;
    if (e2 < -6) return (uint8_t)(sign | (uint8_t)(int)(v * 512.0f));     // k * 2^-9, k = 1..7
    return (uint8_t)(sign | (uint8_t)((e2 + 7) << 3) | (uint8_t)((vu >> 20) & 7u));
}
__host__ __device__ inline float e4m3_to_f32(uint8_t b) {
    const uint32_t sign = (uint32_t)(b & 0x80u) << 24;
    const uint32_t ex = (b >> 3) & 0xfu, man = b & 7u;
    uint32_t u;
    if (ex == 0) {
        float f = (float)man * 0.001953125f;  // man * 2^-9, exact
        __builtin_memcpy(&u, &f, 4);
        u |= sign;
    } else if (ex == 15 && man == 7) {
        u = sign | 0x7fc00000u;
    } else {
        u = sign | ((ex + 120u) << 23) | (man << 20);
    }
    float r;
    __builtin_memcpy(&r, &u, 4);
    return r;
}
// fp8 rows are accepted when their largest magnitude lies in [2^-48, 2^48] (or the row is zero): the coarse pass
// accumulates de-scaled dot products in f32 (kernels_coarse_fp8.h), which two rows at 2^64 would overflow - outside
// that range an add / update / query is rejected with CGV_ERR_INVALID_ARG instead of being answered unsafely.
constexpr int FP8_EXP_MIN = -40, FP8_EXP_MAX = 56;  // exponents of amax = 2^48 .. 2^-48 (amax * 2^e in (224, 448])
// Per-row power-of-two scale exponent of the fp8 storage: largest e with amax * 2^e <= 448
// (integer logic; pure bit arithmetic, no libm).
__host__ __device__ inline int fp8_row_exponent(float amax) {
    uint32_t u;
    __builtin_memcpy(&u, &amax, 4);
    const int bexp = (int)((u >> 23) & 0xffu);
    if (amax == 0.0f || bexp == 0 || bexp == 0xff) return 0;
    const int x = bexp - 127;
    int e = ((u & 0x7fffffu) <= 0x600000u) ? 8 - x : 7 - x;
    if (e > 100) e = 100;
    if (e < -100) e = -100;
    return e;
}

// The boot sample: logical boot row r -> corpus row. The sample is made of aligned 32-row
// groups; group i of the sample is group (i * P) mod R of the corpus (R groups in all, P ~ 0.618 R coprime to
// R: a golden-ratio stride, every prefix spread evenly over the corpus). P = 1: the identity.
struct BootMap {
    uint32_t P, R;
};
__host__ __device__ inline uint32_t boot_row(const BootMap& m, uint32_t r) {
    return (uint32_t)(((uint64_t)(r >> 5) * m.P) % m.R) * 32u + (r & 31u);
}

// ---- storage layouts ---------------------------------------------------------------
// f32 corpora: plain row-major [rows][ld] (exact path only; the reference's own layout).
// bf16/fp16/fp8 corpora: BLOCKED layout "B32", the unit the coarse kernel streams:
//   rows are grouped in tiles of 256; K in chunks of 64 BYTES (32 two-byte elements, 64 fp8
//   codes); block (tile t, chunk kc) is 16 KiB contiguous at byte ((t*KC + kc) << 14),
//   KC = ld*esize/64. Inside a block, row r (0..255) owns 64 B at r*64 and its 16-byte slot p
//   holds piece c = p ^ ((r>>2)&3) of the chunk. That is byte-for-byte the LDS image of one
//   pipeline stage of the MFMA kernel (bank-conflict-free ds_read_b128: the 16 lanes of a
//   read group touch 16 distinct 16-byte bank quads), so the global->LDS DMA is a linear
//   1-KiB copy per wave instruction: contiguous blocks move at ~60 GB/s/CU, row pieces at a
//   1536-B pitch at ~33 (scripts/ubench/dma_ring.hip, dma_bw.hip).
constexpr uint32_t TILE_ROWS = 256, CHUNK_BYTES = 64, BLOCK_BYTES = TILE_ROWS * CHUNK_BYTES;

// elements per 64-byte row chunk
__host__ __device__ inline constexpr uint32_t kchunk_of(int dt) { return dt == DT_FP8 ? 64u : 32u; }

// byte offset of 16-byte piece pc of a row (pieces numbered along K), relative to the row's base
__host__ __device__ inline uint64_t blocked_piece_off(uint32_t pc, uint32_t key) {
    return (uint64_t)(pc >> 2) * BLOCK_BYTES + (((pc & 3u) ^ key) << 4);
}
__host__ __device__ inline uint64_t blocked_row_base(uint64_t R, uint32_t ld, uint32_t kchunk) {
    return ((R >> 8) * (uint64_t)(ld / kchunk)) * BLOCK_BYTES + (R & 255u) * CHUNK_BYTES;
}
__host__ __device__ inline uint32_t blocked_row_key(uint64_t R) { return (uint32_t)((R & 255u) >> 2) & 3u; }

// Query placement of a COARSE_TOP2 launch (kernels_coarse.h: Top2): query j of the one query tile sits in column
// (j / 16) * 64 + j % 16 - 16 queries in the first N-block of each of the four wave columns. (row >> 2) & 3, the slot swizzle key
// of the blocked layout, is the same for j and its column: a row's 64-byte chunks move verbatim.
constexpr uint32_t TOP2_QPW = 16;
__host__ __device__ inline uint32_t top2_col_of(uint32_t j) { return (j / TOP2_QPW) * 64u + (j % TOP2_QPW); }

template <int DT>
struct Elem;
template <>
struct Elem<DT_F32> {
    static constexpr int bytes = 4;
    static __device__ inline float cvt_load(const char* p) { return *(const float*)p; }
    static __device__ inline void cvt_store(char* p, float x) { *(float*)p = x; }
    static __device__ inline float round_trip(float x) { return x; }
    static __host__ __device__ inline uint64_t elem_off(uint32_t i, uint32_t) { return (uint64_t)i * 4; }
};
template <>
struct Elem<DT_BF16> {
    static constexpr int bytes = 2;
    static __device__ inline float cvt_load(const char* p) { return bf16_to_f32(*(const uint16_t*)p); }
    static __device__ inline void cvt_store(char* p, float x) { *(uint16_t*)p = f32_to_bf16_rne(x); }
    static __device__ inline float round_trip(float x) { return bf16_to_f32(f32_to_bf16_rne(x)); }
    static __host__ __device__ inline uint64_t elem_off(uint32_t i, uint32_t key) {
        return blocked_piece_off(i >> 3, key) + ((i & 7u) << 1);
    }
};
template <>
struct Elem<DT_FP16> {
    static constexpr int bytes = 2;
    static __device__ inline float cvt_load(const char* p) { return f16_to_f32(*(const uint16_t*)p); }
    static __device__ inline void cvt_store(char* p, float x) { *(uint16_t*)p = f32_to_f16_rne(x); }
    static __device__ inline float round_trip(float x) { return f16_to_f32(f32_to_f16_rne(x)); }
    static __host__ __device__ inline uint64_t elem_off(uint32_t i, uint32_t key) {
        return blocked_piece_off(i >> 3, key) + ((i & 7u) << 1);
    }
};
template <>
struct Elem<DT_FP8> {  // values are in the row's SCALED domain (x * 2^e), see fp8_row_exponent
    static constexpr int bytes = 1;
    static __device__ inline float cvt_load(const char* p) { return e4m3_to_f32(*(const uint8_t*)p); }
    static __device__ inline void cvt_store(char* p, float x) { *(uint8_t*)p = f32_to_e4m3_rne(x); }
    static __device__ inline float round_trip(float x) { return e4m3_to_f32(f32_to_e4m3_rne(x)); }
    static __host__ __device__ inline uint64_t elem_off(uint32_t i, uint32_t key) {
        return blocked_piece_off(i >> 4, key) + (i & 15u);
    }
};

// One stored row (corpus or query) seen as a sequence of f32 values.
template <int DT>
struct Row {
    const char* p;
    uint32_t key;
    __device__ inline float at(uint32_t i) const { return Elem<DT>::cvt_load(p + Elem<DT>::elem_off(i, key)); }
};
template <int DT>
__device__ inline Row<DT> make_row(const char* base, uint64_t R, uint32_t ld) {
    Row<DT> r;
    if (DT == DT_F32) {
        r.p = base + R * (uint64_t)ld * 4;
        r.key = 0;
    } else {
        r.p = base + blocked_row_base(R, ld, kchunk_of(DT));
        r.key = blocked_row_key(R);
    }
    return r;
}
// address of 16-byte piece pc (consecutive along K) of a stored row
template <int DT>
__device__ inline const char* piece_ptr(const Row<DT>& r, uint32_t pc) {
    if (DT == DT_F32) return r.p + (size_t)pc * 16;
    return r.p + blocked_piece_off(pc, r.key);
}
template <int DT>
__device__ inline char* elem_ptr(char* base, uint64_t R, uint32_t ld, uint32_t i) {
    if (DT == DT_F32) return base + (R * (uint64_t)ld + i) * 4;
    return base + blocked_row_base(R, ld, kchunk_of(DT)) + Elem<DT>::elem_off(i, blocked_row_key(R));
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

// SIMDVectorOps::cosine_similarity_scalar (simd_ops.rs:257-278): ONE sequential accumulator per sum, separate multiply and
// add (no FMA), sqrt of the PRODUCT of the squared norms, 0.0 when that is zero. It is what adaptive_cosine_similarity
// (:281-295) runs for len < 32 on an AVX2 host - and for EVERY length on a host without AVX2 + FMA and on every
// non-x86_64 host (`#[cfg(not(target_arch = "x86_64"))]`, :291-294; SURVEY.md §4: the authors build on aarch64, where this
// is the only path that ever runs). CGV_METRIC_COSINE_SCALAR / CGV_OP_COSINE_SCALAR select it for every length.
template <class RQ, class RC>
__device__ inline float exact_cosine_scalar(const RQ& q, const RC& c, uint32_t D) {
    float dp = 0.0f, na = 0.0f, nb = 0.0f;
    for (uint32_t i = 0; i < D; ++i) {
        const float x = q.at(i), y = c.at(i);
        dp = dp + x * y;
        na = na + x * x;
        nb = nb + y * y;
    }
    const float np = sqrtf(na * nb);
    return (np == 0.0f) ? 0.0f : dp / np;
}

template <class RQ, class RC>
__device__ inline float exact_cosine_group8(const RQ& q, const RC& c, uint32_t D, int l) {
    float result = 0.0f;
    if (D >= 32) {  // adaptive_cosine_similarity dispatch, simd_ops.rs:281-295
        float dp = 0.0f, na = 0.0f, nb = 0.0f;
        const uint32_t chunks = D / 8;
        // (unrolled: the element loads of 8 iterations are issued together; out of LDS one load at a time costs its
        //  full latency per iteration - the FMA order per lane is unchanged)
#pragma unroll 8
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
        result = exact_cosine_scalar(q, c, D);
    }
    return result;
}

template <class RQ, class RC>
__device__ inline float exact_dot_group8(const RQ& q, const RC& c, uint32_t D, int l) {
    // dot_product_avx2, simd_ops.rs:149-183
    float dp = 0.0f;
    const uint32_t chunks = D / 8;
#pragma unroll 8
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

template <class RQ, class RC>
__device__ inline float exact_score_group8(int metric, const RQ& q, const RC& c, uint32_t D, int l) {
    if (metric == METRIC_COSINE_SEQ) return l == 0 ? exact_cosine_seq(q, c, D, false) : 0.0f;
    if (metric == METRIC_COSINE_SCALAR) return l == 0 ? exact_cosine_scalar(q, c, D) : 0.0f;
    return metric == METRIC_DOT ? exact_dot_group8(q, c, D, l) : exact_cosine_group8(q, c, D, l);
}

constexpr int OP_COSINE = 0, OP_DOT = 1, OP_L2 = 2, OP_COSINE_SEQ = 3, OP_COSINE_DISTANCE_SEQ = 4,
              OP_NEG_COSINE_DISTANCE_SEQ = 5,  // 5 (internal): -(distance), so that "larger is better" orders by distance asc
              OP_COSINE_SCALAR = 6;            // cosine_similarity_scalar for every length (a host without AVX2 / not x86_64)

template <class RQ, class RC>
__device__ inline float exact_op_group8(int op, const RQ& q, const RC& c, uint32_t D, int l) {
    switch (op) {
        case OP_DOT: return exact_dot_group8(q, c, D, l);
        case OP_L2: return exact_l2_group8(q, c, D, l);
        case OP_COSINE_SEQ: return l == 0 ? exact_cosine_seq(q, c, D, false) : 0.0f;
        case OP_COSINE_DISTANCE_SEQ: return l == 0 ? exact_cosine_seq(q, c, D, true) : 0.0f;
        case OP_NEG_COSINE_DISTANCE_SEQ: return l == 0 ? -exact_cosine_seq(q, c, D, true) : 0.0f;
        case OP_COSINE_SCALAR: return l == 0 ? exact_cosine_scalar(q, c, D) : 0.0f;
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
// The k-th largest of M <= 1024 floats, by ONE WAVE (result uniform; -inf when fewer than k values exist; -inf entries =
// "no value"). Values only, as 32-bit ordered keys: <= 16 per lane sorted in registers by a bitonic network, then k rounds
// of wave maximum + pop. Used for the first threshold of a search: d[] holds maxima of DISJOINT groups of
// corpus rows, so the k'-th largest of them is a score that at least k' distinct rows reach - by tau_kernel (its own
// launch) and inside the fused sample + emit launch of the coarse kernel (kernels_coarse.h, COARSE_EMIT_BOOT). d is read
// with plain vector loads: a caller that reads values other workgroups wrote in the same launch acquires first.
__device__ inline void cmpx_desc32(uint32_t& x, uint32_t& y) {
    const uint32_t hi = x > y ? x : y, lo = x > y ? y : x;
    x = hi;
    y = lo;
}
// COHERENT: the values were written by OTHER workgroups of the same launch with agent-scope (sc1, write-through) stores and
// are read here with agent-scope loads, which never hit a stale line of this CU's L1 or this XCD's L2 - no acquire fence
// (MI355X_MICROARCH.md: a buffer_inv / buffer_wbl2 pair per hand-off is what makes a fenced hand-off 2-3x dearer).
// hi_rank / hi_val (optional): the value at rank hi_rank <= k of the same descending order (the threshold ladder's spread)
template <bool COHERENT = false>
__device__ inline float kth_largest_wave(const float* d, uint32_t M, uint32_t k, int lane, uint32_t hi_rank = 0,
                                         float* hi_val = nullptr) {
    uint32_t r[16];
    // lane l owns elements 4l..4l+3 of every 256-element slab (16-byte loads, coalesced); M <= 1024
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t e = 256u * j + 4u * (uint32_t)lane;
        float4 v = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        if (COHERENT) {
            if (e < M) v.x = __hip_atomic_load(d + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (e + 1 < M) v.y = __hip_atomic_load(d + e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (e + 2 < M) v.z = __hip_atomic_load(d + e + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (e + 3 < M) v.w = __hip_atomic_load(d + e + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (e + 3 < M) {
            v = *(const float4*)(d + e);
        } else {
            if (e < M) v.x = d[e];
            if (e + 1 < M) v.y = d[e + 1];
            if (e + 2 < M) v.z = d[e + 2];
        }
        // NaN never occurs (non-finite inputs are rejected); -inf = "no row": key 0x007fffff, never 0
        r[4 * j] = f2ord(v.x + 0.0f);
        r[4 * j + 1] = f2ord(v.y + 0.0f);
        r[4 * j + 2] = f2ord(v.z + 0.0f);
        r[4 * j + 3] = f2ord(v.w + 0.0f);
    }
    // bitonic sorting network on 16 registers, descending
#pragma unroll
    for (int k2 = 2; k2 <= 16; k2 <<= 1)
#pragma unroll
        for (int j = k2 >> 1; j > 0; j >>= 1)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    if ((i & k2) == 0) cmpx_desc32(r[i], r[ixj]);
                    else cmpx_desc32(r[ixj], r[i]);
                }
            }
    const uint32_t none = f2ord(-INFINITY);   // "no value" entries never count
    uint32_t cnt = 0, last = 0;
    if (hi_val) *hi_val = -INFINITY;
    while (cnt < k) {
        const uint32_t w = wave_max_u32(r[0]);
        if (w <= none) break;  // fewer than k values
        const bool own = (r[0] == w);
        const uint32_t before = cnt;
        cnt += (uint32_t)__popcll(__ballot(own));  // equal values in several lanes count once each
        if (hi_val && before < hi_rank && cnt >= hi_rank) *hi_val = ord2f(w);
        last = w;
        if (own) {
#pragma unroll
            for (int i = 0; i < 15; ++i) r[i] = r[i + 1];
            r[15] = 0u;
        }
    }
    return (cnt >= k) ? ord2f(last) : -INFINITY;
}

// The same selection with T instead of 16 registers per lane: every lane keeps only the T largest of its <= 16 values
// (insertion as they are loaded), and the k-th largest of the UNION of those lists is returned. Dropping elements of a
// multiset can only lower its k-th largest, so the result is ALWAYS a valid lower bound of the true one - which is all a first
// threshold has to be - and it IS the true one unless a single lane holds more than T of the k largest
// (T = 8, k = 16, 64 lanes: ~1e-9 per query). For the fused sample + emit launch of the coarse kernel, where 128
// accumulators per lane are live around the call: with the 16-register network the kernel spilled, and the allocator's choice
// was reloaded from scratch in every hit path of the whole walk (kernels_coarse.h: boot_resolve).
template <bool COHERENT, int T>
__device__ inline float kth_largest_wave_top(const float* d, uint32_t M, uint32_t k, int lane) {
    uint32_t r[T];
#pragma unroll
    for (int i = 0; i < T; ++i) r[i] = 0u;   // empty: below every key (-inf is 0x007fffff)
#pragma unroll 1
    for (uint32_t j = 0; j < 4u; ++j) {
#pragma unroll
        for (uint32_t c = 0; c < 4u; ++c) {
            const uint32_t e = 256u * j + 4u * (uint32_t)lane + c;
            float v = -INFINITY;
            if (e < M) v = COHERENT ? __hip_atomic_load(d + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : d[e];
            uint32_t key = f2ord(v + 0.0f);
#pragma unroll
            for (int i = 0; i < T; ++i) {   // sorted insertion: r[] stays descending
                const uint32_t hi = r[i] > key ? r[i] : key, lo = r[i] > key ? key : r[i];
                r[i] = hi;
                key = lo;
            }
        }
    }
    const uint32_t none = f2ord(-INFINITY);
    uint32_t cnt = 0, last = 0;
    while (cnt < k) {
        const uint32_t w = wave_max_u32(r[0]);
        if (w <= none) break;  // fewer than k values (kept)
        const bool own = (r[0] == w);
        cnt += (uint32_t)__popcll(__ballot(own));
        last = w;
        if (own) {
#pragma unroll
            for (int i = 0; i < T - 1; ++i) r[i] = r[i + 1];
            r[T - 1] = 0u;
        }
    }
    return (cnt >= k) ? ord2f(last) : -INFINITY;
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
