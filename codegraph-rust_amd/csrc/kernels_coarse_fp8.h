// kernels_coarse_fp8.h — the coarse kernel for fp8 (e4m3) corpora on the gfx950 block-scaled MFMA
// v_mfma_scale_f32_32x32x64_f8f6f4 (both scales 2^0): K = 64 per instruction at twice the FLOP rate
// of the bf16 / non-scaled fp8 MFMA (MI355X_MICROARCH.md: ~5 PF dense). Same tile, ring, barrier and
// epilogue as coarse_kernel (kernels_coarse.h); what changes is the fragment pipeline:
//   * one pipeline stage (a 64-byte K chunk) is exactly ONE K=64 MFMA per 32x32 block: a lane's
//     operand is its row's 32 bytes of the chunk half lane>>5, i.e. two 16-byte LDS pieces;
//   * a full fragment set would be 48 VGPRs and cannot be double-buffered next to 128 accumulators,
//     so a stage runs as two phases of 4 MFMAs: phase 1 = rows 0-63 of the wave tile (A01) x B,
//     phase 2 = rows 64-127 (A23) x B. A01/A23 are single-buffered (each is reloaded while the other
//     phase's MFMAs run), B is double-buffered by stage parity (the loop body is two stages).
// Any K permutation inside the instruction is harmless: A and B use the same one.
#pragma once
#include "kernels_coarse.h"

namespace cgv {

typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x8_t __attribute__((ext_vector_type(8)));

struct Fp8Frag {
    i32x4_t p0, p1;  // the two 16-byte pieces of this lane's 32 bytes
};

__device__ inline f32x16_t mma_fp8_k64(const Fp8Frag& a, const Fp8Frag& b, f32x16_t c) {
    const i32x8_t av = __builtin_shufflevector(a.p0, a.p1, 0, 1, 2, 3, 4, 5, 6, 7);
    const i32x8_t bv = __builtin_shufflevector(b.p0, b.p1, 0, 1, 2, 3, 4, 5, 6, 7);
    // cbsz = blgp = 0: both operands OCP e4m3; E8M0 scale 0x7F = 2^0 for every 32-element block
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
}
// The same with the operands' per-row power-of-two scales applied BY THE INSTRUCTION: a lane's 32 bytes are one
// 32-element scale block of its row (A) / query (B), so byte SA of `sa` (SB of `sb`) is that row's E8M0 scale
// 2^(x - 127) = 2^-e. The accumulators then hold dot products of the DE-SCALED values, every row of a block in the
// same domain - what the norm-bound fast filter of the epilogue needs (with the scales left at 2^0, rows whose
// exponents differ by one sit a factor 2 apart and a quarter of the lanes false-alarm in every block:
// 1949 -> 2443 TFLOP/s on a corpus with one exponent, scripts/gpu_fp8_probe.py).
template <int SA, int SB>
__device__ inline f32x16_t mma_fp8_k64_scaled(const Fp8Frag& a, const Fp8Frag& b, f32x16_t c, int sa, int sb) {
    const i32x8_t av = __builtin_shufflevector(a.p0, a.p1, 0, 1, 2, 3, 4, 5, 6, 7);
    const i32x8_t bv = __builtin_shufflevector(b.p0, b.p1, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 0, 0, SA, sa, SB, sb);
}

template <int MODE>  // COARSE_EMIT / COARSE_DUMP / COARSE_SAMPLE (coarse_launch.h)
__global__ __launch_bounds__(512) void coarse_fp8s_kernel(const CoarseArgs a) {
    constexpr int BM = 256, BN = 256, WN = 4, NT = 512;
    constexpr int WTM = 128, WTN = 64, MB = 4, NB = 2;
    constexpr int A_BYTES = BM * 64, STAGE = (BM + BN) * 64;
    constexpr int NSTAGE = 4, NINV = 8;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* cntq = (uint32_t*)(smem + NSTAGE * STAGE);
    float* invn_s = (float*)(smem + NSTAGE * STAGE + BN * 4);
    float* stat_s = invn_s + NINV * 256;
    int8_t* rexp_s = (int8_t*)(stat_s + NINV * 16);  // [NINV][256] per-row scale exponents, by tile sequence number

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    uint32_t qt, split;
    const uint32_t g = block_to_work(a, qt, split);

    for (int i = tid; i < BN; i += NT) cntq[i] = 0;
    if ((MODE == 0) && a.append_splits != 0u) {   // uniform: the first launch behind an emitting sample (kernels_coarse.h)
        __syncthreads();
        append_sample_candidates<BN, NT>(a, g, qt, split, cntq, tid);
    }

    float tauv[NB], tq[NB], invq[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const uint32_t q = column_query<MODE, BN, WTN>(qt, wn, nb, lane);   // (COARSE_TOP2: the spread placement, top2_col_of)
        const bool valid = q < a.nq;
        const float tau = MODE == 4 ? INFINITY : (valid ? a.tau[q] : INFINITY);   // COARSE_TOP2: no threshold at all
        const float iq = valid ? a.invn_q[q] : 0.0f;  // fp8 is cosine-only
        tauv[nb] = tau;
        invq[nb] = iq;
        tq[nb] = (tau == -INFINITY) ? -INFINITY : (iq == 0.0f ? INFINITY : tau / iq);
    }

    Top2 t2[NB];   // COARSE_TOP2: the lane's cells (kernels_coarse.h)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const bool active = MODE == 4 && top2_query_of(wn, nb, lane) < a.nq;
        t2[nb].s1 = t2[nb].s2 = -INFINITY;
        t2[nb].b = active ? -INFINITY : INFINITY;
        t2[nb].r1 = t2[nb].r2 = 0xFFFFFFFFu;
    }

    const uint32_t jlo = (uint32_t)(((uint64_t)split * a.cnt) / a.nsplit);
    const uint32_t jhi = (uint32_t)(((uint64_t)(split + 1) * a.cnt) / a.nsplit);
    const uint32_t KC = a.kc;
    const uint32_t total = (jhi - jlo) * KC;
    const uint32_t t_first = (total > 0) ? stage_tile(a.T1, a.R, a.P, a.j0 + jlo) - a.T1 : 0u;
    auto next_tile = [&](uint32_t t) {
        const uint32_t u = t + a.P;
        return u >= a.R ? u - a.R : u;
    };

    // ---- DMA issue side: identical to coarse_kernel (endless stream, one instruction at a time) ----
    uint32_t lj = 0, lkc = 0, issued = 0, lt = t_first;
    // buffer_load ... lds: scalar resource + scalar offset + one constant per-lane offset (kernels_coarse.h)
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.rows + (uint64_t)(a.T1 + lt) * KC * BLOCK_BYTES), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.qrows + (uint64_t)qt * KC * BLOCK_BYTES), 0, 0x7fffffff, 0x00020000);
    const uint32_t voff = (uint32_t)lane * 16u;
    auto bdma = [&](__amdgpu_buffer_rsrc_t rs, uint32_t soff, char* l) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)l, 16, voff, soff, 0, 0);
    };
    auto issue_q = [&](int q) {
        char* dst = smem + (issued & (NSTAGE - 1)) * STAGE + wave * 2048;
        const uint64_t koff = (uint64_t)lkc * BLOCK_BYTES;
        const uint32_t so = (uint32_t)koff + (uint32_t)wave * 2048u;
        if (q == 0) {
            if (lkc == 0 && issued < total) {
                if (wave == 0)
                    glds16((const char*)a.invn_c + (uint64_t)(a.T1 + lt) * 1024 + lane * 16,
                           (char*)(invn_s + (lj & (NINV - 1)) * 256));
                if (wave == 1 && lane < 4) {
                    const float* sp = ((lane & 2) ? a.blk_max : a.blk_min) + (uint64_t)(a.T1 + lt) * 8 + (lane & 1) * 4;
                    glds16((const char*)sp, (char*)(stat_s + (lj & (NINV - 1)) * 16));
                }
                if (wave == 2 && lane < 16)  // the tile's 256 scale exponents
                    glds16((const char*)a.rexp_c + (uint64_t)(a.T1 + lt) * 256 + lane * 16,
                           (char*)(rexp_s + (lj & (NINV - 1)) * 256));
            }
            bdma(rsA, so, dst);
        } else if (q == 1) {
            bdma(rsA, so + 1024, dst + 1024);
        } else if (q == 2) {
            bdma(rsB, so, dst + A_BYTES);
        } else {
            bdma(rsB, so + 1024, dst + A_BYTES + 1024);
            ++issued;
            if (issued < total && ++lkc == KC) {
                lkc = 0;
                ++lj;
                lt = next_tile(lt);
                rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(a.rows + (uint64_t)(a.T1 + lt) * KC * BLOCK_BYTES), 0,
                                                        0x7fffffff, 0x00020000);
            }
        }
    };

    if (total == 0) {
        for (int i = tid; i < BN; i += NT) a.cand_cnt[(uint64_t)g * BN + i] = 0;
        return;
    }

    // fragment read offsets: row r = base32 + (lane&31); this lane's 32 bytes are pieces 2h, 2h+1 of
    // the chunk (h = lane>>5), stored at slots piece ^ ((r>>2)&3)
    const uint32_t key = (uint32_t)(lane >> 2) & 3u, h2 = (uint32_t)(lane >> 5) * 2u;
    const uint32_t xo0 = ((h2 ^ key) << 4), xo1 = (((h2 + 1u) ^ key) << 4);
    const uint32_t aoff = (uint32_t)(wm * WTM + (lane & 31)) * 64;
    const uint32_t boff = (uint32_t)A_BYTES + (uint32_t)(wn * WTN + (lane & 31)) * 64;

    f32x16_t acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.0f;

    // E8M0 scale bytes (127 - e = 2^-e): sb = the lane's two query columns (byte nb), constant for the workgroup;
    // sa = its four corpus rows of the current tile (byte mb), reloaded from the LDS ring at every tile boundary
    int sb = 0;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const uint32_t q = column_query<MODE, BN, WTN>(qt, wn, nb, lane);
        const int e = q < a.nq ? (int)a.rexp_q[q] : 0;
        sb |= ((127 - e) & 0xff) << (8 * nb);
    }
    int sa = 0x7f7f7f7f;
    auto load_sa = [&](uint32_t seq) {
        const int8_t* p = rexp_s + (seq & (NINV - 1)) * 256 + wm * WTM + (lane & 31);
        int v = 0;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) v |= ((127 - (int)p[mb * 32]) & 0xff) << (8 * mb);
        sa = v;
    };
    Fp8Frag fa01[2], fa23[2], fbx[2], fby[2];
#define F8_LOAD(F, BASE, OFF)                                  \
    {                                                          \
        F.p0 = *(const i32x4_t*)((BASE) + (OFF) + xo0);        \
        F.p1 = *(const i32x4_t*)((BASE) + (OFF) + xo1);        \
    }
#define F8_LOAD_A01(BASE) { F8_LOAD(fa01[0], BASE, aoff); F8_LOAD(fa01[1], BASE, aoff + 32 * 64); }
#define F8_LOAD_A23(BASE) { F8_LOAD(fa23[0], BASE, aoff + 64 * 64); F8_LOAD(fa23[1], BASE, aoff + 96 * 64); }
#define F8_LOAD_B(FB, BASE) { F8_LOAD(FB[0], BASE, boff); F8_LOAD(FB[1], BASE, boff + 32 * 64); }
#define F8_SB __builtin_amdgcn_sched_barrier(0)
#define F8_EPILOGUE()                                                                                          \
    {                                                                                                          \
        tile_epilogue<BM, BN, WTM, WTN, MB, NB, MODE>(a, acc, ptile, wm, wn, lane, g, qt, tq, tauv, invq, cntq, \
                                                      invn_s + (pj & (NINV - 1)) * 256,                        \
                                                      stat_s + (pj & (NINV - 1)) * 16, a.j0 + jlo + pj, t2);   \
        _Pragma("unroll") for (int mb = 0; mb < MB; ++mb) _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)    \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.0f;                              \
    }
// One stage s (fragments A01(s) in fa01, B(s) in BC already loaded). BN_ receives B(s+1).
//   phase 1: 4 MFMAs A01 x BC; A23(s) is read behind the first one; DMA quarters 2,3 of stage s+3
//   phase 2: 4 MFMAs A23 x BC; behind the first one: wait for my share of stage s+1, barrier, read
//            A01(s+1) and B(s+1); DMA quarters 0,1 of stage s+4. Then the tile epilogue if s ended a tile.
#define F8_STAGE(BC, BN_, SB, SNEXT)                                                             \
    {                                                                                            \
        F8_SB;                                                                                   \
        acc[0][0] = mma_fp8_k64_scaled<0, 0>(fa01[0], BC[0], acc[0][0], sa, sb);                                      \
        F8_SB;                                                                                   \
        F8_LOAD_A23(SB);                                                                         \
        F8_SB;                                                                                   \
        acc[0][1] = mma_fp8_k64_scaled<0, 1>(fa01[0], BC[1], acc[0][1], sa, sb);                                      \
        F8_SB;                                                                                   \
        issue_q(2);                                                                              \
        F8_SB;                                                                                   \
        acc[1][0] = mma_fp8_k64_scaled<1, 0>(fa01[1], BC[0], acc[1][0], sa, sb);                                      \
        F8_SB;                                                                                   \
        issue_q(3);                                                                              \
        F8_SB;                                                                                   \
        acc[1][1] = mma_fp8_k64_scaled<1, 1>(fa01[1], BC[1], acc[1][1], sa, sb);                                      \
        F8_SB;                                                                                   \
        acc[2][0] = mma_fp8_k64_scaled<2, 0>(fa23[0], BC[0], acc[2][0], sa, sb);                                      \
        F8_SB;                                                                                   \
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                         \
        __builtin_amdgcn_s_barrier();                                                            \
        F8_SB;                                                                                   \
        F8_LOAD_A01(SNEXT);                                                                      \
        F8_LOAD_B(BN_, SNEXT);                                                                   \
        F8_SB;                                                                                   \
        acc[2][1] = mma_fp8_k64_scaled<2, 1>(fa23[0], BC[1], acc[2][1], sa, sb);                                      \
        F8_SB;                                                                                   \
        issue_q(0);                                                                              \
        F8_SB;                                                                                   \
        acc[3][0] = mma_fp8_k64_scaled<3, 0>(fa23[1], BC[0], acc[3][0], sa, sb);                                      \
        F8_SB;                                                                                   \
        issue_q(1);                                                                              \
        F8_SB;                                                                                   \
        acc[3][1] = mma_fp8_k64_scaled<3, 1>(fa23[1], BC[1], acc[3][1], sa, sb);                                      \
        F8_SB;                                                                                   \
        if (++ckc == KC) {                                                                       \
            ckc = 0;                                                                             \
            ptile = a.T1 + ct;                                                                   \
            pj = cj;                                                                             \
            ++cj;                                                                                \
            ct = next_tile(ct);                                                                  \
            if (done < total) F8_EPILOGUE();                                                     \
            load_sa(cj); /* the tile that starts with the next stage (its exponents landed with its first stage) */ \
        }                                                                                        \
        ++done;                                                                                  \
    }

    // ---- prologue: stages 0..2 in flight, stage 0 landed, first fragments, stage 3 half issued ----
#pragma unroll 1
    for (int i = 0; i < NSTAGE - 1; ++i) {
        issue_q(0);
        issue_q(1);
        issue_q(2);
        issue_q(3);
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    load_sa(0);
    F8_LOAD_A01(smem);
    F8_LOAD_B(fbx, smem);
    issue_q(0);
    issue_q(1);

    // Stages are processed in pairs (B double-buffered by parity). When `total` is odd the last pair's
    // second stage is a dummy: it multiplies the re-read tail of the DMA stream into accumulators that
    // were already consumed by the last tile's epilogue (tiles end exactly at stage total-1).
    uint32_t cj = 0, ckc = 0, ct = t_first, ptile = 0, pj = 0, done = 0;
#pragma unroll 1
    for (uint32_t s = 0; s < total; s += 2) {
        const char* s0 = smem + (s & (NSTAGE - 1)) * STAGE;
        const char* s1 = smem + ((s + 1) & (NSTAGE - 1)) * STAGE;
        const char* s2 = smem + ((s + 2) & (NSTAGE - 1)) * STAGE;
        F8_STAGE(fbx, fby, s0, s1);
        F8_STAGE(fby, fbx, s1, s2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the DMA tail before the LDS is released
#undef F8_STAGE
#undef F8_EPILOGUE
#undef F8_SB
#undef F8_LOAD_B
#undef F8_LOAD_A23
#undef F8_LOAD_A01
#undef F8_LOAD

    if (MODE == 4) top2_flush<BN, NB>(a, t2, invq, wm, wn, lane, g, qt, cntq);
    __syncthreads();
    for (int i = tid; i < BN; i += NT) {
        const uint32_t c = cntq[i];
        a.cand_cnt[(uint64_t)g * BN + i] = c < CAND_CAPS ? c : CAND_CAPS;
    }
}

}  // namespace cgv
