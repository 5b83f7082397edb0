// kernels_coarse_wg2.h — MEASUREMENT FLAVOUR ONLY (make ABLATE=1; knob `epi` bit 10 of scripts/ab.py): the emitting coarse
// kernel with TWO workgroups per CU (VERDICT r4 'Next' 8: "the barrier term").
//
// coarse_kernel (kernels_coarse.h) runs ONE workgroup of 8 waves per CU: the two waves of a SIMD belong to the same workgroup,
// meet at the same stage barrier and filter at the same tile boundary, so whatever one of them waits for, the other usually
// waits for too (barrier ~9.5 % of the main launch's cycles, epilogue ~5.8 %: HISTORY.md §9.2). Here a workgroup is 4 waves
// (1(M) x 4(N), the SAME 128 x 64 wave tile = the same registers, fragment reads and k-step as coarse_kernel) over a
// 128 x 256 output tile, its ring is three 24-KiB stages (79 KiB of LDS with the side data: two workgroups fit the CU's 160 KiB),
// and the SIMD's second wave comes from ANOTHER workgroup with its own barrier and its own tile boundaries.
// What it costs, by construction: 6 DMA instructions per wave and stage instead of 4 (the query chunk is fetched once per 128
// corpus rows instead of once per 256: 1.5 x the L2 -> LDS traffic per MFMA) and a DMA lead of two stages instead of three.
//
// A workgroup keeps coarse_kernel's work list - tiles [jlo, jhi) of 256 corpus rows in the launch's visiting order - and walks
// every tile as two HALF-TILES of 128 rows (the first / second 8 KiB of each 16-KiB block of the B32 layout), so the host's
// stage plan, the candidate lists and the select / final kernels are untouched; the launch simply has twice the workgroups
// (nsplit = 2 n_cu / nqt, which the select kernels take up to 256: two or more query tiles only). K must be a multiple of 3
// chunks (every half-tile then starts in ring slot 0 and the stage loop, unrolled by the ring size, has constant LDS addresses):
// D = 384, 768, 1536 for the 2-byte types.
#pragma once
#include "../kernels_coarse.h"

namespace cgv {

constexpr size_t COARSE_WG2_LDS_BYTES = 3 * (size_t)(128 + 256) * 64 + (size_t)256 * 4 + 8 * 128 * 4 + 8 * 8 * 4;

template <int DT, bool NTA>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void coarse_wg2_kernel(const CoarseArgs a) {
    constexpr int BM = 128, BN = 256, NT = 256;
    constexpr int WTM = 128, WTN = 64, MB = 4, NB = 2;
    constexpr int A_BYTES = BM * 64, STAGE = (BM + BN) * 64;
    constexpr int NSTAGE = 3, NINV = 8;
    typedef typename Mfma<DT>::frag frag;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* cntq = (uint32_t*)(smem + NSTAGE * STAGE);
    float* invn_s = (float*)(smem + NSTAGE * STAGE + BN * 4);   // [NINV][128], by half-tile sequence number
    float* stat_s = invn_s + NINV * BM;                          // [NINV][8]: 4 block-min + 4 block-max norms

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave;   // 1 x 4 waves

    uint32_t qt, split;
    const uint32_t g = block_to_work(a, qt, split);

    for (int i = tid; i < BN; i += NT) cntq[i] = 0;
    if ((true) && a.append_splits != 0u) {   // uniform: the first launch behind an emitting sample (kernels_coarse.h)
        __syncthreads();
        append_sample_candidates<BN, NT>(a, g, qt, split, cntq, tid);
    }

    float tauv[NB], tq[NB], invq[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const uint32_t q = qt * BN + wn * WTN + nb * 32 + (lane & 31);
        const bool valid = q < a.nq;
        const float tau = valid ? a.tau[q] : INFINITY;
        const float iq = (a.metric == METRIC_DOT) ? 1.0f : (valid ? a.invn_q[q] : 0.0f);
        tauv[nb] = tau;
        invq[nb] = iq;
        tq[nb] = (tau == -INFINITY) ? -INFINITY : (iq == 0.0f ? INFINITY : tau / iq);
    }
    float ta[NB], thr[MB][NB];
    bool tneg[NB], tone[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        tone[nb] = (a.metric == METRIC_DOT) || !(fabsf(tq[nb]) < INFINITY);
        tneg[nb] = tq[nb] < 0.0f;
        ta[nb] = tone[nb] ? tq[nb] : (tneg[nb] ? tq[nb] * (1.0f + 3.8147e-6f) : tq[nb] * (1.0f - 3.8147e-6f));
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) thr[mb][nb] = INFINITY;
    }

    const uint32_t jlo = __builtin_amdgcn_readfirstlane((uint32_t)(((uint64_t)split * a.cnt) / a.nsplit));
    const uint32_t jhi = __builtin_amdgcn_readfirstlane((uint32_t)(((uint64_t)(split + 1) * a.cnt) / a.nsplit));
    const uint32_t KC = a.kc;
    const uint32_t nhl = (jhi - jlo) * 2u;   // half-tiles of this workgroup
    if (nhl == 0) {                           // uniform (the launcher never makes one: cnt >= nsplit)
        __syncthreads();
        for (int i = tid; i < BN; i += NT) a.cand_cnt[(uint64_t)g * BN + i] = 0;
        return;
    }
    const uint32_t t_first = __builtin_amdgcn_readfirstlane(stage_tile(a.T1, a.R, a.P, a.j0 + jlo) - a.T1);
    auto next_tile = [&](uint32_t t) {
        const uint32_t u = t + a.P;
        return u >= a.R ? u - a.R : u;
    };

    // ---- issue side: runs two stages ahead of the MFMAs ---------------------------------------------------------------------
    // A half-tile's chunk kc: 8 KiB at block (tile, kc) + half * 8 KiB; this wave copies KiB 2w, 2w + 1 of it and KiB
    // 4w .. 4w + 3 of the query chunk (16 KiB), one KiB per instruction.
    uint32_t lt = t_first, lhalf = 0;
    __amdgpu_buffer_rsrc_t rsA =
        __builtin_amdgcn_make_buffer_rsrc((void*)(a.rows + (uint64_t)(a.T1 + lt) * KC * BLOCK_BYTES), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB =
        __builtin_amdgcn_make_buffer_rsrc((void*)(a.qrows + (uint64_t)qt * KC * BLOCK_BYTES), 0, 0x7fffffff, 0x00020000);
    const uint32_t voff = (uint32_t)lane * 16u;
    uint32_t si_slot = 0;   // ring position of the stage being issued, bytes
    uint32_t si_so = 0;     // its chunk's offset within the tile's blocks
    uint32_t d_so = 0;
    char* d_dst = smem;
#define CGW_BDMA(RS, DST, IMM) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(RS, (__attribute__((address_space(3))) void*)(DST), 16, voff, d_so, IMM, 0)
#define CGW_BDMA_A(RS, DST, IMM) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(RS, (__attribute__((address_space(3))) void*)(DST), 16, voff, d_so, IMM, NTA ? 2 : 0)
    auto issue_switch_half = [&]() {
        if (lhalf == 0) {
            lhalf = 1;
        } else {
            lhalf = 0;
            lt = next_tile(lt);
        }
        rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(a.rows + (uint64_t)(a.T1 + lt) * KC * BLOCK_BYTES + lhalf * (uint32_t)A_BYTES),
                                                0, 0x7fffffff, 0x00020000);
        si_so = 0;
    };
    auto issue_q = [&](int q) {
        if (q == 0) {
            d_dst = smem + si_slot + wave * 2048;
            d_so = si_so + (uint32_t)wave * 2048u;
            CGW_BDMA_A(rsA, d_dst, 0);
        } else if (q == 1) {
            CGW_BDMA_A(rsA, d_dst, 1024);
        } else if (q == 2) {
            d_dst = smem + si_slot + A_BYTES + wave * 4096;
            d_so = si_so + (uint32_t)wave * 4096u;
            CGW_BDMA(rsB, d_dst, 0);
        } else if (q == 3) {
            CGW_BDMA(rsB, d_dst, 1024);
        } else if (q == 4) {
            CGW_BDMA(rsB, d_dst, 2048);
        } else {
            CGW_BDMA(rsB, d_dst, 3072);
            si_so += BLOCK_BYTES;
        }
    };
    // side data of the half-tile with sequence number seq: 128 inverse norms (32 lanes of wave 0), 4 + 4 block norm bounds
    // (2 lanes of wave 1); read KC >= 3 stages later, behind >= 6 younger DMA instructions of the same wave
    auto issue_side = [&](uint32_t tt, uint32_t half, uint32_t seq) {
        if (wave == 0 && lane < 32)
            glds16((const char*)a.invn_c + ((uint64_t)(a.T1 + tt) * 256 + half * 128u) * 4 + lane * 16,
                   (char*)(invn_s + (seq & (NINV - 1)) * BM));
        if (wave == 1 && lane < 2) {
            const float* sp = ((lane & 1) ? a.blk_max : a.blk_min) + (uint64_t)(a.T1 + tt) * 8 + half * 4u;
            glds16((const char*)sp, (char*)(stat_s + (seq & (NINV - 1)) * 8));
        }
    };

    uint32_t xo[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) xo[kk] = (uint32_t)((((2 * kk + (lane >> 5)) ^ ((lane >> 2) & 3))) << 4);
    const uint32_t aoff = (uint32_t)(lane & 31) * 64;
    const uint32_t boff = (uint32_t)A_BYTES + (uint32_t)(wn * WTN + (lane & 31)) * 64;

    f32x16_t acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.0f;
    f32x16_t zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.0f;
    frag fa0[MB], fb0[NB], fa1[MB], fb1[NB];

#define CGW_LDA(FA, I, BASE, KK) FA[I] = *(const frag*)((BASE) + aoff + (I) * 2048 + xo[KK]);
#define CGW_LDB(FB, I, BASE, KK) FB[I] = *(const frag*)((BASE) + boff + (I) * 2048 + xo[KK]);
#define CGW_LOAD_FRAGS_A(FA, BASE, KK) \
    { CGW_LDA(FA, 0, BASE, KK) CGW_LDA(FA, 1, BASE, KK) CGW_LDA(FA, 2, BASE, KK) CGW_LDA(FA, 3, BASE, KK) }
#define CGW_LOAD_FRAGS_B(FB, BASE, KK) \
    { CGW_LDB(FB, 0, BASE, KK) CGW_LDB(FB, 1, BASE, KK) }
#define CGW_LOAD_FRAGS(FA, FB, BASE, KK) \
    { CGW_LOAD_FRAGS_A(FA, BASE, KK) CGW_LOAD_FRAGS_B(FB, BASE, KK) }
#define CGW_MMA(MBI, NBI, FA, FB) acc[MBI][NBI] = Mfma<DT>::mma(FA[MBI], FB[NBI], acc[MBI][NBI]);
#define CGW_MMAZ(MBI, NBI, FA, FB) acc[MBI][NBI] = Mfma<DT>::mma(FA[MBI], FB[NBI], zero16);
    // program-order pins, as in coarse_kernel: one gap = what is issued between MFMA (MBI, NBI) and the next one
#define CGW_GAP(MBI, NBI, NEXT_OPERAND, ACTION)                        \
    asm volatile("" : "+v"(acc[MBI][NBI])::"memory");                  \
    ACTION;                                                            \
    asm volatile("" : "+v"(NEXT_OPERAND)::"memory");
#define CGW_NOP_ACTION
    // one k-step (serpentine block order), every gap's action spelled out
#define CGW_KSTEP_X(MMA, FA, FB, NA, G0, G1, G2, G3, G4, G5, G6, G7)                     \
    {                                                                                    \
        MMA(0, 0, FA, FB) CGW_GAP(0, 0, FB[1], G0)                                       \
        MMA(0, 1, FA, FB) CGW_GAP(0, 1, FA[1], G1)                                       \
        MMA(1, 1, FA, FB) CGW_GAP(1, 1, FB[0], G2)                                       \
        MMA(1, 0, FA, FB) CGW_GAP(1, 0, FA[2], G3)                                       \
        MMA(2, 0, FA, FB) CGW_GAP(2, 0, FB[1], G4)                                       \
        MMA(2, 1, FA, FB) CGW_GAP(2, 1, FA[3], G5)                                       \
        MMA(3, 1, FA, FB) CGW_GAP(3, 1, FB[0], G6)                                       \
        MMA(3, 0, FA, FB) CGW_GAP(3, 0, NA[0], G7)                                       \
    }
    // stage s landed for this wave (6 younger DMA instructions = stage s + 1), then for all of them; the barrier also frees
    // the slot of stage s - 1 (every wave has its fragments of (s - 1, 1) in registers) for the DMA of stage s + 2
#define CGW_STAGE_SYNC asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); __builtin_amdgcn_s_barrier()
    // iteration s = B phase [k-step (s-1, 1), stage barrier in its first gap, first fragments of stage s behind it, first half of
    // the DMA of stage s + 2] + A phase [k-step (s, 0), second fragments of stage s, second half of the DMA]
#define CGW_B_PHASE(SB_)                                                                                         \
    CGW_KSTEP_X(CGW_MMA, fa1, fb1, fa0, CGW_STAGE_SYNC, CGW_LOAD_FRAGS_A(fa0, SB_, 0), CGW_LOAD_FRAGS_B(fb0, SB_, 0), \
                issue_q(0), CGW_NOP_ACTION, issue_q(1), CGW_NOP_ACTION, issue_q(2))
#define CGW_A_PHASE_(MMA, SB_)                                                                                   \
    CGW_KSTEP_X(MMA, fa0, fb0, fa1, CGW_LOAD_FRAGS(fa1, fb1, SB_, 1), CGW_NOP_ACTION, CGW_NOP_ACTION, issue_q(3),  \
                CGW_NOP_ACTION, issue_q(4), CGW_NOP_ACTION, issue_q(5))
#define CGW_A_PHASE(SB_) CGW_A_PHASE_(CGW_MMA, SB_)
#define CGW_A_PHASE_Z(SB_) CGW_A_PHASE_(CGW_MMAZ, SB_)
    // thresholds of the half-tile that ends here (block_threshold in raw-accumulator units), formed in the gaps of its last k-step
    f32x4_t mn4, mx4;
#define CGW_THR_LOAD(SEQ)                                                  \
    {                                                                      \
        const float* st_ = stat_s + ((SEQ) & (NINV - 1)) * 8;              \
        mn4 = *(const f32x4_t*)st_;                                        \
        mx4 = *(const f32x4_t*)(st_ + 4);                                  \
    }
#define CGW_THR(MBI, NBI, MN, MX)                                          \
    {                                                                      \
        float mul_ = tone[NBI] ? 1.0f : (tneg[NBI] ? (MX) : (MN));         \
        asm volatile("" : "+v"(mul_));                                     \
        thr[MBI][NBI] = ta[NBI] * mul_;                                    \
        asm volatile("" : "+v"(thr[MBI][NBI]));                            \
    }
#define CGW_B_PHASE_LAST(SB_, SEQ)                                                                                  \
    CGW_KSTEP_X(CGW_MMA, fa1, fb1, fa0, CGW_STAGE_SYNC, CGW_LOAD_FRAGS_A(fa0, SB_, 0); CGW_THR_LOAD(SEQ),           \
                CGW_LOAD_FRAGS_B(fb0, SB_, 0); CGW_THR(0, 0, mn4.x, mx4.x) CGW_THR(0, 1, mn4.x, mx4.x),             \
                issue_q(0); CGW_THR(1, 0, mn4.y, mx4.y), CGW_THR(1, 1, mn4.y, mx4.y) CGW_THR(2, 0, mn4.z, mx4.z),   \
                issue_q(1); CGW_THR(2, 1, mn4.z, mx4.z), CGW_THR(3, 0, mn4.w, mx4.w), issue_q(2); CGW_THR(3, 1, mn4.w, mx4.w))
#define CGW_THR_ALL(SEQ)                                                                                            \
    {                                                                                                               \
        CGW_THR_LOAD(SEQ)                                                                                           \
        CGW_THR(0, 0, mn4.x, mx4.x) CGW_THR(0, 1, mn4.x, mx4.x) CGW_THR(1, 0, mn4.y, mx4.y) CGW_THR(1, 1, mn4.y, mx4.y) \
        CGW_THR(2, 0, mn4.z, mx4.z) CGW_THR(2, 1, mn4.z, mx4.z) CGW_THR(3, 0, mn4.w, mx4.w) CGW_THR(3, 1, mn4.w, mx4.w) \
    }
    // filter of block (MBI, NBI) of the half-tile that just ended, in the gap in front of the zero-C MFMA that overwrites it
    uint32_t ftile = 0;
    const float* finv = invn_s;
    auto filt_block = [&](auto mb_c, auto nb_c) __attribute__((always_inline)) {
        constexpr int MBI = decltype(mb_c)::value, NBI = decltype(nb_c)::value;
        asm volatile("" : "+v"(acc[MBI][NBI])::"memory");
        if (__builtin_expect(__ballot(block_max(acc[MBI][NBI]) > thr[MBI][NBI]) != 0ull, 0))  // cold, out of line
            block_hits<BM, BN>(a, acc[MBI][NBI], thr[MBI][NBI], tauv[NBI], invq[NBI], (uint32_t)(MBI * 32),
                               (uint32_t)(wn * WTN + NBI * 32 + (lane & 31)), ftile, lane, g, qt, cntq, finv, (const float4*)nullptr);
    };
#define CGW_FILT(MBI, NBI) filt_block(IntC<MBI>{}, IntC<NBI>{});
#define CGW_A_PHASE_ZF(SB_)                                                                                         \
    {                                                                                                               \
        CGW_FILT(0, 0)                                                                                              \
        asm volatile("" : "+v"(fa0[0])::"memory");                                                                  \
        CGW_KSTEP_X(CGW_MMAZ, fa0, fb0, fa1, CGW_LOAD_FRAGS(fa1, fb1, SB_, 1); CGW_FILT(0, 1), CGW_FILT(1, 1),      \
                    CGW_FILT(1, 0), issue_q(3); CGW_FILT(2, 0), CGW_FILT(2, 1), issue_q(4); CGW_FILT(3, 1),         \
                    CGW_FILT(3, 0), issue_q(5))                                                                     \
    }

    // ---- prologue: stages 0 and 1 of the first half-tile, its side data ----------------------------------------------------
    issue_side(t_first, 0, 0);
#pragma unroll 1
    for (int i = 0; i < NSTAGE - 1; ++i) {
        si_slot = (uint32_t)(i * STAGE);
        issue_q(0);
        issue_q(1);
        issue_q(2);
        issue_q(3);
        issue_q(4);
        issue_q(5);
    }
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");    // stage 0 (and the side data before it) landed; stage 1 may be in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my cntq zero-stores done
    __builtin_amdgcn_s_barrier();
    CGW_LOAD_FRAGS(fa0, fb0, smem, 0);
    si_slot = (uint32_t)(2 * STAGE);
    issue_q(0);   // stage 2 -> slot 2 (never used so far); its second half goes out in the A phase below
    issue_q(1);
    issue_q(2);
    CGW_A_PHASE_Z(smem);

    // ring-unrolled stage loop: every half-tile starts in slot 0 (KC % 3 == 0)
#define CGW_ITER_AT(SLOT)                                                         \
    {                                                                             \
        const char* sbc = smem + (SLOT) * STAGE;                                  \
        si_slot = (uint32_t)((((SLOT) + 2) % NSTAGE) * STAGE);                    \
        CGW_B_PHASE(sbc);                                                         \
        CGW_A_PHASE(sbc);                                                         \
    }
    // (the iteration of stage KC - 2 issues stage KC = the next half-tile's first: the issue side switches in front of it)
#define CGW_HALF_REST                                                             \
    {                                                                             \
        const uint32_t ng = KC / 3u;                                              \
        if (ng == 1) issue_switch_half();                                         \
        CGW_ITER_AT(1) CGW_ITER_AT(2)                                             \
        _Pragma("unroll 1") for (uint32_t gi = 1; gi < ng; ++gi) {                \
            CGW_ITER_AT(0)                                                        \
            if (gi + 1 == ng) issue_switch_half();                                \
            CGW_ITER_AT(1) CGW_ITER_AT(2)                                         \
        }                                                                         \
    }
    CGW_HALF_REST

    uint32_t ct = t_first, chalf = 0;
#pragma unroll 1
    for (uint32_t hl = 1; hl < nhl; ++hl) {
        // boundary: last k-step of the half-tile that ends (thresholds in its gaps), its filters in front of the zero-C MFMAs
        // of the half-tile that starts
        si_slot = (uint32_t)(2 * STAGE);
        CGW_B_PHASE_LAST(smem, hl - 1);
        ftile = (a.T1 + ct) * 2u + chalf;
        finv = invn_s + ((hl - 1) & (NINV - 1)) * BM;
        if (chalf == 0) {
            chalf = 1;
        } else {
            chalf = 0;
            ct = next_tile(ct);
        }
        issue_side(ct, chalf, hl);
        CGW_A_PHASE_ZF(smem);
        CGW_HALF_REST
    }
    // tail: second k-step of the last stage, then the last half-tile's epilogue
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = Mfma<DT>::mma(fa1[mb], fb1[nb], acc[mb][nb]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the DMA that ran on past the last half-tile
    __builtin_amdgcn_s_barrier();
    CGW_THR_ALL(nhl - 1);
    tile_filter_emit<BM, BN, WTM, WTN, MB, NB>(a, acc, thr, (a.T1 + ct) * 2u + chalf, 0, wn, lane, g, qt, tauv, invq, cntq,
                                               invn_s + ((nhl - 1) & (NINV - 1)) * BM, (const float4*)nullptr);
#undef CGW_HALF_REST
#undef CGW_ITER_AT
#undef CGW_A_PHASE_ZF
#undef CGW_FILT
#undef CGW_THR_ALL
#undef CGW_B_PHASE_LAST
#undef CGW_THR
#undef CGW_THR_LOAD
#undef CGW_A_PHASE_Z
#undef CGW_A_PHASE
#undef CGW_A_PHASE_
#undef CGW_B_PHASE
#undef CGW_STAGE_SYNC
#undef CGW_KSTEP_X
#undef CGW_NOP_ACTION
#undef CGW_GAP
#undef CGW_MMAZ
#undef CGW_MMA
#undef CGW_LOAD_FRAGS
#undef CGW_LOAD_FRAGS_B
#undef CGW_LOAD_FRAGS_A
#undef CGW_LDB
#undef CGW_LDA
#undef CGW_BDMA_A
#undef CGW_BDMA

    __syncthreads();
    for (int i = tid; i < BN; i += NT) {
        const uint32_t c = cntq[i];
        a.cand_cnt[(uint64_t)g * BN + i] = c < CAND_CAPS ? c : CAND_CAPS;
    }
}

}  // namespace cgv
