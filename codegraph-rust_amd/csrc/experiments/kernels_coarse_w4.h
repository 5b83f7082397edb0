// kernels_coarse_w4.h — the bf16 / fp16 coarse kernel as ONE WAVE PER SIMD: 4 waves per workgroup, each owning a
// 128 x 128 output tile (4 x 4 blocks of 32x32 MFMA) of the same 256 x 256 workgroup tile, the 256 accumulator
// registers in the accumulator half (AGPRs) of the 512-entry register file.
//
// MEASUREMENT FLAVOUR ONLY (make ABLATE=1; coarse kernel epi bit 7, knob `epi` of scripts/ab.py). History: built in
// round 2 (8 instead of 12 ds_read_b128 per 16 MFMAs, no SIMD partner competing for the matrix pipe), measured equal to the
// 8-wave kernel on the C2 main launch and 8 % slower on hit-heavy launches - with one wave on the SIMD nothing covered its
// per-tile epilogue (256 v_accvgpr_read + the maxima) - and removed in round 3. VERDICT r3 'Next' 3 asked for the A/B that
// was never run: THIS layout with the fp8 one-wave kernel's FOLDED epilogue (kernels_coarse_fp8_w4.h): at a tile boundary the
// next tile's zero-C k-step copies a block's 16 accumulators to VGPRs, issues the MFMA that overwrites them and filters the
// copy while that MFMA occupies the matrix pipe. Resurrected from 559d79f^ with that fold; HISTORY.md §9.2 has the result.
// Everything else is the 8-wave kernel's round-2 design (kernels_coarse.h): B32 blocked operands, 4-slot LDS ring filled by
// buffer_load ... lds three stages ahead and retired by a counted vmcnt, one barrier per stage, tile-structured loop with
// zero-C MFMAs at the tile boundary, XCD-aware workgroup mapping (block_to_work), dynamic issue side. Differences:
//   * a wave copies 4 KiB of the A block and 4 KiB of the B block per stage: 8 DMA instructions, the four of
//     a block sharing one M0 / scalar offset and stepping by the instruction's immediate offset (1 KiB);
//   * with ONE wave on the SIMD nothing covers an issue stall, so everything is placed: the fragment reads of
//     the next k-step go one per MFMA gap behind the first 8 MFMAs, the DMA pieces behind the last 8;
//   * the per-tile side data (inverse norms, block bounds) is issued in the straight-line tile-boundary block
//     for the tile that STARTS there, so the stage loop has no branch besides its back edge.
// A 64-byte stage is two K=16 k-steps (A phase, B phase); needs kc >= 4.
#pragma once
#include "../kernels_coarse.h"

namespace cgv {

// FOLD: the epilogue of a tile runs inside the next tile's zero-C k-step (else: as one block at the tile boundary, the
// round-2 form). ABL: timing-only ablation mask (results are WRONG for ABL != 0): 1 = skip the epilogue, 2 = skip the DMA,
// 4 = skip the barrier, 8 = read the fragments once (real data) and never again, 16 = skip the counted vmcnt wait.
template <int DT, bool FOLD = true, int ABL = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void coarse_w4_kernel(const CoarseArgs a) {
    constexpr int BM = 256, BN = 256, WN = 2, NT = 256;
    constexpr int WTM = 128, WTN = 128, MB = 4, NB = 4;
    constexpr int A_BYTES = BM * 64, STAGE = (BM + BN) * 64;
    constexpr int NSTAGE = 4, NINV = 8;
    static_assert(DT == DT_BF16 || DT == DT_FP16, "two-byte operands only");
    typedef typename Mfma<DT>::frag frag;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* cntq = (uint32_t*)(smem + NSTAGE * STAGE);
    float* invn_s = (float*)(smem + NSTAGE * STAGE + BN * 4);  // [NINV][256], by tile sequence number
    float* stat_s = invn_s + NINV * 256;                        // [NINV][16]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

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

    // uniform by construction; readfirstlane makes it provable (the 64-bit divisions run on the VALU)
    const uint32_t jlo = __builtin_amdgcn_readfirstlane((uint32_t)(((uint64_t)split * a.cnt) / a.nsplit));
    const uint32_t jhi = __builtin_amdgcn_readfirstlane((uint32_t)(((uint64_t)(split + 1) * a.cnt) / a.nsplit));
    const uint32_t KC = a.kc;
    const uint32_t UNITS = KC;  // loop bodies (two k-steps each) per tile
    const uint32_t total = (jhi - jlo) * KC;  // pipeline stages of this workgroup
    const uint32_t ntl = jhi - jlo;
    if (total == 0) {  // uniform: nothing to stream for this workgroup
        for (int i = tid; i < BN; i += NT) a.cand_cnt[(uint64_t)g * BN + i] = 0;
        return;
    }

    const uint32_t t_first = __builtin_amdgcn_readfirstlane(stage_tile(a.T1, a.R, a.P, a.j0 + jlo) - a.T1);
    auto next_tile = [&](uint32_t t) {
        const uint32_t u = t + a.P;  // P < R <= 2^24 tiles: no overflow
        return u >= a.R ? u - a.R : u;
    };

    // ---- DMA issue side: three stages ahead of the consume side --------------------------------------
    constexpr int RS_FLAGS = 0x00020000;
    const uint32_t voff = (uint32_t)lane * 16u;
    uint32_t lkc = 0, issued = 0, lt = t_first;
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(a.rows + (uint64_t)(a.T1 + lt) * KC * BLOCK_BYTES), 0,
                                                                   0x7fffffff, RS_FLAGS);
    const __amdgpu_buffer_rsrc_t rsB =
        __builtin_amdgcn_make_buffer_rsrc((void*)(a.qrows + (uint64_t)qt * KC * BLOCK_BYTES), 0, 0x7fffffff, RS_FLAGS);
    uint32_t d_so = 0;   // scalar offset of the stage being issued (chunk * 16 KiB + wave * 4 KiB)
    char* d_dst = smem;  // LDS base of this wave's share of it
#define CGV_DMA(RS, DST, IMM) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(RS, (__attribute__((address_space(3))) void*)(DST), 16, voff, d_so, IMM, 0)
    // piece Q of the stage: 0..3 = KiB 0..3 of this wave's share of the A block, 4..7 = of the B block
#define CGV_ISSUE(Q)                                                                                     \
    {                                                                                                    \
        if (!(ABL & 2)) {                                                                                \
            if ((Q) == 0) {                                                                              \
                d_so = lkc * BLOCK_BYTES + (uint32_t)wave * 4096u;                                       \
                d_dst = smem + (issued & (NSTAGE - 1)) * STAGE + wave * 4096;                            \
            }                                                                                            \
            if ((Q) == 0) CGV_DMA(rsA, d_dst, 0);                                                        \
            if ((Q) == 1) CGV_DMA(rsA, d_dst, 1024);                                                     \
            if ((Q) == 2) CGV_DMA(rsA, d_dst, 2048);                                                     \
            if ((Q) == 3) CGV_DMA(rsA, d_dst, 3072);                                                     \
            if ((Q) == 4) CGV_DMA(rsB, d_dst + A_BYTES, 0);                                              \
            if ((Q) == 5) CGV_DMA(rsB, d_dst + A_BYTES, 1024);                                           \
            if ((Q) == 6) CGV_DMA(rsB, d_dst + A_BYTES, 2048);                                           \
            if ((Q) == 7) CGV_DMA(rsB, d_dst + A_BYTES, 3072);                                           \
        }                                                                                                \
        if ((Q) == 7) {                                                                                  \
            ++issued;                                                                                    \
            /* the stream never ends: past the last stage it re-reads the last one into the free slot */ \
            if (issued < total && ++lkc == KC) {                                                         \
                lkc = 0;                                                                                 \
                lt = next_tile(lt);                                                                      \
                rsA = __builtin_amdgcn_make_buffer_rsrc(                                                 \
                    (void*)(a.rows + (uint64_t)(a.T1 + lt) * KC * BLOCK_BYTES), 0, 0x7fffffff, RS_FLAGS); \
            }                                                                                            \
        }                                                                                                \
    }
    // side data of the tile with sequence number seq (absolute tile T1 + tt): 256 inverse norms by wave 0,
    // 8 + 8 block norm bounds by 4 lanes of wave 1. Issued at the tile boundary where the tile starts; its
    // epilogue runs >= 3 stages later, behind a counted wait that leaves fewer DMA instructions in flight
    // than were issued after these (4 + 8 (KC - 1) >= 16 for KC >= 3).
    auto issue_side = [&](uint32_t tt, uint32_t seq) {
        if (ABL & 2) return;
        if (wave == 0)
            glds16((const char*)a.invn_c + (uint64_t)(a.T1 + tt) * 1024 + lane * 16, (char*)(invn_s + (seq & (NINV - 1)) * 256));
        if (wave == 1 && lane < 4) {
            const float* sp = ((lane & 2) ? a.blk_max : a.blk_min) + (uint64_t)(a.T1 + tt) * 8 + (lane & 1) * 4;
            glds16((const char*)sp, (char*)(stat_s + (seq & (NINV - 1)) * 16));
        }
    };

    // fragment read offsets (bytes): row r = base32 + (lane&31); the lane's piece of k-step kk is
    // c = 2*kk + (lane>>5). Piece c sits at slot c ^ ((r>>2)&3).
    const uint32_t key = (uint32_t)(lane >> 2) & 3u, hh = (uint32_t)(lane >> 5);
    const uint32_t xo[2] = {(hh ^ key) << 4, ((2 + hh) ^ key) << 4};
    const uint32_t aoff = (uint32_t)(wm * WTM + (lane & 31)) * 64;
    const uint32_t boff = (uint32_t)A_BYTES + (uint32_t)(wn * WTN + (lane & 31)) * 64;

    f32x16_t acc[MB][NB];
    f32x16_t zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.0f;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = zero16;

    frag fa0[MB], fb0[NB], fa1[MB], fb1[NB];
    // one fragment = one ds_read_b128 (piece xo[KK] of the stage)
#define CGV_LDF(F, ADDR, KK) (F) = *(const frag*)((ADDR) + xo[KK]);
#define CGV_LDA(FA, I, BASE, KK) if (!(ABL & 8)) CGV_LDF(FA[I], (BASE) + aoff + (I) * 2048, KK)
#define CGV_LDB(FB, I, BASE, KK) if (!(ABL & 8)) CGV_LDF(FB[I], (BASE) + boff + (I) * 2048, KK)
#define CGV_MMA(MBI, NBI, FA, FB) acc[MBI][NBI] = Mfma<DT>::mma(FA[MBI], FB[NBI], acc[MBI][NBI]);
#define CGV_MMAZ(MBI, NBI, FA, FB) acc[MBI][NBI] = Mfma<DT>::mma(FA[MBI], FB[NBI], zero16);
    // Program-order pins. An MFMA is a pure register operation: instruction selection is free to place it
    // anywhere its operands allow, sched_barrier or not (the first build of this kernel had the phase's first
    // MFMA sunk below the fragment reads, so its lgkmcnt(0) waited for the reads just issued). An empty asm
    // that consumes a result (the MFMA writing it comes before this point) or redefines an operand (the MFMAs
    // reading it come after this point) ties them to the chain of side-effecting instructions (LDS reads,
    // DMA, barrier, waits: "memory"), which keeps its order. One gap = what is issued between MFMA (MBI, NBI)
    // and the next one.
#define CGV_PIN_OPERAND(X) asm volatile("" : "+v"(X)::"memory");
    // all 8 fragments of the k-step named at its start: hipcc places its (single) lgkmcnt wait for them HERE,
    // where they were issued a whole k-step ago, instead of in front of their first use in the middle of the
    // phase, where it would also wait for the reads just issued
#define CGV_PIN_ALL(FA, FB)                                                                             \
    asm volatile("" : "+v"(FA[0]), "+v"(FA[1]), "+v"(FA[2]), "+v"(FA[3]), "+v"(FB[0]), "+v"(FB[1]),     \
                      "+v"(FB[2]), "+v"(FB[3])::"memory");
#define CGV_GAP(MBI, NBI, NEXT_OPERAND, ACTION)                        \
    asm volatile("" : "+a"(acc[MBI][NBI])::"memory");                  \
    ACTION;                                                            \
    CGV_PIN_OPERAND(NEXT_OPERAND)
#define CGV_NOP_ACTION
    // the DMA pieces of a k-step, by gap: 4 pieces (Q0..Q0+3) in gaps 10, 12, 14, 16
#define CGV_DMAS(G, Q0)                                                       \
    {                                                                         \
        if ((G) == 10) CGV_ISSUE(Q0);                                         \
        if ((G) == 12) CGV_ISSUE((Q0) + 1);                                   \
        if ((G) == 14) CGV_ISSUE((Q0) + 2);                                   \
        if ((G) == 16) CGV_ISSUE((Q0) + 3);                                   \
    }
    // One k-step: 16 MFMAs on fragments FA/FB; NA/NB (the other buffer) are filled for the next k-step from
    // LDS stage NBASE (its k-step NKK); FIRST = what follows the first MFMA (the stage's counted
    // wait + barrier when the next k-step starts a new stage).
#define CGV_KSTEP(MMA, FA, FB, NA, NB_, NBASE, NKK, Q0, FIRST)                                \
    {                                                                                         \
        CGV_PIN_ALL(FA, FB)                                                                   \
        MMA(0, 0, FA, FB) CGV_GAP(0, 0, FB[1], FIRST; CGV_LDA(NA, 0, NBASE, NKK))             \
        MMA(0, 1, FA, FB) CGV_GAP(0, 1, FB[2], CGV_LDB(NB_, 0, NBASE, NKK))                   \
        MMA(0, 2, FA, FB) CGV_GAP(0, 2, FB[3], CGV_LDB(NB_, 1, NBASE, NKK))                   \
        MMA(0, 3, FA, FB) CGV_GAP(0, 3, FA[1], CGV_LDB(NB_, 2, NBASE, NKK))                   \
        MMA(1, 0, FA, FB) CGV_GAP(1, 0, FB[1], CGV_LDB(NB_, 3, NBASE, NKK))                   \
        MMA(1, 1, FA, FB) CGV_GAP(1, 1, FB[2], CGV_LDA(NA, 1, NBASE, NKK))                    \
        MMA(1, 2, FA, FB) CGV_GAP(1, 2, FB[3], CGV_LDA(NA, 2, NBASE, NKK))                    \
        MMA(1, 3, FA, FB) CGV_GAP(1, 3, FA[2], CGV_LDA(NA, 3, NBASE, NKK))                    \
        MMA(2, 0, FA, FB) CGV_GAP(2, 0, FB[1], CGV_DMAS(9, Q0))                               \
        MMA(2, 1, FA, FB) CGV_GAP(2, 1, FB[2], CGV_DMAS(10, Q0))                              \
        MMA(2, 2, FA, FB) CGV_GAP(2, 2, FB[3], CGV_DMAS(11, Q0))                              \
        MMA(2, 3, FA, FB) CGV_GAP(2, 3, FA[3], CGV_DMAS(12, Q0))                              \
        MMA(3, 0, FA, FB) CGV_GAP(3, 0, FB[1], CGV_DMAS(13, Q0))                              \
        MMA(3, 1, FA, FB) CGV_GAP(3, 1, FB[2], CGV_DMAS(14, Q0))                              \
        MMA(3, 2, FA, FB) CGV_GAP(3, 2, FB[3], CGV_DMAS(15, Q0))                              \
        MMA(3, 3, FA, FB) CGV_GAP(3, 3, NA[0], CGV_DMAS(16, Q0))                              \
    }
    // The tile boundary with FOLD: the zero-C k-step of the NEXT tile with the epilogue of the PREVIOUS one folded into its
    // gaps, block by block: copy the block's 16 accumulators to VGPRs (v_accvgpr_read), issue the zero-C MFMA that overwrites
    // them, and run the fast filter (+ the cold slow path) on the copy while that MFMA occupies the matrix pipe.
    f32x16_t etmp;
    float emn[MB], emx[MB];
    uint32_t ep_tile = 0, ep_row0 = 0;
    const float* ep_invn = invn_s;
#define CGV_EPI_READ(MBI, NBI)                          \
    if (!(ABL & 1)) {                                   \
        etmp = acc[MBI][NBI];                           \
        asm volatile("" : "+v"(etmp)::"memory");        \
    }
#define CGV_EPI_TEST(MBI, NBI)                                                                                        \
    if (!(ABL & 1)) {                                                                                                 \
        const float t_ = block_threshold(a, tq[NBI], emn[MBI], emx[MBI]);                                             \
        if (__builtin_expect(ep_row0 + (uint32_t)((MBI) * 32) < a.n && block_max(etmp) > t_, 0))                      \
            block_hits<BM, BN>(a, etmp, t_, tauv[NBI], invq[NBI], (uint32_t)(wm * WTM + (MBI) * 32),                  \
                               (uint32_t)(wn * WTN + (NBI) * 32 + (lane & 31)), ep_tile, lane, g, qt, cntq, ep_invn); \
    }
#define CGV_EGAP(MBI, NBI, NEXT_OPERAND, ACTION)                       \
    asm volatile("" : "+a"(acc[MBI][NBI])::"memory");                  \
    ACTION;                                                            \
    CGV_EPI_TEST(MBI, NBI)                                             \
    CGV_PIN_OPERAND(NEXT_OPERAND)
#define CGV_KSTEP_EPI(FA, FB, NA, NB_, NBASE, NKK, Q0, FIRST)                                                   \
    {                                                                                                           \
        CGV_PIN_ALL(FA, FB)                                                                                     \
        CGV_EPI_READ(0, 0) CGV_MMAZ(0, 0, FA, FB) CGV_EGAP(0, 0, FB[1], FIRST; CGV_LDA(NA, 0, NBASE, NKK))      \
        CGV_EPI_READ(0, 1) CGV_MMAZ(0, 1, FA, FB) CGV_EGAP(0, 1, FB[2], CGV_LDB(NB_, 0, NBASE, NKK))            \
        CGV_EPI_READ(0, 2) CGV_MMAZ(0, 2, FA, FB) CGV_EGAP(0, 2, FB[3], CGV_LDB(NB_, 1, NBASE, NKK))            \
        CGV_EPI_READ(0, 3) CGV_MMAZ(0, 3, FA, FB) CGV_EGAP(0, 3, FA[1], CGV_LDB(NB_, 2, NBASE, NKK))            \
        CGV_EPI_READ(1, 0) CGV_MMAZ(1, 0, FA, FB) CGV_EGAP(1, 0, FB[1], CGV_LDB(NB_, 3, NBASE, NKK))            \
        CGV_EPI_READ(1, 1) CGV_MMAZ(1, 1, FA, FB) CGV_EGAP(1, 1, FB[2], CGV_LDA(NA, 1, NBASE, NKK))             \
        CGV_EPI_READ(1, 2) CGV_MMAZ(1, 2, FA, FB) CGV_EGAP(1, 2, FB[3], CGV_LDA(NA, 2, NBASE, NKK))             \
        CGV_EPI_READ(1, 3) CGV_MMAZ(1, 3, FA, FB) CGV_EGAP(1, 3, FA[2], CGV_LDA(NA, 3, NBASE, NKK))             \
        CGV_EPI_READ(2, 0) CGV_MMAZ(2, 0, FA, FB) CGV_EGAP(2, 0, FB[1], CGV_DMAS(9, Q0))                        \
        CGV_EPI_READ(2, 1) CGV_MMAZ(2, 1, FA, FB) CGV_EGAP(2, 1, FB[2], CGV_DMAS(10, Q0))                       \
        CGV_EPI_READ(2, 2) CGV_MMAZ(2, 2, FA, FB) CGV_EGAP(2, 2, FB[3], CGV_DMAS(11, Q0))                       \
        CGV_EPI_READ(2, 3) CGV_MMAZ(2, 3, FA, FB) CGV_EGAP(2, 3, FA[3], CGV_DMAS(12, Q0))                       \
        CGV_EPI_READ(3, 0) CGV_MMAZ(3, 0, FA, FB) CGV_EGAP(3, 0, FB[1], CGV_DMAS(13, Q0))                       \
        CGV_EPI_READ(3, 1) CGV_MMAZ(3, 1, FA, FB) CGV_EGAP(3, 1, FB[2], CGV_DMAS(14, Q0))                       \
        CGV_EPI_READ(3, 2) CGV_MMAZ(3, 2, FA, FB) CGV_EGAP(3, 2, FB[3], CGV_DMAS(15, Q0))                       \
        CGV_EPI_READ(3, 3) CGV_MMAZ(3, 3, FA, FB) CGV_EGAP(3, 3, NA[0], CGV_DMAS(16, Q0))                       \
    }
    // counted wait: the DMA instructions of the two stages behind the one being published may stay in flight
#define CGV_STAGE_SYNC                                                                        \
    if (!(ABL & 16)) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                        \
    if (!(ABL & 4)) __builtin_amdgcn_s_barrier()
    // A phase = k-step 0 of stage SA_ (fragments fa0/fb0), filling fa1/fb1 from the same stage's k-step 1;
    // B phase = k-step 1 of the previous stage (fa1/fb1), the stage barrier, filling fa0/fb0 from stage SB_ (= SA_).
#define CGV_A_PHASE(MMA, SA_) CGV_KSTEP(MMA, fa0, fb0, fa1, fb1, SA_, 1, 4, CGV_NOP_ACTION)
#define CGV_B_PHASE(SB_) CGV_KSTEP(CGV_MMA, fa1, fb1, fa0, fb0, SB_, 0, 0, CGV_STAGE_SYNC)
#define CGV_EPILOGUE(TILE, SEQ)                                                                                    \
    if (!(ABL & 1))                                                                                                \
        tile_epilogue<BM, BN, WTM, WTN, MB, NB, 0, true>(a, acc, TILE, wm, wn, lane, g, qt, tq, tauv, invq, cntq,         \
                                                      invn_s + ((SEQ) & (NINV - 1)) * 256,                         \
                                                      stat_s + ((SEQ) & (NINV - 1)) * 16);
    // LDS stage the B / A phase of loop body s fills its fragments from
    auto stage_b = [&](uint32_t s) { return smem + (s & (NSTAGE - 1)) * STAGE; };
    auto stage_a = stage_b;

    // ---- prologue: side data of the first tile, three stages in flight ---------------------------------
    issue_side(t_first, 0);
#pragma unroll 1
    for (int i = 0; i < NSTAGE - 1; ++i) {
        CGV_ISSUE(0) CGV_ISSUE(1) CGV_ISSUE(2) CGV_ISSUE(3) CGV_ISSUE(4) CGV_ISSUE(5) CGV_ISSUE(6) CGV_ISSUE(7)
    }
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // stage 0 (and the side data before it) landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my cntq zero-stores done
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < MB; ++i) CGV_LDF(fa0[i], smem + aoff + i * 2048, 0)
#pragma unroll
    for (int i = 0; i < NB; ++i) CGV_LDF(fb0[i], smem + boff + i * 2048, 0)
    if (ABL & 8) {  // timing only: fragments read ONCE (real data: zero operands would raise the clock), never refreshed
#pragma unroll
        for (int i = 0; i < MB; ++i) fa1[i] = fa0[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) fb1[i] = fb0[i];
    }

    // Loop body s = [B phase][A phase]; a tile is UNITS bodies and starts with an A phase (zero-C MFMAs) in the
    // straight-line tile-boundary block. Body s covers k-step 1 of stage s-1 and k-step 0 of stage s;
    // the B phase's barrier frees slot (s-1)&3 for the DMA of stage s+3 and publishes stage s. DMA lead: 3 stages.
    uint32_t ct = t_first, s = 1;
    CGV_ISSUE(0) CGV_ISSUE(1) CGV_ISSUE(2) CGV_ISSUE(3)  // first half of stage 3 -> slot 3 (never used so far)
    CGV_A_PHASE(CGV_MMAZ, stage_a(0));
#pragma unroll 1
    for (uint32_t u = 1; u < UNITS; ++u, ++s) {  // rest of the first tile
        CGV_B_PHASE(stage_b(s));
        CGV_A_PHASE(CGV_MMA, stage_a(s));
    }
#pragma unroll 1
    for (uint32_t tl = 1; tl < ntl; ++tl) {
        {
            CGV_B_PHASE(stage_b(s));
            const uint32_t nt = next_tile(ct);
            issue_side(nt, tl);  // the tile that starts here
            if constexpr (FOLD) {
                // the ended tile's epilogue rides in the gaps of the starting tile's zero-C k-step
                ep_tile = a.T1 + ct;
                ep_row0 = ep_tile * (uint32_t)BM + (uint32_t)(wm * WTM);
                ep_invn = invn_s + ((tl - 1) & (NINV - 1)) * 256;
                const float* st_ = stat_s + ((tl - 1) & (NINV - 1)) * 16 + (wm * WTM) / 32;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    emn[mb] = st_[mb];
                    emx[mb] = st_[8 + mb];
                }
                ct = nt;
                CGV_KSTEP_EPI(fa0, fb0, fa1, fb1, stage_a(s), 1, 4, CGV_NOP_ACTION);
            } else {
                CGV_EPILOGUE(a.T1 + ct, tl - 1);
                ct = nt;
                CGV_A_PHASE(CGV_MMAZ, stage_a(s));
            }
            ++s;
        }
#pragma unroll 1
        for (uint32_t u = 1; u < UNITS; ++u, ++s) {
            CGV_B_PHASE(stage_b(s));
            CGV_A_PHASE(CGV_MMA, stage_a(s));
        }
    }
    // tail: the last k-step of the last tile, then its epilogue
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = Mfma<DT>::mma(fa1[mb], fb1[nb], acc[mb][nb]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the dummy DMA tail (and a short tile's side data)
    __builtin_amdgcn_s_barrier();
    CGV_EPILOGUE(a.T1 + ct, ntl - 1);
#undef CGV_EPILOGUE
#undef CGV_KSTEP_EPI
#undef CGV_EGAP
#undef CGV_EPI_TEST
#undef CGV_EPI_READ
#undef CGV_B_PHASE
#undef CGV_A_PHASE
#undef CGV_A_SYNC
#undef CGV_STAGE_SYNC
#undef CGV_KSTEP
#undef CGV_DMAS
#undef CGV_NOP_ACTION
#undef CGV_GAP
#undef CGV_PIN_ALL
#undef CGV_PIN_OPERAND
#undef CGV_MMAZ
#undef CGV_MMA
#undef CGV_LDB
#undef CGV_LDA
#undef CGV_LDF
#undef CGV_ISSUE
#undef CGV_DMA

    __syncthreads();
    for (int i = tid; i < BN; i += NT) {
        const uint32_t c = cntq[i];
        a.cand_cnt[(uint64_t)g * BN + i] = c < CAND_CAPS ? c : CAND_CAPS;
    }
}

}  // namespace cgv
