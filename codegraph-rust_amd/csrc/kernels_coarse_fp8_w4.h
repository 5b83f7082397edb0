// kernels_coarse_fp8_w4.h — the fp8 coarse kernel as ONE WAVE PER SIMD: 4 waves per workgroup, each owning a
// 128 x 128 output tile (4 x 4 blocks of the 32x32x64 block-scaled MFMA) of the same 256 x 256 workgroup tile,
// the 256 accumulator registers in the accumulator half (AGPRs) of the 512-entry register file.
//
// Why: the 8-wave fp8 kernel (kernels_coarse_fp8.h) has 128 architectural VGPRs beside its 128 accumulators,
// not enough for double-buffered K=64 fragments (8 VGPRs per operand block): its A halves are single-buffered
// and its accumulators are cleared per tile. One wave per SIMD has 256 + 256: fragments fully double-buffered
// (2 x 8 blocks x 8 VGPRs), zero-C MFMAs at the tile start, 16 reads per 16 MFMAs (64 cycles each), and every
// gap between two MFMAs carries either two fragment reads or one DMA piece.
// Everything else follows experiments/kernels_coarse_w4.h / kernels_coarse.h: B32 blocked operands, 4-slot LDS ring filled
// by buffer_load ... lds three stages ahead and retired by a counted vmcnt, the fused threshold top-k' epilogue
// (tile_epilogue), XCD-aware workgroup mapping (block_to_work). A stage is ONE K=64 k-step, so both phases of a
// loop body are whole stages with their own counted wait + barrier (needs an even kc >= 4; the host falls back
// to the 8-wave kernel otherwise).
// Row scales (HISTORY.md §4.1): the E8M0 scale operand of a lane is 127 - e of its row, 4 corpus bytes (one per
// 32-row block of the wave tile) + 4 query bytes. The corpus bytes of a tile are needed by its FIRST MFMA, so a
// tile's 256 exponents are DMA'd into the 8-deep LDS ring one tile boundary EARLIER than its inverse norms
// (which only the epilogue needs).
#pragma once
#include "kernels_coarse.h"
#include "kernels_coarse_fp8.h"

namespace cgv {

// ABL: timing-only ablation mask (results are WRONG for ABL != 0; CGV_ABLATE_W4): 1 = skip the epilogue,
// 2 = skip the DMA, 4 = skip the barrier, 8 = read the fragments once (real data) and never again.
// SI = 2 (kc % 4 == 0): static issue side + loop unrolled by the ring size, as in kernels_coarse.h - the DMA stream's ring slot
// and the fragment reads' LDS addresses are constants of the unrolled iteration, the chunk offset one running scalar, the tile
// switch of the stream happens in a known iteration (the last body of a tile); 0 = the dynamic form (any even kc >= 4).
// EPI2 (round 5): the epilogue spread over TWO k-steps. v_accvgpr_read_b32 issues at 6 cycles per wave64 instruction on this part
// (scripts/ubench/accread.hip: 16 reads = 96 cycles), so one 32 x 32 block's epilogue - 16 reads + ~13 VALU of filter - is
// ~148 cycles against the 64-cycle MFMA it sits in front of: the EPI k-step runs VALU-bound, ~1340 exposed cycles per tile (the
// no-epilogue ablation: MFMA busy 0.675 -> 0.882). A block's accumulators are final once its MFMA of the tile's LAST k-step has
// run, so blocks 0..7 (serpentine order) are read and filtered in the gaps of that k-step already - read in gap 2i + 1, filtered
// in gap 2i + 2, two 16-register copies alternating - and blocks 8..15 in the zero-C k-step, each read at least one gap before
// the MFMA that overwrites it: one read (96 cycles) or one filter (~52) per gap instead of both.
template <int ABL = 0, int SI = 0, bool EPI2 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void coarse_fp8s_w4_kernel(const CoarseArgs a) {
    constexpr bool DUMP = false;
    constexpr int BM = 256, BN = 256, WN = 2, NT = 256;
    constexpr int WTM = 128, WTN = 128, MB = 4, NB = 4;
    constexpr int A_BYTES = BM * 64, STAGE = (BM + BN) * 64;
    constexpr int NSTAGE = 4, NINV = 8;
    typedef Fp8Frag frag;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* cntq = (uint32_t*)(smem + NSTAGE * STAGE);
    float* invn_s = (float*)(smem + NSTAGE * STAGE + BN * 4);  // [NINV][256], by tile sequence number
    float* stat_s = invn_s + NINV * 256;                        // [NINV][16]
    int8_t* rexp_s = (int8_t*)(stat_s + NINV * 16);             // [NINV][256] scale exponents, by tile sequence number

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    uint32_t qt, split;
    const uint32_t g = block_to_work(a, qt, split);
    const Pace pace = pace_init(a, g, qt);

    for (int i = tid; i < BN; i += NT) cntq[i] = 0;
    if ((!DUMP) && a.append_splits != 0u) {   // uniform: the first launch behind an emitting sample (kernels_coarse.h)
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
    const uint32_t UNITS = KC / 2;            // loop bodies (two stages each) per tile
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
    uint32_t si_slot = 0;                          // SI: ring slot (bytes) of the stage being issued, set by the caller
    uint32_t si_so = (uint32_t)wave * 4096u;       // SI: chunk offset within the tile + this wave's 4 KiB slab
    auto issue_switch_tile = [&]() {               // SI: the issue side enters the next tile of the visiting order
        lt = next_tile(lt);
        rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(a.rows + (uint64_t)(a.T1 + lt) * KC * BLOCK_BYTES), 0, 0x7fffffff,
                                                RS_FLAGS);
        si_so = (uint32_t)wave * 4096u;
    };
#define CGV_ISSUE(Q)                                                                                     \
    {                                                                                                    \
        if (!(ABL & 2)) {                                                                                \
            if ((Q) == 0) {                                                                              \
                if (SI != 0) {                                                                           \
                    d_so = si_so;                                                                        \
                    d_dst = smem + si_slot + wave * 4096;                                                \
                } else {                                                                                 \
                    d_so = lkc * BLOCK_BYTES + (uint32_t)wave * 4096u;                                   \
                    d_dst = smem + (issued & (NSTAGE - 1)) * STAGE + wave * 4096;                        \
                }                                                                                        \
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
        if ((Q) == 7 && SI != 0) si_so += BLOCK_BYTES;                                                   \
        if ((Q) == 7 && SI == 0) {                                                                       \
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
    // than were issued after these. The tile's scale exponents (wave 2, 16 lanes) go one boundary earlier.
    auto issue_side = [&](uint32_t tt, uint32_t seq) {
        if (ABL & 2) return;
        if (wave == 0)
            glds16((const char*)a.invn_c + (uint64_t)(a.T1 + tt) * 1024 + lane * 16, (char*)(invn_s + (seq & (NINV - 1)) * 256));
        if (wave == 1 && lane < 4) {
            const float* sp = ((lane & 2) ? a.blk_max : a.blk_min) + (uint64_t)(a.T1 + tt) * 8 + (lane & 1) * 4;
            glds16((const char*)sp, (char*)(stat_s + (seq & (NINV - 1)) * 16));
        }
    };

    auto issue_rexp = [&](uint32_t tt, uint32_t seq) {
        if (ABL & 2) return;
        if (wave == 2 && lane < 16)
            glds16((const char*)a.rexp_c + (uint64_t)(a.T1 + tt) * 256 + lane * 16, (char*)(rexp_s + (seq & (NINV - 1)) * 256));
    };
    // E8M0 scale bytes (127 - e = 2^-e): sb = the lane's four query columns (byte nb), constant for the
    // workgroup; sa = its four corpus rows of the current tile (byte mb), re-read at every tile boundary
    int sb = 0;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const uint32_t q = qt * BN + wn * WTN + nb * 32 + (lane & 31);
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

    // fragment read offsets (bytes): row r = base32 + (lane&31); the lane's 32 bytes of a stage are pieces
    // 2h, 2h+1 (h = lane>>5). Piece c sits at slot c ^ ((r>>2)&3).
    const uint32_t key = (uint32_t)(lane >> 2) & 3u, hh = (uint32_t)(lane >> 5);
    const uint32_t xo[2] = {((2 * hh) ^ key) << 4, ((2 * hh + 1) ^ key) << 4};
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
    // one fragment = two ds_read_b128 (both pieces of the lane's 32 bytes)
#define CGV_LDF(F, ADDR, KK)                                \
    {                                                       \
        (F).p0 = *(const i32x4_t*)((ADDR) + xo[0]);         \
        (F).p1 = *(const i32x4_t*)((ADDR) + xo[1]);         \
    }
#define CGV_LDA(FA, I, BASE, KK) if (!(ABL & 8)) CGV_LDF(FA[I], (BASE) + aoff + (I) * 2048, KK)
#define CGV_LDB(FB, I, BASE, KK) if (!(ABL & 8)) CGV_LDF(FB[I], (BASE) + boff + (I) * 2048, KK)
#define CGV_MMA(MBI, NBI, FA, FB) acc[MBI][NBI] = mma_fp8_k64_scaled<MBI, NBI>(FA[MBI], FB[NBI], acc[MBI][NBI], sa, sb);
#define CGV_MMAZ(MBI, NBI, FA, FB) acc[MBI][NBI] = mma_fp8_k64_scaled<MBI, NBI>(FA[MBI], FB[NBI], zero16, sa, sb);
    // Program-order pins. An MFMA is a pure register operation: instruction selection is free to place it
    // anywhere its operands allow, sched_barrier or not (the first build of this kernel had the phase's first
    // MFMA sunk below the fragment reads, so its lgkmcnt(0) waited for the reads just issued). An empty asm
    // that consumes a result (the MFMA writing it comes before this point) or redefines an operand (the MFMAs
    // reading it come after this point) ties them to the chain of side-effecting instructions (LDS reads,
    // DMA, barrier, waits: "memory"), which keeps its order. One gap = what is issued between MFMA (MBI, NBI)
    // and the next one.
#define CGV_PIN_OPERAND(X) asm volatile("" : "+v"((X).p0)::"memory");
    // all 8 fragments of the k-step named at its start: hipcc places its (single) lgkmcnt wait for them HERE,
    // where they were issued a whole k-step ago, instead of in front of their first use in the middle of the
    // phase, where it would also wait for the reads just issued
#define CGV_PIN_ALL(FA, FB)                                                                                  \
    asm volatile("" : "+v"(FA[0].p0), "+v"(FA[0].p1), "+v"(FA[1].p0), "+v"(FA[1].p1), "+v"(FA[2].p0),        \
                      "+v"(FA[2].p1), "+v"(FA[3].p0), "+v"(FA[3].p1), "+v"(FB[0].p0), "+v"(FB[0].p1),        \
                      "+v"(FB[1].p0), "+v"(FB[1].p1), "+v"(FB[2].p0), "+v"(FB[2].p1), "+v"(FB[3].p0),        \
                      "+v"(FB[3].p1)::"memory");
#define CGV_GAP(MBI, NBI, NEXT_OPERAND, ACTION)                        \
    asm volatile("" : "+a"(acc[MBI][NBI])::"memory");                  \
    ACTION;                                                            \
    CGV_PIN_OPERAND(NEXT_OPERAND)
#define CGV_NOP_ACTION
    // the DMA pieces of a stage, by gap: all 8 pieces in gaps 9..16
#define CGV_DMAS(G, Q0) CGV_ISSUE((G)-9)
    // One k-step: 16 MFMAs on fragments FA/FB in serpentine block order (consecutive MFMAs share one operand block); NA/NB (the other buffer) are filled for the next k-step from
    // LDS stage NBASE; FIRST = what follows the first MFMA (the stage's counted
    // wait + barrier when the next k-step starts a new stage).
#define CGV_KSTEP(MMA, FA, FB, NA, NB_, NBASE, NKK, Q0, FIRST)                                \
    {                                                                                         \
        CGV_PIN_ALL(FA, FB)                                                                   \
        MMA(0, 0, FA, FB) CGV_GAP(0, 0, FB[1], FIRST; CGV_LDA(NA, 0, NBASE, NKK))             \
        MMA(0, 1, FA, FB) CGV_GAP(0, 1, FB[2], CGV_LDB(NB_, 0, NBASE, NKK))                   \
        MMA(0, 2, FA, FB) CGV_GAP(0, 2, FB[3], CGV_LDB(NB_, 1, NBASE, NKK))                   \
        MMA(0, 3, FA, FB) CGV_GAP(0, 3, FA[1], CGV_LDB(NB_, 2, NBASE, NKK))                   \
        MMA(1, 3, FA, FB) CGV_GAP(1, 3, FB[2], CGV_LDB(NB_, 3, NBASE, NKK))                   \
        MMA(1, 2, FA, FB) CGV_GAP(1, 2, FB[1], CGV_LDA(NA, 1, NBASE, NKK))                    \
        MMA(1, 1, FA, FB) CGV_GAP(1, 1, FB[0], CGV_LDA(NA, 2, NBASE, NKK))                    \
        MMA(1, 0, FA, FB) CGV_GAP(1, 0, FA[2], CGV_LDA(NA, 3, NBASE, NKK))                    \
        MMA(2, 0, FA, FB) CGV_GAP(2, 0, FB[1], CGV_DMAS(9, Q0))                               \
        MMA(2, 1, FA, FB) CGV_GAP(2, 1, FB[2], CGV_DMAS(10, Q0))                              \
        MMA(2, 2, FA, FB) CGV_GAP(2, 2, FB[3], CGV_DMAS(11, Q0))                              \
        MMA(2, 3, FA, FB) CGV_GAP(2, 3, FA[3], CGV_DMAS(12, Q0))                              \
        MMA(3, 3, FA, FB) CGV_GAP(3, 3, FB[2], CGV_DMAS(13, Q0))                              \
        MMA(3, 2, FA, FB) CGV_GAP(3, 2, FB[1], CGV_DMAS(14, Q0))                              \
        MMA(3, 1, FA, FB) CGV_GAP(3, 1, FB[0], CGV_DMAS(15, Q0))                              \
        MMA(3, 0, FA, FB) CGV_GAP(3, 0, NA[0], CGV_DMAS(16, Q0))                              \
    }
    // The tile boundary: the zero-C k-step of the NEXT tile with the epilogue of the PREVIOUS one folded into its
    // gaps, block by block: copy the block's 16 accumulators to VGPRs (v_accvgpr_read), issue the zero-C MFMA that
    // overwrites them, and run the fast filter (+ the cold slow path) on the copy while that MFMA occupies the
    // matrix pipe for 64 cycles. With one wave per SIMD nothing else would cover the epilogue's ~500 VALU
    // instructions per tile (measured: 2.2 of 12.5 ms on the C5-mini main launch when run as a separate block).
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
        CGV_EPI_READ(1, 3) CGV_MMAZ(1, 3, FA, FB) CGV_EGAP(1, 3, FB[2], CGV_LDB(NB_, 3, NBASE, NKK))            \
        CGV_EPI_READ(1, 2) CGV_MMAZ(1, 2, FA, FB) CGV_EGAP(1, 2, FB[1], CGV_LDA(NA, 1, NBASE, NKK))             \
        CGV_EPI_READ(1, 1) CGV_MMAZ(1, 1, FA, FB) CGV_EGAP(1, 1, FB[0], CGV_LDA(NA, 2, NBASE, NKK))             \
        CGV_EPI_READ(1, 0) CGV_MMAZ(1, 0, FA, FB) CGV_EGAP(1, 0, FA[2], CGV_LDA(NA, 3, NBASE, NKK))             \
        CGV_EPI_READ(2, 0) CGV_MMAZ(2, 0, FA, FB) CGV_EGAP(2, 0, FB[1], CGV_DMAS(9, Q0))                        \
        CGV_EPI_READ(2, 1) CGV_MMAZ(2, 1, FA, FB) CGV_EGAP(2, 1, FB[2], CGV_DMAS(10, Q0))                       \
        CGV_EPI_READ(2, 2) CGV_MMAZ(2, 2, FA, FB) CGV_EGAP(2, 2, FB[3], CGV_DMAS(11, Q0))                       \
        CGV_EPI_READ(2, 3) CGV_MMAZ(2, 3, FA, FB) CGV_EGAP(2, 3, FA[3], CGV_DMAS(12, Q0))                       \
        CGV_EPI_READ(3, 3) CGV_MMAZ(3, 3, FA, FB) CGV_EGAP(3, 3, FB[2], CGV_DMAS(13, Q0))                       \
        CGV_EPI_READ(3, 2) CGV_MMAZ(3, 2, FA, FB) CGV_EGAP(3, 2, FB[1], CGV_DMAS(14, Q0))                       \
        CGV_EPI_READ(3, 1) CGV_MMAZ(3, 1, FA, FB) CGV_EGAP(3, 1, FB[0], CGV_DMAS(15, Q0))                       \
        CGV_EPI_READ(3, 0) CGV_MMAZ(3, 0, FA, FB) CGV_EGAP(3, 0, NA[0], CGV_DMAS(16, Q0))                       \
    }
    // ---- EPI2: the same epilogue over the tile's last k-step + the zero-C k-step (two alternating copies etmp / etmp2) ----
    f32x16_t etmp2;
#define CGV_E2_READ(SLOT, MBI, NBI)                     \
    if (!(ABL & 1)) {                                   \
        SLOT = acc[MBI][NBI];                           \
        asm volatile("" : "+v"(SLOT)::"memory");        \
    }
#define CGV_E2_TEST(SLOT, MBI, NBI)                                                                                   \
    if (!(ABL & 1)) {                                                                                                 \
        const float t_ = block_threshold(a, tq[NBI], emn[MBI], emx[MBI]);                                             \
        if (__builtin_expect(ep_row0 + (uint32_t)((MBI) * 32) < a.n && block_max(SLOT) > t_, 0))                      \
            block_hits<BM, BN>(a, SLOT, t_, tauv[NBI], invq[NBI], (uint32_t)(wm * WTM + (MBI) * 32),                  \
                               (uint32_t)(wn * WTN + (NBI) * 32 + (lane & 31)), ep_tile, lane, g, qt, cntq, ep_invn); \
    }
    // the tile's LAST k-step (a B phase: fa1 / fb1 -> fa0 / fb0 from NBASE): block b_i final after MFMA i; R(b_i) in gap 2i + 1,
    // T(b_i) in gap 2i + 2 for i = 0..7 (T(b_7) in the first gap of the zero-C k-step)
#define CGV_KSTEP_LASTF(FA, FB, NA, NB_, NBASE, NKK, Q0, FIRST)                                                          \
    {                                                                                                                    \
        CGV_PIN_ALL(FA, FB)                                                                                              \
        CGV_MMA(0, 0, FA, FB) CGV_GAP(0, 0, FB[1], FIRST; CGV_LDA(NA, 0, NBASE, NKK))                                    \
        CGV_MMA(0, 1, FA, FB) CGV_GAP(0, 1, FB[2], CGV_LDB(NB_, 0, NBASE, NKK); CGV_E2_READ(etmp, 0, 0))                 \
        CGV_MMA(0, 2, FA, FB) CGV_GAP(0, 2, FB[3], CGV_LDB(NB_, 1, NBASE, NKK); CGV_E2_TEST(etmp, 0, 0))                 \
        CGV_MMA(0, 3, FA, FB) CGV_GAP(0, 3, FA[1], CGV_LDB(NB_, 2, NBASE, NKK); CGV_E2_READ(etmp2, 0, 1))                \
        CGV_MMA(1, 3, FA, FB) CGV_GAP(1, 3, FB[2], CGV_LDB(NB_, 3, NBASE, NKK); CGV_E2_TEST(etmp2, 0, 1))                \
        CGV_MMA(1, 2, FA, FB) CGV_GAP(1, 2, FB[1], CGV_LDA(NA, 1, NBASE, NKK); CGV_E2_READ(etmp, 0, 2))                  \
        CGV_MMA(1, 1, FA, FB) CGV_GAP(1, 1, FB[0], CGV_LDA(NA, 2, NBASE, NKK); CGV_E2_TEST(etmp, 0, 2))                  \
        CGV_MMA(1, 0, FA, FB) CGV_GAP(1, 0, FA[2], CGV_LDA(NA, 3, NBASE, NKK); CGV_E2_READ(etmp2, 0, 3))                 \
        CGV_MMA(2, 0, FA, FB) CGV_GAP(2, 0, FB[1], CGV_DMAS(9, Q0); CGV_E2_TEST(etmp2, 0, 3))                            \
        CGV_MMA(2, 1, FA, FB) CGV_GAP(2, 1, FB[2], CGV_DMAS(10, Q0); CGV_E2_READ(etmp, 1, 3))                            \
        CGV_MMA(2, 2, FA, FB) CGV_GAP(2, 2, FB[3], CGV_DMAS(11, Q0); CGV_E2_TEST(etmp, 1, 3))                            \
        CGV_MMA(2, 3, FA, FB) CGV_GAP(2, 3, FA[3], CGV_DMAS(12, Q0); CGV_E2_READ(etmp2, 1, 2))                           \
        CGV_MMA(3, 3, FA, FB) CGV_GAP(3, 3, FB[2], CGV_DMAS(13, Q0); CGV_E2_TEST(etmp2, 1, 2))                           \
        CGV_MMA(3, 2, FA, FB) CGV_GAP(3, 2, FB[1], CGV_DMAS(14, Q0); CGV_E2_READ(etmp, 1, 1))                            \
        CGV_MMA(3, 1, FA, FB) CGV_GAP(3, 1, FB[0], CGV_DMAS(15, Q0); CGV_E2_TEST(etmp, 1, 1))                            \
        CGV_MMA(3, 0, FA, FB) CGV_GAP(3, 0, NA[0], CGV_DMAS(16, Q0); CGV_E2_READ(etmp2, 1, 0))                           \
    }
    // the zero-C k-step behind it: T(b_7), then blocks 8..15, each READ at least one gap before the MFMA that overwrites it
#define CGV_KSTEP_EPI2(FA, FB, NA, NB_, NBASE, NKK, Q0, FIRST)                                                           \
    {                                                                                                                    \
        CGV_PIN_ALL(FA, FB)                                                                                              \
        CGV_MMAZ(0, 0, FA, FB) CGV_GAP(0, 0, FB[1], FIRST; CGV_LDA(NA, 0, NBASE, NKK); CGV_E2_TEST(etmp2, 1, 0))         \
        CGV_MMAZ(0, 1, FA, FB) CGV_GAP(0, 1, FB[2], CGV_LDB(NB_, 0, NBASE, NKK); CGV_E2_READ(etmp, 2, 0))                \
        CGV_MMAZ(0, 2, FA, FB) CGV_GAP(0, 2, FB[3], CGV_LDB(NB_, 1, NBASE, NKK); CGV_E2_TEST(etmp, 2, 0))                \
        CGV_MMAZ(0, 3, FA, FB) CGV_GAP(0, 3, FA[1], CGV_LDB(NB_, 2, NBASE, NKK); CGV_E2_READ(etmp2, 2, 1))               \
        CGV_MMAZ(1, 3, FA, FB) CGV_GAP(1, 3, FB[2], CGV_LDB(NB_, 3, NBASE, NKK); CGV_E2_TEST(etmp2, 2, 1))               \
        CGV_MMAZ(1, 2, FA, FB) CGV_GAP(1, 2, FB[1], CGV_LDA(NA, 1, NBASE, NKK); CGV_E2_READ(etmp, 2, 2))                 \
        CGV_MMAZ(1, 1, FA, FB) CGV_GAP(1, 1, FB[0], CGV_LDA(NA, 2, NBASE, NKK); CGV_E2_TEST(etmp, 2, 2))                 \
        CGV_MMAZ(1, 0, FA, FB) CGV_GAP(1, 0, FA[2], CGV_LDA(NA, 3, NBASE, NKK); CGV_E2_READ(etmp2, 2, 3))                \
        CGV_MMAZ(2, 0, FA, FB) CGV_GAP(2, 0, FB[1], CGV_DMAS(9, Q0); CGV_E2_TEST(etmp2, 2, 3))                           \
        CGV_MMAZ(2, 1, FA, FB) CGV_GAP(2, 1, FB[2], CGV_DMAS(10, Q0); CGV_E2_READ(etmp, 3, 3))                           \
        CGV_MMAZ(2, 2, FA, FB) CGV_GAP(2, 2, FB[3], CGV_DMAS(11, Q0); CGV_E2_TEST(etmp, 3, 3))                           \
        CGV_MMAZ(2, 3, FA, FB) CGV_GAP(2, 3, FA[3], CGV_DMAS(12, Q0); CGV_E2_READ(etmp2, 3, 2))                          \
        CGV_MMAZ(3, 3, FA, FB) CGV_GAP(3, 3, FB[2], CGV_DMAS(13, Q0); CGV_E2_TEST(etmp2, 3, 2))                          \
        CGV_MMAZ(3, 2, FA, FB) CGV_GAP(3, 2, FB[1], CGV_DMAS(14, Q0); CGV_E2_READ(etmp, 3, 1))                           \
        CGV_MMAZ(3, 1, FA, FB) CGV_GAP(3, 1, FB[0], CGV_DMAS(15, Q0); CGV_E2_TEST(etmp, 3, 1); CGV_E2_READ(etmp2, 3, 0)) \
        CGV_MMAZ(3, 0, FA, FB) CGV_GAP(3, 0, NA[0], CGV_DMAS(16, Q0); CGV_E2_TEST(etmp2, 3, 0))                          \
    }
    // counted wait: the DMA instructions of the stage behind the one being published may stay in flight
    // (one stage = 8; the phase's own 8 pieces are issued after the wait)
#define CGV_STAGE_SYNC                                  \
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    \
    if (!(ABL & 4)) __builtin_amdgcn_s_barrier()
    // A phase = an even stage of the tile (fa0/fb0), barrier, filling fa1/fb1 from the NEXT stage SA_;
    // B phase = an odd stage (fa1/fb1), barrier, filling fa0/fb0 from the next stage SB_.
#define CGV_A_PHASE(MMA, SA_) CGV_KSTEP(MMA, fa0, fb0, fa1, fb1, SA_, 1, 4, CGV_STAGE_SYNC)
#define CGV_B_PHASE(SB_) CGV_KSTEP(CGV_MMA, fa1, fb1, fa0, fb0, SB_, 0, 0, CGV_STAGE_SYNC)
#define CGV_EPILOGUE(TILE, SEQ)                                                                                    \
    if (!(ABL & 1))                                                                                                \
        tile_epilogue<BM, BN, WTM, WTN, MB, NB, DUMP, true>(a, acc, TILE, wm, wn, lane, g, qt, tq, tauv, invq, cntq,      \
                                                      invn_s + ((SEQ) & (NINV - 1)) * 256,                         \
                                                      stat_s + ((SEQ) & (NINV - 1)) * 16);
    // LDS stage the B / A phase of loop body s fills its fragments from
    auto stage_b = [&](uint32_t s) { return smem + ((2 * s) & (NSTAGE - 1)) * STAGE; };
    auto stage_a = [&](uint32_t s) { return smem + ((2 * s + 1) & (NSTAGE - 1)) * STAGE; };

    // ---- prologue: side data of the first tile (and the second tile's exponents), three stages in flight --
    issue_side(t_first, 0);
    issue_rexp(t_first, 0);
    issue_rexp(next_tile(t_first), 1);
#pragma unroll 1
    for (int i = 0; i < NSTAGE - 1; ++i) {
        si_slot = (uint32_t)(i * STAGE);
        CGV_ISSUE(0) CGV_ISSUE(1) CGV_ISSUE(2) CGV_ISSUE(3) CGV_ISSUE(4) CGV_ISSUE(5) CGV_ISSUE(6) CGV_ISSUE(7)
    }
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // stage 0 (and the side data before it) landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my cntq zero-stores done
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < MB; ++i) CGV_LDF(fa0[i], smem + aoff + i * 2048, 0)
#pragma unroll
    for (int i = 0; i < NB; ++i) CGV_LDF(fb0[i], smem + boff + i * 2048, 0)
    load_sa(0);
    if (ABL & 8) {  // timing only: fragments read ONCE (real data: zero operands would raise the clock), never refreshed
#pragma unroll
        for (int i = 0; i < MB; ++i) fa1[i] = fa0[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) fb1[i] = fb0[i];
    }

    // Loop body s = [B phase][A phase]; a tile is UNITS bodies and starts with an A phase (zero-C MFMAs) in the
    // straight-line tile-boundary block. Body s covers stages 2s-1 and 2s, each phase publishing the next stage.
    // DMA lead: 3 stages.
    uint32_t ct = t_first, s = 1;
    si_slot = (uint32_t)(3 * STAGE);
    CGV_A_PHASE(CGV_MMAZ, stage_a(0));
    // SI: body parity P (0: stages in slots 0, 1; 1: slots 2, 3). The B phase computes stage 2s - 1, fills from stage 2s and
    // issues stage 2s + 2; the A phase computes 2s, fills from 2s + 1, issues 2s + 3. Every tile starts with an even body.
#define CGV_BODY_AT(P)                                                            \
    {                                                                             \
        si_slot = (uint32_t)(((2 * (P) + 2) & (NSTAGE - 1)) * STAGE);             \
        CGV_B_PHASE(smem + (2 * (P)) * STAGE);                                    \
        si_slot = (uint32_t)(((2 * (P) + 3) & (NSTAGE - 1)) * STAGE);             \
        CGV_A_PHASE(CGV_MMA, smem + (2 * (P) + 1) * STAGE);                       \
    }
    // the stream crosses into the next tile in the LAST body of a tile (its B phase issues stage KC = the next tile's first)
#define CGV_TILE_REST_SI                                                          \
    {                                                                             \
        const uint32_t ng = UNITS >> 1;                                           \
        if (ng == 1) issue_switch_tile();                                         \
        CGV_BODY_AT(1)                                                            \
        _Pragma("unroll 1") for (uint32_t gi = 1; gi < ng; ++gi) {                \
            CGV_BODY_AT(0)                                                        \
            if (gi + 1 == ng) issue_switch_tile();                                \
            CGV_BODY_AT(1)                                                        \
        }                                                                         \
    }
    if constexpr (SI != 0) {
        CGV_TILE_REST_SI
    } else {
#pragma unroll 1
        for (uint32_t u = 1; u < UNITS; ++u, ++s) {  // rest of the first tile
            CGV_B_PHASE(stage_b(s));
            CGV_A_PHASE(CGV_MMA, stage_a(s));
        }
    }
#pragma unroll 1
    for (uint32_t tl = 1; tl < ntl; ++tl) {
        {
            si_slot = (uint32_t)(2 * STAGE);
            if constexpr (EPI2) {   // the ending tile's epilogue starts inside its last k-step: its constants first
                ep_tile = a.T1 + ct;
                ep_row0 = ep_tile * (uint32_t)BM + (uint32_t)(wm * WTM);
                ep_invn = invn_s + ((tl - 1) & (NINV - 1)) * 256;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    emn[mb] = stat_s[((tl - 1) & (NINV - 1)) * 16 + (wm * WTM) / 32 + mb];
                    emx[mb] = stat_s[((tl - 1) & (NINV - 1)) * 16 + 8 + (wm * WTM) / 32 + mb];
                }
                if constexpr (SI != 0) {
                    CGV_KSTEP_LASTF(fa1, fb1, fa0, fb0, smem, 0, 0, CGV_STAGE_SYNC);
                } else {
                    CGV_KSTEP_LASTF(fa1, fb1, fa0, fb0, stage_b(s), 0, 0, CGV_STAGE_SYNC);
                }
            } else if constexpr (SI != 0) {
                CGV_B_PHASE(smem);
            } else {
                CGV_B_PHASE(stage_b(s));
            }
            if (wave == 0) pace_step(pace, tl + 1, lane);  // keep the split's workgroups within the L2's reach
            const uint32_t nt = next_tile(ct);
            issue_side(nt, tl);                   // the tile that starts here
            issue_rexp(next_tile(nt), tl + 1);    // exponents of the one after it
            // epilogue of tile ct (sequence number tl - 1), folded into the zero-C k-step of tile nt
            if constexpr (!EPI2) {
                ep_tile = a.T1 + ct;
                ep_row0 = ep_tile * (uint32_t)BM + (uint32_t)(wm * WTM);
                ep_invn = invn_s + ((tl - 1) & (NINV - 1)) * 256;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    emn[mb] = stat_s[((tl - 1) & (NINV - 1)) * 16 + (wm * WTM) / 32 + mb];
                    emx[mb] = stat_s[((tl - 1) & (NINV - 1)) * 16 + 8 + (wm * WTM) / 32 + mb];
                }
            }
            ct = nt;
            load_sa(tl);  // issued at the previous boundary: a whole tile of counted waits + barriers ago
            si_slot = (uint32_t)(3 * STAGE);
            if constexpr (EPI2) {
                if constexpr (SI != 0) {
                    CGV_KSTEP_EPI2(fa0, fb0, fa1, fb1, smem + STAGE, 1, 4, CGV_STAGE_SYNC);
                } else {
                    CGV_KSTEP_EPI2(fa0, fb0, fa1, fb1, stage_a(s), 1, 4, CGV_STAGE_SYNC);
                }
            } else if constexpr (SI != 0) {
                CGV_KSTEP_EPI(fa0, fb0, fa1, fb1, smem + STAGE, 1, 4, CGV_STAGE_SYNC);
            } else {
                CGV_KSTEP_EPI(fa0, fb0, fa1, fb1, stage_a(s), 1, 4, CGV_STAGE_SYNC);
            }
            ++s;
        }
        if constexpr (SI != 0) {
            CGV_TILE_REST_SI
        } else {
#pragma unroll 1
            for (uint32_t u = 1; u < UNITS; ++u, ++s) {
                CGV_B_PHASE(stage_b(s));
                CGV_A_PHASE(CGV_MMA, stage_a(s));
            }
        }
    }
#undef CGV_TILE_REST_SI
#undef CGV_BODY_AT
    // tail: the last k-step of the last tile, then its epilogue
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int v = (mb == 0 ? sa : mb == 1 ? sa >> 8 : mb == 2 ? sa >> 16 : sa >> 24);
            const int w = (nb == 0 ? sb : nb == 1 ? sb >> 8 : nb == 2 ? sb >> 16 : sb >> 24);
            acc[mb][nb] = mma_fp8_k64_scaled<0, 0>(fa1[mb], fb1[nb], acc[mb][nb], v, w);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the dummy DMA tail (and a short tile's side data)
    __builtin_amdgcn_s_barrier();
    CGV_EPILOGUE(a.T1 + ct, ntl - 1);
#undef CGV_EPILOGUE
#undef CGV_B_PHASE
#undef CGV_A_PHASE
#undef CGV_STAGE_SYNC
#undef CGV_KSTEP
#undef CGV_KSTEP_EPI
#undef CGV_KSTEP_EPI2
#undef CGV_KSTEP_LASTF
#undef CGV_E2_TEST
#undef CGV_E2_READ
#undef CGV_EGAP
#undef CGV_EPI_TEST
#undef CGV_EPI_READ
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
    if (tid == 0) pace_done(pace);
    for (int i = tid; i < BN; i += NT) {
        const uint32_t c = cntq[i];
        a.cand_cnt[(uint64_t)g * BN + i] = c < CAND_CAPS ? c : CAND_CAPS;
    }
}

}  // namespace cgv
