// plan.cpp - host-side planner of a search (plan.h): error bound, candidates per query, staged launch plan, knobs.
#include "plan.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace cgv {

// Bound on |coarse score - exact (reference-arithmetic) score| in units of |q||c| (cosine: absolute),
// u = 2^-24. DESIGN.md §5.3 derives it; the three terms are
//  (1) the MFMA accumulation, under the ALIGNED-ADDEND TRUNCATION MODEL of the matrix pipe: one
//      instruction returns C + sum of its K exact products with an absolute error of at most
//      (K + 1) * 2^-23 * max(|C|, |result|, max |product|) - every one of the K + 1 addends may lose up
//      to one unit in the last place of the largest one (truncation, not rounding). ld/K instructions
//      deep and every partial sum <= sum |x_i y_i| <= |q||c|:  (ld/K) * (K+1) * 2u;
//  (2) the coarse scaling by the two inverse norms (each 1/sqrt of an ld/64-deep fma chain + a 6-level
//      tree: relative error <= (ld/128 + 5)u) and two multiplications:  (ld/64 + 12)u;
//  (3) the reference arithmetic itself against the real-number value: AVX2 order = ld/8-deep fma chain
//      per lane + 3-level tree for the dot product and for both squared norms, sqrt, divide:
//      (ld/4 + 10)u; the sequential formula (CGV_METRIC_COSINE_SEQ, search.rs:519-533): (2 ld + 4)u.
// The model (1) is an assumption about undocumented hardware; tests/test_gpu_guarantee.py measures it
// with adversarial same-sign / alternating-sign / one-huge-many-tiny inputs, and rescore_body's
// trip-wire sends any query with an observed candidate error above eps/2 to the exact scan.
//  (1') fp8 (round 6): the block-scaled K = 64 instruction does NOT follow (1), and rounds 2-5 priced it as if it did. Measured
//      (scripts/fp8_mfma_align_probe.py, scripts/fp8_mfma_error_probe.py -> profiles/r06_fp8_mfma_error.txt): one product of
//      448 * 448 and 63 equal small ones in one K = 64 block - down to 2^-13.6 of the large product all 63 arrive; from 2^-15.6 on
//      exactly 56 of 63 arrive, whatever their sign, down to 2^-23.6: the SEVEN products that share a group of 8 with the large one
//      are dropped, the other 56 are added exactly. So the instruction sums its products in groups of 8, each group aligned to its
//      largest product and cut 13-14 bits below that product's exponent, and adds the group sums at full f32 precision. A group
//      loses < 7 * 2^-13 * max|product of the group|; the groups' largest products sum to at most |q||c| (Cauchy-Schwarz over the
//      groups): 7 * 2^-13 = 14336 u of |q||c| in all, WHATEVER D (measured worst on random data: 420 u; heavy-tailed magnitudes:
//      4650 u at D = 64, 1530 u at D = 768). The old price, (ld/16) * 34 u, was 136 u at D = 64: an fp8 index of D <= 256 could
//      see coarse errors above its eps (random data, D = 64: 2.4 x). The trip-wire caught that on candidates - those queries took
//      the exact scan - but rows OUTSIDE the candidate set were covered by the model alone. tests/test_gpu_guarantee.py now
//      measures D = 64 .. 256 and heavy-tailed / near-duplicate data too. (The non-scaled K = 16 fp8 instruction of the boot
//      kernel is priced the same way.)
//  (1'') round 6: an emitting sample tags its candidates' coarse scores in their 6 low mantissa bits (kernels_coarse.h, SAMPLE
//      mode): < 64 ulp = 2^-17 relative = 128 u of |q||c|, priced whether or not a search's sample emits.
float coarse_eps_scale(uint32_t ld_coarse, uint32_t ld_exact, uint32_t k_inst, int metric, bool fp8) {
    const double u = 5.9604644775390625e-8;
    const double n_inst = (double)((ld_coarse + k_inst - 1) / k_inst);
    const double mfma = n_inst * (double)(k_inst + 1) * 2.0 + (fp8 ? 14336.0 : 0.0) + 128.0;
    const double scale = (double)ld_coarse / 64.0 + 12.0;
    const bool sequential = metric == CGV_METRIC_COSINE_SEQ || metric == CGV_METRIC_COSINE_SCALAR;  // one accumulator per sum
    const double ref = sequential ? 2.0 * ld_exact + 4.0 : (double)ld_exact / 4.0 + 10.0;
    return (float)((mfma + scale + ref) * u * 1.0001);
}

// Query tiles per XCD (kernels_coarse.h block_to_work): the largest power of two that keeps their rows
// (256 x ld x esize bytes each) within ~1.5 MiB of the XCD's 4 MiB L2, and divides nqt. CGV_QGROUP overrides.
uint32_t query_group(uint32_t nqt, uint32_t ld, int dtype) {
#ifdef CGV_ABLATE_BUILD
    static const int forced = getenv("CGV_QGROUP") ? atoi(getenv("CGV_QGROUP")) : -1;
    if (forced >= 0) return (uint32_t)forced;
#endif
    const size_t tile = (size_t)256 * ld * esize_of(dtype);
    uint32_t g = 1;
    while (g * 2 <= nqt && nqt % (g * 2) == 0 && (size_t)(g * 2) * tile <= (3u << 19)) g *= 2;
    return g;
}

// Candidates kept per query. The check is e_k > (k'-th best coarse score) + eps: k' - k is what separates them on ordinary data.
// fp8: eps is ~8.6e-4 whatever D (coarse_eps_scale (1')) - at C5's 62.5M rows per shard the 10th and the 16th best scores of a
// query are closer than that for ~0.7 % of the queries, and every one of those costs an exact scan of 48 GB; k' = 2k + 12 (32 for
// k = 10: 22 order-statistic spacings instead of 6) puts the k'-th score ~7e-3 below the k-th there.
uint32_t kprime_of(uint32_t k, bool fp8) {
    if (fp8) return ((2u * k + 12u + 7u) / 8u) * 8u;
    uint32_t m = std::max<uint32_t>(6u, k / 8u);
    return ((k + m + 7u) / 8u) * 8u;
}


// Staged thresholds. The first threshold comes from a SAMPLE: boot_kernel scores T1 tiles'
// worth of rows (128 aligned 32-row groups spread over the corpus with a golden-ratio stride, boot_row())
// densely and select_kernel publishes their k'-th best score - a valid lower bound of the final k'-th best
// whatever the insertion order (with the first 4096 rows as the sample, a topic-sorted corpus sent 242 of 256
// queries to the exact scan). Corpora of <= T1 tiles are covered by the boot stage alone (identity map, its
// top-k' are the candidates). Otherwise the sample only sets tau and ALL R tiles are visited in the
// golden-ratio order of stage_tile() (tile j = (j * P) mod R) in a few launches of geometrically growing size: a launch covering N rows with a
// threshold learnt from C earlier rows emits about k' * N / C candidates per query, spread
// over nsplit (workgroup, query) lists of CAND_CAPS entries — N is chosen so that the expected
// list length stays at EMIT_TARGET and the per-query total at MERGE_TARGET. The last launch is
// the dominant one.
// Expected entries per (workgroup, query) list per launch. The threshold of a launch was learnt from
// `seen` rows, so k' * 1024 / seen scores of every 32 x 32 block pass the epilogue's fast filter and
// take its slow path: ~19 us per tile at 16 hits per block (seen = 1024 rows), ~8 us at 4, nothing
// at 0.2 (measured, r01d timelines). A 4096-row boot and shorter early launches keep the hit rate
// down where it matters: C2 682 k -> 726 k q/s, C3 shard +3 %, f32 + shadow (k' = 56, lower target)
// 590 k -> 632 k; C4 / C5 / small shards flat.
constexpr uint32_t EMIT_TARGET = 12;
constexpr uint32_t EMIT_TARGET_WIDE = 6;  // k' > 32 (the hit rate scales with k')
constexpr uint32_t MERGE_TARGET = 2048;  // expected candidates per query per launch (select holds 8192;
                                         // the count fluctuates by ~1/sqrt(k') around its mean)


uint32_t gcd_u32(uint32_t a, uint32_t b) {
    while (b) {
        const uint32_t t = a % b;
        a = b;
        b = t;
    }
    return a;
}

// P ~ 0.618 R coprime to R (1 for R <= 2): j -> (j * P) mod R visits every residue once, every prefix evenly spread
uint32_t golden_stride(uint32_t R) {
    if (R <= 2) return 1;
    uint32_t P = (uint32_t)((double)R * 0.6180339887498949);
    if (P < 1) P = 1;
    while (gcd_u32(P, R) != 1) ++P;
    return P;
}

StagePlan plan_stages_legacy(uint64_t n, uint32_t kprime, uint32_t nsplit_max) {
    StagePlan p;
    p.sample_tiles = 0;
    p.ntiles = (uint32_t)((n + BM - 1) / BM);
    const uint32_t boot = BOOT_TILES;
    const uint32_t emit_target = kprime > 32 ? EMIT_TARGET_WIDE : EMIT_TARGET;
    p.T1 = std::min<uint32_t>(std::max<uint32_t>((kprime + BM - 1) / BM, boot), p.ntiles);
    p.R = std::max<uint32_t>(p.ntiles, 1u);
    p.P = golden_stride(p.R);
    uint64_t seen = (uint64_t)p.T1 * BM;                    // rows behind the current threshold
    uint32_t left = p.ntiles > p.T1 ? p.ntiles : 0;         // (<= T1 tiles: the boot stage covers them)
    while (left > 0) {
        const uint32_t nsplit = std::min<uint32_t>(left, nsplit_max);
        // rows this launch may cover: k' * N / seen / nsplit <= EMIT_TARGET
        uint64_t rows = seen * std::min<uint64_t>((uint64_t)nsplit * emit_target, MERGE_TARGET) / std::max<uint32_t>(kprime, 1);
        uint32_t tiles = (uint32_t)std::min<uint64_t>(left, std::max<uint64_t>(rows / BM, nsplit));
        if (tiles * 3 >= left * 2) tiles = left;  // do not leave a small tail for another launch
        p.counts.push_back(tiles);
        left -= tiles;
        seen += (uint64_t)tiles * BM;
    }
    return p;
}

#ifdef CGV_ABLATE_BUILD   // (the production library reads nothing from the environment)
double env_double(const char* name, double dflt) {
    const char* v = getenv(name);
    return v ? atof(v) : dflt;
}
#endif

Tunables& tun() {
    static Tunables t;
    return t;
}

// Round-3 plan. The first threshold comes from a SAMPLE LAUNCH of the coarse kernel itself
// (COARSE_SAMPLE: the first S tiles of the visiting order, one tile per CU, block maxima -> tau_kernel): as many
// rows as one pass of the chip scores at tile-kernel speed (C2: 64 tiles = 16 k rows in ~25 us; the dense boot
// kernel needed 33 + 25 us for 4 k). It contributes no candidates, so the emitting launches visit ALL tiles.
// Their number m and sizes minimise a measured cost model:
//   * a launch whose threshold was learnt from `seen` rows sends k' * 1024 / seen scores of every 32 x 32 block down
//     the epilogue's slow path, ~HIT_US per tile per (hit per block) (r01d / r02 timelines: +19 us per tile at 16
//     hits per block, +8 at 4): covering N rows costs  N / seen * kappa,  kappa = nqt * k' * 4 * HIT_US / n_cu;
//   * every launch costs LAUNCH_US of ramp + select.
// With `seen` growing geometrically (ratio rho per launch, rho^m = (N + S) / S) the total is m * (LAUNCH_US +
// kappa * (rho - 1)); m is the cheapest count whose expected emissions fit the candidate lists (LIST_TARGET per
// (workgroup, query) list of CAND_CAPS entries, MERGE_TARGET per query). C2: S = 16 k rows, m = 2 (112 k + 888 k
// rows); the 125 k-row shard of C2 at 8 GPUs: ONE launch.
constexpr uint32_t LIST_TARGET = 32;
// Inverse of the standard normal distribution function (Acklam's rational approximation, |error| < 1.2e-9): the expected
// position of the final k'-th best score relative to the sample's order statistics (plan_ladder_scale).
double inv_norm_cdf(double p) {
    static const double a[] = {-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02, 1.383577518672690e+02,
                               -3.066479806614716e+01, 2.506628277459239e+00};
    static const double b[] = {-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02, 6.680131188771972e+01,
                               -1.328068155288572e+01};
    static const double c[] = {-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00, -2.549732539343734e+00,
                               4.374664141464968e+00, 2.938163982698783e+00};
    static const double d[] = {7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00, 3.754408661907416e+00};
    if (p <= 0.0) return -1e300;
    if (p >= 1.0) return 1e300;
    if (p < 0.02425) {
        const double q = sqrt(-2.0 * log(p));
        return (((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) / ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1.0);
    }
    if (p > 1.0 - 0.02425) {
        const double q = sqrt(-2.0 * log(1.0 - p));
        return -(((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) / ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1.0);
    }
    const double q = p - 0.5, r = q * q;
    return (((((a[0] * r + a[1]) * r + a[2]) * r + a[3]) * r + a[4]) * r + a[5]) * q /
           (((((b[0] * r + b[1]) * r + b[2]) * r + b[3]) * r + b[4]) * r + 1.0);
}

// Threshold ladder (kernels_coarse.h): delta = scale x (sample value at rank hi_rank - sample value at rank k'), four levels.
// Under a normal tail the score at tail probability p sits at z(p) sigmas; the sample of S rows shows z(hi/S) and z(k'/S), the
// final k'-th best of N rows is expected at z(k'/N): the ladder spans 1.25 x that distance in 4 levels. 0 = no ladder.
float plan_ladder_scale(double S_rows, double N_rows, uint32_t kprime, uint32_t hi_rank) {
    if (!(S_rows > 4.0 * kprime) || !(N_rows > S_rows) || hi_rank == 0 || hi_rank >= kprime) return 0.0f;
    const double zk = -inv_norm_cdf((double)kprime / S_rows), zh = -inv_norm_cdf((double)hi_rank / S_rows),
                 zn = -inv_norm_cdf((double)kprime / N_rows);
    if (!(zh > zk) || !(zn > zk)) return 0.0f;
    return (float)(1.25 * (zn - zk) / (zh - zk) / 4.0);
}

// allow_emit (round 6): the sample launch may emit its own candidates (kernels_coarse.h, SAMPLE mode of tile_epilogue), so that the
// emitting launches start BEHIND it in the visiting order instead of scoring its tiles again: 64 of C2's 3907 tiles, 64 of the
// 489 of its 8-GPU shard. Taken when the corpus is at least four times the sample (else the old form: the launches cover everything).
StagePlan plan_stages(uint64_t n, uint32_t kprime, uint32_t nqt, uint32_t n_cu, uint32_t nsplit_max, int force_m, bool allow_emit) {
    const Tunables& t = tun();
    StagePlan p;
    p.ntiles = (uint32_t)((n + BM - 1) / BM);
    if (t.plan_legacy || p.ntiles <= BOOT_TILES) return plan_stages_legacy(n, kprime, nsplit_max);
    const int forced_s = t.sample_tiles, forced_m = force_m > 0 ? force_m : t.plan_launches;
    const double hit_us = t.hit_us, launch_us = t.launch_us;
    // one pass of the chip: one tile per CU, but never more than 1/8 of the corpus (it is scored again by the launches)
    uint32_t S = std::min<uint32_t>(std::max<uint32_t>(n_cu / std::max<uint32_t>(nqt, 1u), 8u), SAMPLE_TILES_MAX);
    while (S > 64 && S * 8 > p.ntiles) S /= 2;
    if (forced_s > 0) S = std::min<uint32_t>((uint32_t)forced_s, SAMPLE_TILES_MAX);
    // the k'-th largest of 16 S block maxima: keep a few times k' of them
    while (S < 64 && 16u * S < 4u * kprime) S *= 2;
    S = std::min(S, p.ntiles);
    p.sample_tiles = S;
    p.T1 = 0;
    p.R = p.ntiles;
    p.P = golden_stride(p.R);
    // (Three of a query's top-(k + 1) rows in one 64-row cell of the sample fail its floor check: C(k + 1, 3) (S / R)^3 / 256^2 per
    // query - 3e-4 at R = 2 S, 4e-5 at 4 S, 6e-6 on C2's 8-GPU shard, 1e-8 on C2. While such a query cost the exact scan of the whole
    // corpus the sample only emitted from 4 S tiles on; the final kernel now re-scores the offending cell itself
    // (sample_floor_repair, kernels_select.h: one workgroup, 64 rows), so from 2 S on the tile-time saved - up to a third of the
    // GEMM of a mid-size corpus - is kept. Below 2 S the launch behind the sample would be shorter than the sample.)
    p.sample_emits = allow_emit && t.sample_emit != 0 && kprime <= 64 && p.ntiles >= 2 * S;
    const uint32_t emit_tiles = p.sample_emits ? p.ntiles - S : p.ntiles;   // tiles the emitting launches cover
    const double seen0 = (double)S * BM, total = (double)emit_tiles * BM;
    const double kappa = (double)nqt * kprime * 4.0 * hit_us / (double)std::max<uint32_t>(n_cu, 1u);
    uint32_t best_m = 0;
    double best_cost = 0.0;
    for (uint32_t m = 1; m <= 8; ++m) {
        const double rho = pow((total + seen0) / seen0, 1.0 / m);
        const double emit = kprime * (rho - 1.0);  // expected candidates per query per launch
        const bool fits = emit <= MERGE_TARGET && emit / std::min<double>(nsplit_max, total / BM / m) <= LIST_TARGET;
        const double cost = m * (launch_us + kappa * (rho - 1.0));
        if (forced_m > 0 ? m == (uint32_t)forced_m : (fits && (best_m == 0 || cost < best_cost))) {
            best_m = m;
            best_cost = cost;
        }
    }
    if (best_m == 0) best_m = 8;
    const double rho = pow((total + seen0) / seen0, 1.0 / best_m);
    const uint32_t unit = std::max<uint32_t>(n_cu / std::max<uint32_t>(nqt, 1u), 1u);  // tiles of one full pass of the chip
    uint32_t left = emit_tiles;
    double seen = seen0;
    for (uint32_t i = 0; i < best_m && left > 0; ++i) {
        uint32_t tiles = left;
        if (i + 1 < best_m) {
            tiles = (uint32_t)std::min<double>(left, std::max(1.0, seen * (rho - 1.0) / BM));
            if (tiles >= 2 * unit) tiles = (tiles + unit / 2) / unit * unit;  // whole passes: no idle CUs in the last one
            tiles = std::min(tiles, left);
            if ((uint64_t)tiles * 4 >= (uint64_t)left * 3) tiles = left;      // no small tail for another launch
        }
        p.counts.push_back(tiles);
        left -= tiles;
        seen += (double)tiles * BM;
    }
    return p;
}

}  // namespace cgv
