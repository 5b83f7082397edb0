// plan.h - the host-side planner of a search (no device code): the error bound of the guarantee check, the number of candidates
// kept per query, the staged launch plan and the knobs of the measurement flavour. Split out of cgvec.hip in
// round 6 (VERDICT r5 'Next' 8); tests/test_plan_host.py drives it through cgv_debug_plan_ without a GPU.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/cgvec.h"

namespace cgv {

constexpr int BM = 256, BN = 256;  // coarse tile (corpus rows x queries)

inline uint32_t esize_of(int dtype) { return dtype == CGV_DTYPE_F32 ? 4u : (dtype == CGV_DTYPE_FP8E4M3 ? 1u : 2u); }

float coarse_eps_scale(uint32_t ld_coarse, uint32_t ld_exact, uint32_t k_inst, int metric, bool fp8 = false);
uint32_t query_group(uint32_t nqt, uint32_t ld, int dtype);
uint32_t kprime_of(uint32_t k, bool fp8 = false);

constexpr uint32_t BOOT_TILES = 16;  // 4096 rows scored densely by boot_kernel (25 us); the register-only select takes <= 4096
constexpr uint32_t SAMPLE_TILES_MAX = 256;  // tau_kernel takes <= 1024 group maxima per query: 16 per tile up to 64 tiles,
                                           // 8 up to 128, 4 up to 256 (sample_vals_of)
inline uint32_t sample_vals_of(uint32_t sample_tiles) { return sample_tiles <= 64 ? 16u : (sample_tiles <= 128 ? 8u : 4u); }

struct StagePlan {
    uint32_t ntiles, T1, R, P;
    uint32_t sample_tiles = 0;     // > 0: first threshold from a sample launch of the coarse kernel (plan_stages)
    bool sample_emits = false;     // the sample launch emits its own candidates: `counts` cover the tiles BEHIND it in the visiting order
    std::vector<uint32_t> counts;  // tiles per launch, in visiting order
};

uint32_t golden_stride(uint32_t R);
StagePlan plan_stages_legacy(uint64_t n, uint32_t kprime, uint32_t nsplit_max);
StagePlan plan_stages(uint64_t n, uint32_t kprime, uint32_t nqt, uint32_t n_cu, uint32_t nsplit_max, int force_m = 0,
                      bool allow_emit = false);
float plan_ladder_scale(double S_rows, double N_rows, uint32_t kprime, uint32_t hi_rank);

#ifdef CGV_ABLATE_BUILD
double env_double(const char* name, double dflt);
#endif

// Planner / host knobs. The production library runs on the defaults below, full stop; the measurement flavour
// (`make ABLATE=1`, CGV_ABLATE_BUILD) also reads them from the environment at load time and lets cgv_debug_set_() change
// them at run time (in-process A/B: scripts/ab.py).
#ifdef CGV_ABLATE_BUILD
#define CGV_ENV_INT(NAME, DFLT) (getenv(NAME) ? atoi(getenv(NAME)) : (DFLT))
#define CGV_ENV_DBL(NAME, DFLT) env_double(NAME, DFLT)
#else
#define CGV_ENV_INT(NAME, DFLT) (DFLT)
#define CGV_ENV_DBL(NAME, DFLT) (DFLT)
#endif
struct Tunables {
#ifdef CGV_ABLATE_BUILD
    int plan_legacy = getenv("CGV_PLAN") && !strcmp(getenv("CGV_PLAN"), "legacy");
    int pace = getenv("CGV_NO_PACE") ? 0 : 1;                 // soft lockstep of the coarse workgroups (Pace)
#else
    int plan_legacy = 0;
    int pace = 1;
#endif
    int sample_tiles = CGV_ENV_INT("CGV_SAMPLE_TILES", 0);    // 0 = automatic
    int plan_launches = CGV_ENV_INT("CGV_PLAN_LAUNCHES", 0);  // 0 = cost model
    double hit_us = CGV_ENV_DBL("CGV_PLAN_HIT_US", 1.7);
    double launch_us = CGV_ENV_DBL("CGV_PLAN_LAUNCH_US", 40.0);
    int zero_copy = CGV_ENV_INT("CGV_ZERO_COPY", 3);          // pinned host buffers in place: 1 queries, 2 results
    int epi = CGV_ENV_INT("CGV_EPI", 1);                      // emitting epilogue variant of the bf16 coarse kernel (A/B)
    int fuse_sample = CGV_ENV_INT("CGV_FUSE_SAMPLE", 0);      // sample + tau + first emitting launch as ONE launch (measurement flavour only)
    int sample_emit = CGV_ENV_INT("CGV_SAMPLE_EMIT", 1);      // the sample launch emits its own candidates; the launches behind it skip its tiles (A/B: 0)
    int top2 = CGV_ENV_INT("CGV_TOP2", 1);                    // small batches (nq <= 64): COARSE_TOP2, one launch without thresholds (A/B: 0)
    int launch_events = CGV_ENV_INT("CGV_LAUNCH_EVENTS", 1);  // profiling: the timed launch carries its own start / end events (A/B: 0 = marker events around it)
    int sample_repair = CGV_ENV_INT("CGV_SAMPLE_REPAIR", 1);  // emitting sample: floor violations put right inside the final kernel (A/B: 0 = exact scan)
    int top2_repair = CGV_ENV_INT("CGV_TOP2_REPAIR", 1);      // COARSE_TOP2 floor violations: re-scan the offending cells only (A/B: 0 = exact scan)
    int exact_small = CGV_ENV_INT("CGV_EXACT_SMALL", 1);      // exact scan of <= 8 queries as ONE kernel (kernels_exact_small.h; A/B: 0)
    int fetch_queries = CGV_ENV_INT("CGV_FETCH_QUERIES", 1);  // batches in flight: host queries fetched by the copy engine (A/B: 0 = converted in place)
    int self_publish = CGV_ENV_INT("CGV_SELF_PUBLISH", 1);    // <= 64 queries: the final kernel's last workgroup publishes the flags (A/B: 0)
    // threshold ladder (kernels_coarse.h; MEASUREMENT FLAVOUR ONLY - a measured negative result, profiles/r05_tau_ladder_ab.txt):
    // 0 = staged launches; 1 = ladder inside the planned launches; 2 = ladder + ONE emitting launch behind the sample
    int ladder = CGV_ENV_INT("CGV_LADDER", 0);
};
Tunables& tun();
inline bool exact_small_enabled() { return tun().exact_small != 0; }

}  // namespace cgv
