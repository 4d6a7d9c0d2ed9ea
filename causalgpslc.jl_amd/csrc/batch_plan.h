// Chunk sizing of the ensemble driver (run_predict, api.hip) as plain host arithmetic — no HIP in here, so that the CPU test
// suite can compile and exercise it (tests/c/batch_plan_test.cpp).
//   per        bytes of workspace per posterior sample of a chunk (tiles of A, inverted blocks, sums, ...)
//   unit       bytes per (sample, level) unit of the full-ITE-covariance sub-batch (W, CovITE, draw workspaces); 0 = no unit B
//   lvl_extra  bytes per intervention level beyond the sub-batch of the level-sweep draw staging (0 = none)
//   want_b / want_bb   the chunk / sub-batch sizes the schedule would like (0 < want_b; want_bb = 0 without unit B)
// Rules: the sub-batch never exceeds the call's S * L pairs (an S = 1, L = 1 draws call needs ONE unit's workspace, not 128);
// everything — chunk, sub-batch, staging — fits 70 % of (free + already held) memory per stream; the AUTOMATIC chunk (no
// gpslc_set_tuning(max_batch)) additionally keeps the per-sample part of ONE stream's chunk under 30 % of the device (the arenas
// never shrink and the process usually shares the device; a caller who asks for more streams gets the same chunk per stream as
// long as the 70 % rule allows it: the operating point the measurements describe does not move with the stream count); when not even one sample fits beside the wanted sub-batch, the sub-batch
// is halved until it does.  ok = false: one sample + one unit do not fit at all.
#pragma once
#include <algorithm>
#include <cstddef>

struct BatchPlan {
    long long Bt = 0;        // posterior samples per chunk
    long long Bb = 0;        // (sample, level) pairs per unit-B sub-batch (0 without unit B)
    size_t fixed = 0;        // bytes beside the per-sample part: sub-batch + staging + slack
    bool ok = false;
};

inline BatchPlan plan_batch(long long want_b, long long want_bb, long long S, long long L, size_t per, size_t unit, size_t lvl_extra,
                            size_t free_b, size_t held_b, size_t total_b, int nstreams, bool automatic) {
    BatchPlan p;
    const long long Lc = std::max<long long>(L, 1);
    long long bb = unit > 0 ? std::max<long long>(1, std::min(want_bb, S * Lc)) : 0;
    const double hard = 0.70 * ((double)free_b + (double)held_b) / (double)std::max(1, nstreams);
    const double soft = 0.30 * (double)total_b;
    for (;;) {
        const size_t extra = (lvl_extra > 0 && Lc > bb) ? (size_t)(Lc - bb) * lvl_extra : 0;
        const size_t fixed = (size_t)bb * unit + extra + ((size_t)1 << 20);
        double budget = hard - (double)fixed;
        if (automatic) budget = std::min(budget, soft);
        const long long cap = budget > 0 ? (long long)(budget / (double)per) : 0;
        if (cap >= 1) {
            p.Bt = std::max<long long>(1, std::min(std::min(want_b, cap), S));
            p.Bb = unit > 0 ? std::min(bb, p.Bt * Lc) : 0;
            const size_t extra2 = (lvl_extra > 0 && Lc > p.Bb && p.Bb > 0) ? (size_t)(Lc - p.Bb) * lvl_extra : extra;
            p.fixed = (size_t)p.Bb * unit + extra2 + ((size_t)1 << 20);
            p.ok = true;
            return p;
        }
        if (bb <= 1) return p;       // not even one sample beside one unit
        bb = (bb + 1) / 2;
    }
}
