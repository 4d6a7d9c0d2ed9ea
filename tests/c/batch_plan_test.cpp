// CPU test of the chunk-sizing arithmetic of the ensemble driver (causalgpslc.jl_amd/csrc/batch_plan.h): compiled and
// run by tests/test_batch_plan.py.  Prints one line per case: "name Bt Bb ok fixed".
#include <cstdio>
#include "../../causalgpslc.jl_amd/csrc/batch_plan.h"

static void show(const char* name, const BatchPlan& p) {
    std::printf("%s %lld %lld %d %zu\n", name, p.Bt, p.Bb, p.ok ? 1 : 0, p.fixed);
}

int main() {
    const size_t GB = (size_t)1 << 30;
    // N = 4096: 561 + 32 tiles of 128 KiB and the small vectors per sample; a unit-B pair holds W (1024 tiles) + CovITE (528 tiles)
    const size_t per = (size_t)(561 + 32) * 131072 + 3 * 32768 + 2 * 32 * 32768, unit = (size_t)(1024 + 528) * 131072 + 600000;
    const size_t tot = 288 * GB, fre = 280 * GB;
    // unit A (no unit B), automatic chunk: 1,024 samples whatever the stream count up to 2; fewer, not an error, beyond
    show("unitA_1stream", plan_batch(1024, 0, 5000, 1, per, 0, 0, fre, 0, tot, 1, true));
    show("unitA_2streams", plan_batch(1024, 0, 5000, 1, per, 0, 0, fre, 0, tot, 2, true));
    show("unitA_4streams", plan_batch(1024, 0, 5000, 1, per, 0, 0, fre, 0, tot, 4, true));
    // a draws call with ONE sample and one level needs one unit's workspace, on any stream count
    show("draws_S1_1stream", plan_batch(1024, 128, 1, 1, per, unit, 32768 * 10, fre, 0, tot, 1, true));
    show("draws_S1_4streams", plan_batch(1024, 128, 1, 1, per, unit, 32768 * 10, fre, 0, tot, 4, true));
    // 64 samples x 1 level with draws on 4 streams: the old rule (30 % of the device per stream MINUS the sub-batch) refused this
    show("draws_S64_4streams", plan_batch(1024, 128, 64, 1, per, unit, 32768 * 10, fre, 0, tot, 4, true));
    // level sweep longer than the sub-batch: the staging of the extra levels is part of the fixed bytes
    show("sweep_S8_L200", plan_batch(1024, 128, 8, 200, per, unit, 32768 * 10, fre, 0, tot, 1, true));
    // a 64 GB device with 20 GB free: the sub-batch is halved until a sample fits beside it
    show("small_device", plan_batch(1024, 128, 64, 16, per, unit, 32768 * 10, 20 * GB, 0, 64 * GB, 1, true));
    // nothing fits: one sample + one unit exceed what is free
    show("too_small", plan_batch(1024, 128, 64, 16, per, unit, 32768 * 10, (size_t)200 << 20, 0, 64 * GB, 1, true));
    // gpslc_set_tuning(max_batch = 2000): only what is free counts (70 %)
    show("explicit_2000", plan_batch(2000, 0, 5000, 1, per, 0, 0, fre, 0, tot, 1, false));
    return 0;
}
