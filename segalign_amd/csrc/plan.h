// plan.h -- the iteration plan SeedAndFilter derives from the hit prefix (src/seed_filter.cu:718-745).
#pragma once
#include <stdint.h>
namespace sa {
constexpr uint32_t PLAN_MAX_ITER = 1000;
struct IterPlan {
    uint64_t num_hits;                 // inclusive prefix of the last seed (:716)
    uint32_t num_iter;                 // iterations to run (0 when there is nothing to do)
    uint32_t overflow;                 // != 0: plan needs this many iterations (> PLAN_MAX_ITER)
    int64_t limit_pos[PLAN_MAX_ITER];  // last seed index of iteration i (-1 = reference's wrapped index, H5)
    uint64_t upto[PLAN_MAX_ITER];      // inclusive hit prefix at limit_pos[i]
};
}  // namespace sa
