/* common/tuning.h -- process-wide performance knobs (include/nvcomp/amd_ext.h); defined in api/tuning_api.hip. */
#pragma once

#include <stddef.h>

namespace nvcomp_amd_tuning {
extern size_t lz_index_min_batch;
extern size_t lz_pair_max_batch;
}
