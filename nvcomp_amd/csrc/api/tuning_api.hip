/* api/tuning_api.hip -- the knobs of include/nvcomp/amd_ext.h. */
#include "nvcomp/amd_ext.h"

#include "common/tuning.h"

namespace nvcomp_amd_tuning {
size_t lz_index_min_batch = NVCOMP_AMD_LZ_INDEX_MIN_BATCH_DEFAULT;
size_t lz_pair_max_batch = NVCOMP_AMD_LZ_PAIR_MAX_BATCH_DEFAULT;
}

extern "C" size_t nvcompAmdSetLZIndexMinBatch(size_t min_batch)
{
  const size_t old = nvcomp_amd_tuning::lz_index_min_batch;
  nvcomp_amd_tuning::lz_index_min_batch = min_batch;
  return old;
}

extern "C" size_t nvcompAmdSetLZPairMaxBatch(size_t max_batch)
{
  const size_t old = nvcomp_amd_tuning::lz_pair_max_batch;
  nvcomp_amd_tuning::lz_pair_max_batch = max_batch;
  return old;
}
