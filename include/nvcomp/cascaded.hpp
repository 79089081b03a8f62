/* nvcomp/cascaded.hpp -- CascadedManager (reference call site: benchmarks/benchmark_hlif.cpp:199-205). */
#pragma once

#include "nvcomp/cascaded.h"
#include "nvcomp/nvcompManager.hpp"

namespace nvcomp {

struct CascadedManager : BatchedManager
{
  CascadedManager(size_t uncomp_chunk_size,
                  const nvcompBatchedCascadedOpts_t& format_opts = nvcompBatchedCascadedDefaultOpts,
                  hipStream_t user_stream = 0, const int device_id = 0,
                  ChecksumPolicy checksum_policy = NoComputeNoVerify)
      : BatchedManager(kCascaded, uncomp_chunk_size, &format_opts, sizeof(format_opts), user_stream, device_id,
                       checksum_policy)
  {
  }
};

} // namespace nvcomp
