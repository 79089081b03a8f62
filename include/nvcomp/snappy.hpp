/* nvcomp/snappy.hpp -- SnappyManager (reference call site: benchmarks/benchmark_hlif.cpp:191-192). */
#pragma once

#include "nvcomp/nvcompManager.hpp"
#include "nvcomp/snappy.h"

namespace nvcomp {

struct SnappyManager : BatchedManager
{
  SnappyManager(size_t uncomp_chunk_size, const nvcompBatchedSnappyOpts_t& format_opts = nvcompBatchedSnappyDefaultOpts,
                hipStream_t user_stream = 0, const int device_id = 0, ChecksumPolicy checksum_policy = NoComputeNoVerify)
      : BatchedManager(kSnappy, uncomp_chunk_size, &format_opts, sizeof(format_opts), user_stream, device_id,
                       checksum_policy)
  {
  }
};

} // namespace nvcomp
