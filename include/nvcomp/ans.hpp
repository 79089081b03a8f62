/* nvcomp/ans.hpp -- ANSManager (reference call site: benchmarks/benchmark_hlif.cpp:195). */
#pragma once

#include "nvcomp/ans.h"
#include "nvcomp/nvcompManager.hpp"

namespace nvcomp {

struct ANSManager : BatchedManager
{
  ANSManager(size_t uncomp_chunk_size, const nvcompBatchedANSOpts_t& format_opts = nvcompBatchedANSDefaultOpts,
             hipStream_t user_stream = 0, const int device_id = 0, ChecksumPolicy checksum_policy = NoComputeNoVerify)
      : BatchedManager(kANS, uncomp_chunk_size, &format_opts, sizeof(format_opts), user_stream, device_id,
                       checksum_policy)
  {
  }
};

} // namespace nvcomp
