/* nvcomp/bitcomp.hpp -- BitcompManager (reference call site: benchmarks/benchmark_hlif.cpp:193). */
#pragma once

#include "nvcomp/bitcomp.h"
#include "nvcomp/nvcompManager.hpp"

namespace nvcomp {

struct BitcompManager : BatchedManager
{
  BitcompManager(size_t uncomp_chunk_size,
                 const nvcompBatchedBitcompFormatOpts& format_opts = nvcompBatchedBitcompDefaultOpts,
                 hipStream_t user_stream = 0, const int device_id = 0,
                 ChecksumPolicy checksum_policy = NoComputeNoVerify)
      : BatchedManager(kBitcomp, uncomp_chunk_size, &format_opts, sizeof(format_opts), user_stream, device_id,
                       checksum_policy)
  {
  }
};

} // namespace nvcomp
