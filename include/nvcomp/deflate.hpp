/* nvcomp/deflate.hpp -- DeflateManager (reference call site: benchmarks/benchmark_hlif.cpp:208-209). The chunks inside
 * the container are raw DEFLATE streams (nvcomp/deflate.h); chunk sizes of at most 65 536 bytes. */
#pragma once

#include "nvcomp/deflate.h"
#include "nvcomp/nvcompManager.hpp"

namespace nvcomp {

struct DeflateManager : BatchedManager
{
  DeflateManager(size_t uncomp_chunk_size, const nvcompBatchedDeflateOpts_t& format_opts = nvcompBatchedDeflateDefaultOpts,
                 hipStream_t user_stream = 0, const int device_id = 0, ChecksumPolicy checksum_policy = NoComputeNoVerify)
      : BatchedManager(kDeflate, uncomp_chunk_size, &format_opts, sizeof(format_opts), user_stream, device_id,
                       checksum_policy)
  {
  }
};

} // namespace nvcomp
