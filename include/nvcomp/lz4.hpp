/* nvcomp/lz4.hpp -- LZ4Manager (reference call sites: benchmarks/benchmark_allgather.cpp:496,
 * benchmarks/benchmark_hlif.cpp:189, examples/nvcomp_gds.cu:132, examples/high_level_quickstart_example.cpp:76). */
#pragma once

#include "nvcomp/lz4.h"
#include "nvcomp/nvcompManager.hpp"

namespace nvcomp {

struct LZ4Manager : BatchedManager
{
  LZ4Manager(size_t uncomp_chunk_size, const nvcompBatchedLZ4Opts_t& format_opts = nvcompBatchedLZ4DefaultOpts,
             hipStream_t user_stream = 0, const int device_id = 0, ChecksumPolicy checksum_policy = NoComputeNoVerify)
      : BatchedManager(kLZ4, uncomp_chunk_size, &format_opts, sizeof(format_opts), user_stream, device_id, checksum_policy)
  {
  }
};

} // namespace nvcomp
