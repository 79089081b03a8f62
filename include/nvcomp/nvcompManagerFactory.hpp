/* nvcomp/nvcompManagerFactory.hpp -- create_manager(): rebuild the right manager from a
 * compressed buffer (reference call sites: examples/high_level_quickstart_example.cpp:88,356). */
#pragma once

#include <memory>

#include "nvcomp/nvcompManager.hpp"

namespace nvcomp {

/* Reads the container header from device memory: synchronises the stream. Throws
 * std::runtime_error for a buffer this library did not produce. */
std::shared_ptr<nvcompManagerBase> create_manager(
    const uint8_t* comp_buffer, hipStream_t stream = 0, const int device_id = 0,
    ChecksumPolicy checksum_policy = NoComputeNoVerify);

} // namespace nvcomp
