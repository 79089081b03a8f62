/*
 * nvcomp/nvcompManager.hpp -- high-level C++ interface (HLIF), MI355X build.
 *
 * Interface reconstructed from the reference's call sites (the header itself is
 * not part of the reference tree):
 *   configure_compression / compress / configure_decompression (2 overloads) /
 *   decompress / get_compressed_output_size
 *       doc/highlevel_cpp_quickstart.md:84,99-102,120,132,146-149
 *       benchmarks/benchmark_hlif.hpp:68,84,101,120,137
 *   CompressionConfig::max_compressed_buffer_size      benchmarks/benchmark_hlif.hpp:70
 *   DecompressionConfig::decomp_data_size, get_status() benchmarks/benchmark_hlif.hpp:122,
 *                                                       examples/high_level_quickstart_example.cpp:313
 *   ChecksumPolicy (5 modes)                            examples/high_level_quickstart_example.cpp:252-282
 * A manager cuts one contiguous buffer into chunks, runs the batched low-level
 * codec on them ("HLIF now dispatches to LLIF", CHANGELOG.md:17) and writes a
 * self-describing container, so that create_manager(comp_buffer) can rebuild the
 * right manager from the compressed bytes alone. The container layout is this
 * library's own (DESIGN.md "HLIF container"). Errors are reported by exceptions
 * (doc/highlevel_cpp_quickstart.md:3-5).
 */
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>

#include <hip/hip_runtime_api.h>

#include "nvcomp/shared_types.h"

namespace nvcomp {

enum ChecksumPolicy
{
  NoComputeNoVerify = 0,
  ComputeAndNoVerify = 1,
  NoComputeAndVerifyIfPresent = 2,
  ComputeAndVerifyIfPresent = 3,
  ComputeAndVerify = 4
};

namespace detail {
struct StatusWord; /* pinned host word the device writes the batch status into */
struct ManagerImpl;
} // namespace detail

struct CompressionConfig
{
  size_t uncompressed_buffer_size = 0;
  size_t max_compressed_buffer_size = 0;
  size_t num_chunks = 0;
  /* nvcompSuccess after a successful compress(); valid after the stream is synchronised */
  nvcompStatus_t* get_status() const;
  std::shared_ptr<detail::StatusWord> status;
};

struct DecompressionConfig
{
  size_t decomp_data_size = 0;
  uint32_t num_chunks = 0;
  /* uncompressed chunk size the buffer was written with (from its header, or the compressing manager's); a manager
   * decompresses buffers of any chunk size of its format */
  size_t chunk_size = 0;
  /* nvcompSuccess / nvcompErrorBadChecksum / nvcompErrorCannotDecompress; valid after the stream is synchronised */
  nvcompStatus_t* get_status() const;
  std::shared_ptr<detail::StatusWord> status;
};

struct nvcompManagerBase
{
  virtual CompressionConfig configure_compression(const size_t decomp_buffer_size) = 0;
  virtual void compress(const uint8_t* decomp_buffer, uint8_t* comp_buffer, const CompressionConfig& comp_config) = 0;
  /* reads the container header: synchronises the stream (doc/highlevel_cpp_quickstart.md:113-115) */
  virtual DecompressionConfig configure_decompression(const uint8_t* comp_buffer) = 0;
  /* no synchronisation: sizes come from the compression configuration */
  virtual DecompressionConfig configure_decompression(const CompressionConfig& comp_config) = 0;
  virtual void decompress(uint8_t* decomp_buffer, const uint8_t* comp_buffer, const DecompressionConfig& decomp_config) = 0;
  /* synchronises the stream and returns the size the last compress() produced */
  virtual size_t get_compressed_output_size(uint8_t* comp_buffer) = 0;
  virtual ~nvcompManagerBase() = default;
};

/* Common implementation behind every format's manager: chunking, scratch and the
 * container around the batched low-level codec selected by `format`. */
class BatchedManager : public nvcompManagerBase
{
public:
  enum Format : uint32_t { kLZ4 = 1, kSnappy = 2, kCascaded = 3, kBitcomp = 4, kANS = 5, kDeflate = 6 };

  BatchedManager(Format format, size_t uncomp_chunk_size, const void* format_opts, size_t format_opts_bytes,
                 hipStream_t user_stream, int device_id, ChecksumPolicy checksum_policy);
  ~BatchedManager() override;
  BatchedManager(const BatchedManager&) = delete;
  BatchedManager& operator=(const BatchedManager&) = delete;

  CompressionConfig configure_compression(const size_t decomp_buffer_size) override;
  void compress(const uint8_t* decomp_buffer, uint8_t* comp_buffer, const CompressionConfig& comp_config) override;
  DecompressionConfig configure_decompression(const uint8_t* comp_buffer) override;
  DecompressionConfig configure_decompression(const CompressionConfig& comp_config) override;
  void decompress(uint8_t* decomp_buffer, const uint8_t* comp_buffer, const DecompressionConfig& decomp_config) override;
  size_t get_compressed_output_size(uint8_t* comp_buffer) override;

private:
  std::unique_ptr<detail::ManagerImpl> impl_;
};

} // namespace nvcomp
