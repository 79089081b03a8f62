/*
 * examples/high_level_quickstart_example.cpp -- the high-level (manager) interface
 * on 1,000,000 pseudo-random bytes: compress with an LZ4Manager, rebuild a manager
 * from the compressed buffer alone, decompress, and do the same with checksums and
 * with several buffers in flight on one stream. Covers the scenarios of the
 * reference's example of the same name (examples/high_level_quickstart_example.cpp:66-381)
 * and verifies every result byte for byte. Exit code 0 = all scenarios passed.
 */
#include <cstring>
#include <random>

#include "nvcomp.hpp"
#include "util.hpp"

using namespace nvcomp;

namespace {

struct DeviceBuf
{
  uint8_t* p = nullptr;
  explicit DeviceBuf(size_t n) { HIP_CHECK(hipMalloc((void**)&p, n ? n : 1)); }
  ~DeviceBuf() { (void)hipFree(p); }
  DeviceBuf(const DeviceBuf&) = delete;
};

void expect_equal(const uint8_t* device, const std::vector<uint8_t>& host, const char* what)
{
  std::vector<uint8_t> back(host.size());
  HIP_CHECK(hipMemcpy(back.data(), device, host.size(), hipMemcpyDeviceToHost));
  if (back != host) {
    throw std::runtime_error(std::string(what) + ": decompressed bytes differ from the input");
  }
}

/* 1) compress, 2) build a new manager from the compressed buffer, 3) decompress with it */
void with_manager_factory(const uint8_t* d_in, const std::vector<uint8_t>& host, ChecksumPolicy comp_policy,
                          ChecksumPolicy decomp_policy)
{
  hipStream_t stream;
  HIP_CHECK(hipStreamCreate(&stream));
  {
    const size_t chunk_size = 1 << 16;
    nvcompBatchedLZ4Opts_t opts{NVCOMP_TYPE_CHAR};
    LZ4Manager manager{chunk_size, opts, stream, 0, comp_policy};
    CompressionConfig cc = manager.configure_compression(host.size());
    DeviceBuf comp(cc.max_compressed_buffer_size);
    manager.compress(d_in, comp.p, cc);
    auto other = create_manager(comp.p, stream, 0, decomp_policy); /* synchronises the stream */
    DecompressionConfig dc = other->configure_decompression(comp.p);
    if (dc.decomp_data_size != host.size()) {
      throw std::runtime_error("factory: wrong decompressed size in the header");
    }
    DeviceBuf out(dc.decomp_data_size);
    other->decompress(out.p, comp.p, dc);
    HIP_CHECK(hipStreamSynchronize(stream));
    if (*dc.get_status() != nvcompSuccess) {
      throw std::runtime_error("factory: status " + std::to_string((int)*dc.get_status()));
    }
    expect_equal(out.p, host, "factory");
    if (manager.get_compressed_output_size(comp.p) > cc.max_compressed_buffer_size) {
      throw std::runtime_error("factory: compressed size exceeds the configured maximum");
    }
  }
  HIP_CHECK(hipStreamDestroy(stream));
}

/* one manager for both directions; corrupt the buffer to see the checksum verdict */
void single_manager_with_checksums(const uint8_t* d_in, const std::vector<uint8_t>& host)
{
  hipStream_t stream;
  HIP_CHECK(hipStreamCreate(&stream));
  {
    nvcompBatchedLZ4Opts_t opts{NVCOMP_TYPE_CHAR};
    LZ4Manager manager{1 << 16, opts, stream, 0, ComputeAndVerify};
    CompressionConfig cc = manager.configure_compression(host.size());
    DeviceBuf comp(cc.max_compressed_buffer_size);
    manager.compress(d_in, comp.p, cc);
    DecompressionConfig dc = manager.configure_decompression(comp.p);
    DeviceBuf out(dc.decomp_data_size);
    manager.decompress(out.p, comp.p, dc);
    HIP_CHECK(hipStreamSynchronize(stream));
    if (*dc.get_status() != nvcompSuccess) {
      throw std::runtime_error("checksums: clean buffer reported status " + std::to_string((int)*dc.get_status()));
    }
    expect_equal(out.p, host, "checksums");
    /* flip one payload byte: the compressed-chunk CRC must catch it */
    const size_t total = manager.get_compressed_output_size(comp.p);
    uint8_t b;
    HIP_CHECK(hipMemcpy(&b, comp.p + total - 9, 1, hipMemcpyDeviceToHost));
    b ^= 0x20;
    HIP_CHECK(hipMemcpy(comp.p + total - 9, &b, 1, hipMemcpyHostToDevice));
    DecompressionConfig dc2 = manager.configure_decompression(comp.p);
    manager.decompress(out.p, comp.p, dc2);
    HIP_CHECK(hipStreamSynchronize(stream));
    if (*dc2.get_status() == nvcompSuccess) {
      throw std::runtime_error("checksums: corruption went unnoticed");
    }
  }
  HIP_CHECK(hipStreamDestroy(stream));
}

/* several buffers through one manager on one stream, configs kept in vectors */
template <class Manager, class Opts>
void multi_buffer_streamed(const uint8_t* d_in, const std::vector<uint8_t>& host, const Opts& opts, size_t chunk)
{
  hipStream_t stream;
  HIP_CHECK(hipStreamCreate(&stream));
  {
    const size_t parts = 10;
    const size_t part = (host.size() / parts) & ~(size_t)7;
    Manager manager{chunk, opts, stream};
    std::vector<CompressionConfig> ccs;
    std::vector<std::unique_ptr<DeviceBuf>> comps, outs;
    for (size_t i = 0; i < parts; ++i) {
      ccs.push_back(manager.configure_compression(part));
      comps.emplace_back(new DeviceBuf(ccs.back().max_compressed_buffer_size));
      manager.compress(d_in + i * part, comps.back()->p, ccs.back());
    }
    std::vector<DecompressionConfig> dcs;
    for (size_t i = 0; i < parts; ++i) {
      dcs.push_back(manager.configure_decompression(ccs[i])); /* no synchronisation */
      outs.emplace_back(new DeviceBuf(part));
      manager.decompress(outs.back()->p, comps[i]->p, dcs.back());
    }
    HIP_CHECK(hipStreamSynchronize(stream));
    for (size_t i = 0; i < parts; ++i) {
      if (*dcs[i].get_status() != nvcompSuccess) {
        throw std::runtime_error("streamed: part " + std::to_string(i) + " failed");
      }
      std::vector<uint8_t> expect(host.begin() + (std::ptrdiff_t)(i * part), host.begin() + (std::ptrdiff_t)((i + 1) * part));
      expect_equal(outs[i]->p, expect, "streamed");
    }
  }
  HIP_CHECK(hipStreamDestroy(stream));
}

} // namespace

/* A manager decompresses what a manager of another chunk size and other format options wrote (the header carries the
 * chunk size, the decoders are driven by the stream alone), refuses a
 * buffer whose header does not add up, and a CascadedManager refuses a buffer that is not a whole number of elements
 * instead of dropping its tail. */
void header_is_authoritative(const uint8_t* d_in, const std::vector<uint8_t>& host)
{
  hipStream_t stream;
  HIP_CHECK(hipStreamCreate(&stream));
  {
    nvcompBatchedLZ4Opts_t opts{NVCOMP_TYPE_CHAR};
    /* the writer declares 4-byte elements, the reader below does not: the options are the compressor's business */
    LZ4Manager writer{1 << 15, nvcompBatchedLZ4Opts_t{NVCOMP_TYPE_INT}, stream, 0, NoComputeNoVerify};
    CompressionConfig cc = writer.configure_compression(host.size());
    DeviceBuf comp(cc.max_compressed_buffer_size);
    writer.compress(d_in, comp.p, cc);
    LZ4Manager reader{1 << 16, opts, stream, 0, NoComputeNoVerify};
    DecompressionConfig dc = reader.configure_decompression(comp.p);
    if (dc.chunk_size != (1u << 15) || dc.num_chunks != cc.num_chunks) {
      throw std::runtime_error("header: the reader did not take the writer's chunk size");
    }
    DeviceBuf out(dc.decomp_data_size + 4096);
    HIP_CHECK(hipMemset(out.p + dc.decomp_data_size, 0xa5, 4096));
    reader.decompress(out.p, comp.p, dc);
    HIP_CHECK(hipStreamSynchronize(stream));
    if (*dc.get_status() != nvcompSuccess) {
      throw std::runtime_error("header: status " + std::to_string((int)*dc.get_status()));
    }
    expect_equal(out.p, host, "header");
    std::vector<uint8_t> fence(4096);
    HIP_CHECK(hipMemcpy(fence.data(), out.p + dc.decomp_data_size, 4096, hipMemcpyDeviceToHost));
    for (uint8_t b : fence) {
      if (b != 0xa5) {
        throw std::runtime_error("header: wrote past the output buffer");
      }
    }
    /* a header that claims more chunks than its size allows is rejected before any kernel runs */
    uint32_t bogus = (uint32_t)cc.num_chunks + 7;
    HIP_CHECK(hipMemcpy(comp.p + 20, &bogus, 4, hipMemcpyHostToDevice)); /* Header::num_chunks */
    bool threw = false;
    try {
      (void)reader.configure_decompression(comp.p);
    } catch (const std::exception&) {
      threw = true;
    }
    if (!threw) {
      throw std::runtime_error("header: an inconsistent header was accepted");
    }
    threw = false;
    try {
      CascadedManager casc{1 << 16, nvcompBatchedCascadedDefaultOpts, stream, 0, NoComputeNoVerify};
      (void)casc.configure_compression(1000003); /* int elements: not a multiple of 4 */
    } catch (const std::invalid_argument&) {
      threw = true;
    }
    if (!threw) {
      throw std::runtime_error("header: CascadedManager accepted a ragged buffer");
    }
  }
  HIP_CHECK(hipStreamDestroy(stream));
}

int main()
{
  try {
    const size_t n = 1000000;
    std::vector<uint8_t> data(n);
    std::mt19937 gen(42);
    std::uniform_int_distribution<short> dist(0, 255);
    for (size_t i = 0; i < n; ++i) {
      /* half noise, half repetitive, so that the codecs have something to find */
      data[i] = (i / 4096) % 2 ? (uint8_t)dist(gen) : (uint8_t)((i * 7) % 251 & 0xf0);
    }
    DeviceBuf d_in(n);
    HIP_CHECK(hipMemcpy(d_in.p, data.data(), n, hipMemcpyHostToDevice));
    with_manager_factory(d_in.p, data, NoComputeNoVerify, NoComputeNoVerify);
    with_manager_factory(d_in.p, data, ComputeAndNoVerify, NoComputeAndVerifyIfPresent);
    with_manager_factory(d_in.p, data, NoComputeNoVerify, ComputeAndVerifyIfPresent);
    single_manager_with_checksums(d_in.p, data);
    header_is_authoritative(d_in.p, data);
    multi_buffer_streamed<LZ4Manager>(d_in.p, data, nvcompBatchedLZ4Opts_t{NVCOMP_TYPE_CHAR}, 1 << 16);
    multi_buffer_streamed<SnappyManager>(d_in.p, data, nvcompBatchedSnappyDefaultOpts, 1 << 15);
    multi_buffer_streamed<DeflateManager>(d_in.p, data, nvcompBatchedDeflateDefaultOpts, 1 << 16);
    multi_buffer_streamed<CascadedManager>(d_in.p, data, nvcompBatchedCascadedDefaultOpts, 1 << 16);
    multi_buffer_streamed<BitcompManager>(d_in.p, data, nvcompBatchedBitcompFormatOpts{0, NVCOMP_TYPE_INT}, 1 << 16);
    multi_buffer_streamed<ANSManager>(d_in.p, data, nvcompBatchedANSOpts_t{}, 1 << 16);
    std::cout << "high_level_quickstart_example: all scenarios passed" << std::endl;
    return 0;
  } catch (const std::exception& e) {
    std::cerr << e.what() << std::endl;
    return 1;
  }
}
