/*
 * examples/standard_crc_checksum.cpp -- the managers' per-chunk checksums are the STANDARD CRC-32
 * (IEEE 802.3, reflected, initial value and final xor 0xffffffff): what boost::crc_32_type and zlib's crc32()
 * compute. The reference pins exactly this in its example of the same name
 * (examples/standard_crc_checksum.cpp:94-107: device CRC of every chunk == boost::crc_32_type of the same bytes);
 * boost is not in this image, zlib is, and both implement the same function.
 *
 * For several managers and buffer sizes (a ragged last chunk, a single chunk, many chunks) the buffer is compressed
 * with ComputeAndNoVerify, the container is copied to the host, and the CRC of every uncompressed chunk and of every
 * compressed chunk stored in it is compared with zlib's. Exit code 0 = all equal.
 *
 * Container layout (DESIGN.md "HLIF container"): 64-byte header | u64 comp_size[N] | u64 comp_offset[N+1]
 * | u32 crc_uncomp[N] | u32 crc_comp[N] | chunks.
 */
#include <zlib.h>

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

size_t round8(size_t v) { return (v + 7) & ~(size_t)7; }

template <class Manager, class Opts>
size_t check(const char* name, const std::vector<uint8_t>& host, size_t chunk_size, Opts opts)
{
  hipStream_t stream;
  HIP_CHECK(hipStreamCreate(&stream));
  size_t chunks = 0;
  {
    DeviceBuf in(host.size());
    HIP_CHECK(hipMemcpy(in.p, host.data(), host.size(), hipMemcpyHostToDevice));
    Manager manager{chunk_size, opts, stream, 0, ComputeAndNoVerify};
    CompressionConfig cc = manager.configure_compression(host.size());
    DeviceBuf comp(cc.max_compressed_buffer_size);
    manager.compress(in.p, comp.p, cc);
    HIP_CHECK(hipStreamSynchronize(stream));
    const size_t total = manager.get_compressed_output_size(comp.p);
    std::vector<uint8_t> c(total);
    HIP_CHECK(hipMemcpy(c.data(), comp.p, total, hipMemcpyDeviceToHost));
    const size_t n = cc.num_chunks;
    const size_t tables = round8(8 * n + 8 * (n + 1) + 8 * n);
    const uint64_t* sizes = reinterpret_cast<const uint64_t*>(c.data() + 64);
    const uint64_t* offsets = sizes + n;
    const uint32_t* crc_uncomp = reinterpret_cast<const uint32_t*>(c.data() + 64 + 8 * n + 8 * (n + 1));
    const uint32_t* crc_comp = crc_uncomp + n;
    for (size_t i = 0; i < n; ++i) {
      const size_t lo = i * chunk_size;
      const size_t len = host.size() - lo < chunk_size ? host.size() - lo : chunk_size;
      const uint32_t want_u = (uint32_t)crc32(0L, host.data() + lo, (uInt)len);
      const uint32_t want_c = (uint32_t)crc32(0L, c.data() + 64 + tables + offsets[i], (uInt)sizes[i]);
      if (crc_uncomp[i] != want_u || crc_comp[i] != want_c) {
        throw std::runtime_error(std::string(name) + ": chunk " + std::to_string(i) + " CRC differs from zlib's crc32()");
      }
    }
    chunks = n;
  }
  HIP_CHECK(hipStreamDestroy(stream));
  return chunks;
}

} // namespace

int main()
{
  try {
    std::mt19937 gen(12); /* the reference's seed */
    size_t chunks = 0;
    for (const size_t bytes : {(size_t)1024 * 1024 + 333, (size_t)1000, (size_t)65536, (size_t)5 * 65536 + 1}) {
      std::vector<uint8_t> host(bytes);
      for (size_t i = 0; i < bytes; ++i) {
        /* compressible and not: a CRC must not care */
        host[i] = (i / 4096) % 2 ? (uint8_t)gen() : (uint8_t)("checksum"[i % 8]);
      }
      chunks += check<LZ4Manager>("LZ4Manager", host, 1 << 16, nvcompBatchedLZ4Opts_t{NVCOMP_TYPE_CHAR});
      chunks += check<SnappyManager>("SnappyManager", host, 1 << 12, nvcompBatchedSnappyOpts_t{0});
      chunks += check<ANSManager>("ANSManager", host, 1 << 15, nvcompBatchedANSOpts_t{nvcomp_rANS});
    }
    std::cout << chunks << " chunks: CRC-32 of every uncompressed and compressed chunk equals zlib's crc32()" << std::endl;
  } catch (const std::exception& e) {
    std::cerr << "FAILED: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
