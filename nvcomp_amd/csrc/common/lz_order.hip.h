/*
 * common/lz_order.hip.h -- in which ORDER the persistent waves of the batched LZ decoders take the chunks of a batch.
 *
 * Chunks of one batch take very different times (the mix: text 0.8 ms, incompressible 0.2 ms, runs 0.1 ms per chunk and
 * wave), and a launch of persistent waves ends when its slowest wave does: with the chunks handed out in the caller's order
 * a batch of 16 384 chunks (2.3 per wave) loses a quarter of its time to the last round -- 492 GB/s against 676 at 65 536
 * chunks (profiles/r04_final_nsweep.jsonl). Handing out the expensive chunks FIRST (longest processing time first) leaves
 * the cheap ones for the end of the launch.
 *
 * MEASURED (round 5, MI355X, gpurun r5j): +3 % at 16 384 chunks, nothing at 32 768 and 65 536, -6 % at 8 192 and on batches
 * of uniform chunks (the two kernels in front cost ~0.1 ms) -- the last round of a launch is not where medium batches lose
 * their time (the first is: 7 168 waves start in the same phase). NVCOMP_LZ_ORDERED is therefore 0 in the shipped build; the
 * order itself stays available for inspection (nvcompAmdBatched<Fmt>DecompressOrderAsync, tests/test_chunk_order.py).
 *
 * Two small kernels in front of the decoder, one thread per chunk:
 *   1. cost: the chunk's sequence count is estimated from the first tokens of its stream (a walk over at most 24 tokens
 *      or 384 bytes, the first sequence -- a chunk opens with a long literal run -- left out, scaled by the compressed size) -- what a chunk costs the one-wave decoder is its number of
 *      sequences, not its bytes; the estimate goes into one of 16 classes (powers of two), the classes are counted;
 *   2. order: a chunk's place = the chunks of the more expensive classes + a ticket of its own class.
 * The order inside a class is whatever the atomics give: it decides which wave decodes a chunk, never what is written.
 * Temp buffer (words): [0] the decoder's ticket counter, [4, 20) class counts, [20, 36) class tickets, from byte 256 the
 * order (u32 per chunk), then the classes (u8 per chunk).
 */
#pragma once

#include "nvcomp/shared_types.h"

#include "common/lz_launch.hip.h"

namespace lzo {

constexpr uint32_t kClasses = 16;
constexpr uint32_t kHeaderBytes = 256;
constexpr uint32_t kSampleBytes = 384;
constexpr uint32_t kSampleTokens = 24;

/* Bytes of temp storage for a batch of n chunks (ticket counter included). */
inline size_t temp_bytes(size_t n)
{
  return kHeaderBytes + ((5 * n + 15) & ~(size_t)15);
}

/* Estimated number of sequences of an LZ4 block from its first tokens. */
struct Lz4Cost
{
  static __device__ __forceinline__ uint32_t estimate(const uint8_t* in, uint32_t len)
  {
    const uint32_t limit = len < kSampleBytes ? len : kSampleBytes;
    uint32_t pos = 0, tokens = 0, first_end = 0;
    while (pos < limit && tokens < kSampleTokens) {
      first_end = tokens == 1 ? pos : first_end;
      const uint32_t t = in[pos++];
      uint32_t ll = t >> 4;
      if (ll == 15) {
        uint32_t b = 255;
        while (b == 255 && pos < limit) {
          b = in[pos++];
          ll += b;
        }
      }
      pos += ll + 2;
      if ((t & 15u) == 15) {
        uint32_t b = 255;
        while (b == 255 && pos < limit) {
          b = in[pos++];
        }
      }
      ++tokens;
    }
    /* the first sequence of a chunk has nothing to refer to: its literal run says little about the rest */
    return tokens >= 2 ? scale(tokens - 1, pos - first_end, len - first_end) : scale(tokens, pos, len);
  }
  static __device__ __forceinline__ uint32_t scale(uint32_t tokens, uint32_t pos, uint32_t len)
  {
    /* tokens seen in `pos` stream bytes (the last one may reach far beyond the sample: a long literal run) -> the whole
     * stream's; plus a small share for the bytes themselves, so that a chunk stored as one literal run is not free. (The
     * start of a stream has fewer matches than its body -- the history is short --, so that the count comes out about
     * three times too low for text; the share of the bytes is set with that in mind: an incompressible 64 KiB chunk takes
     * the decoder a quarter of the time of a chunk of text and must rank below it.) */
    const uint64_t seqs = pos ? (uint64_t)tokens * len / pos : 0;
    return (uint32_t)(seqs < 0x7fffffffu ? seqs : 0x7fffffffu) + (len >> 8);
  }
};

/* The same for a Snappy raw stream: elements (a literal element and the copy behind it cost like one LZ4 sequence). */
struct SnappyCost
{
  static __device__ __forceinline__ uint32_t estimate(const uint8_t* in, uint32_t len)
  {
    const uint32_t limit = len < kSampleBytes ? len : kSampleBytes;
    uint32_t pos = 0, copies = 0, elements = 0;
    while (pos < limit && (in[pos++] & 128u)) { /* the preamble: varint of the uncompressed length */
    }
    uint32_t first = pos;
    while (pos < limit && elements < 2 * kSampleTokens) {
      if (elements == 1 && copies == 0) { /* the opening literal element says little about the rest */
        first = pos;
      }
      const uint32_t tag = in[pos++];
      const uint32_t kind = tag & 3u;
      if (kind == 0) {
        uint32_t n = tag >> 2;
        if (n >= 60) {
          const uint32_t nb = n - 59;
          n = 0;
          for (uint32_t i = 0; i < nb && pos < limit; ++i) {
            n |= (uint32_t)in[pos++] << (8 * i);
          }
        }
        pos += n + 1;
      } else {
        pos += kind == 1 ? 1u : kind == 2 ? 2u : 4u;
        ++copies;
      }
      ++elements;
    }
    const uint32_t seen = pos > first ? pos - first : 0u;
    return Lz4Cost::scale(copies ? copies : (elements ? 1u : 0u), seen, len > first ? len - first : len);
  }
};

__device__ __forceinline__ uint32_t cost_class(uint32_t cost)
{
  const uint32_t c = 32u - (uint32_t)__builtin_clz(cost | 1u); /* 1 ... 32 */
  return c > kClasses ? kClasses - 1 : c - 1;
}

/* Counters of one address are the slow part of such kernels (16 384 atomics on one word: a third of a millisecond, measured):
 * both kernels count inside the workgroup first, in LDS, and touch the shared words once per class and workgroup. */
template <class Cost>
__global__ void __launch_bounds__(256) cost_kernel(lzl::Batch b, uint32_t* temp)
{
  __shared__ uint32_t local[kClasses];
  if (threadIdx.x < kClasses) {
    local[threadIdx.x] = 0;
  }
  __syncthreads();
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < b.batch_size) {
    const uint8_t* in = (const uint8_t*)b.comp_ptrs[i];
    const size_t len64 = b.comp_bytes[i];
    const uint32_t len = len64 < 0x7fffffffu ? (uint32_t)len64 : 0u; /* (a stream the decoder refuses: no cost) */
    const uint32_t c = len && in != nullptr ? cost_class(Cost::estimate(in, len)) : 0u;
    uint8_t* classes = (uint8_t*)temp + kHeaderBytes + 4 * b.batch_size;
    classes[i] = (uint8_t)c;
    atomicAdd(local + c, 1u);
  }
  __syncthreads();
  if (threadIdx.x < kClasses && local[threadIdx.x] != 0) {
    atomicAdd(temp + 4 + threadIdx.x, local[threadIdx.x]);
  }
}

template <class Cost> /* (a kernel per format: the header is part of both translation units) */
__global__ void __launch_bounds__(256) order_kernel(size_t n, uint32_t* temp)
{
  __shared__ uint32_t local[kClasses]; /* the workgroup's chunks per class, then where its share of the class begins */
  if (threadIdx.x < kClasses) {
    local[threadIdx.x] = 0;
  }
  __syncthreads();
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const uint8_t* classes = (const uint8_t*)temp + kHeaderBytes + 4 * n;
  uint32_t c = 0, mine = 0;
  if (i < n) {
    c = classes[i];
    mine = atomicAdd(local + c, 1u);
  }
  __syncthreads();
  if (threadIdx.x < kClasses) {
    const uint32_t k = threadIdx.x;
    uint32_t before = 0; /* the chunks of the more expensive classes */
    for (uint32_t j = k + 1; j < kClasses; ++j) {
      before += temp[4 + j];
    }
    const uint32_t count = local[k];
    local[k] = before + (count ? atomicAdd(temp + 4 + kClasses + k, count) : 0u);
  }
  __syncthreads();
  if (i < n) {
    uint32_t* order = (uint32_t*)((uint8_t*)temp + kHeaderBytes);
    order[local[c] + mine] = (uint32_t)i;
  }
}

/* The cheap variant (NVCOMP_LZ_ORDERED == 2): no look at the streams, one kernel. A chunk whose compressed size is (nearly)
 * its capacity is one literal run, a chunk that shrank eight times or more is a few long matches: both cost the decoder a
 * fraction of what text does. Those are handed out LAST (from the end of the order downwards), everything else first in
 * roughly the caller's order: the launch then drains through short chunks, and the first round keeps its mix. */
template <class Cost>
__global__ void __launch_bounds__(256) order_by_sizes_kernel(lzl::Batch b, uint32_t* temp)
{
  __shared__ uint32_t local[2], base[2];
  if (threadIdx.x < 2) {
    local[threadIdx.x] = 0;
  }
  __syncthreads();
  const size_t n = b.batch_size;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t cheap = 0, mine = 0;
  if (i < n) {
    const size_t comp = b.comp_bytes[i], cap = b.out_caps != nullptr ? b.out_caps[i] : 0;
    cheap = (comp == 0 || comp + (cap >> 6) >= cap || comp * 8 <= cap) ? 1u : 0u;
    mine = atomicAdd(local + cheap, 1u);
  }
  __syncthreads();
  if (threadIdx.x < 2 && local[threadIdx.x] != 0) {
    base[threadIdx.x] = atomicAdd(temp + 4 + threadIdx.x, local[threadIdx.x]);
  }
  __syncthreads();
  if (i < n) {
    uint32_t* order = (uint32_t*)((uint8_t*)temp + kHeaderBytes);
    const uint32_t at = base[cheap] + mine;
    order[cheap ? (uint32_t)n - 1 - at : at] = (uint32_t)i;
  }
}

template <class Cost>
inline const uint32_t* make_order_by_sizes(const lzl::Batch& b, void* temp, size_t bytes, hipStream_t stream)
{
  if (temp == nullptr || ((uintptr_t)temp & 15u) != 0 || bytes < temp_bytes(b.batch_size) || b.batch_size > 0x7fffffffu) {
    return nullptr;
  }
  if (hipMemsetAsync(temp, 0, kHeaderBytes, stream) != hipSuccess) {
    return nullptr;
  }
  const unsigned groups = (unsigned)((b.batch_size + 255) / 256);
  hipLaunchKernelGGL((order_by_sizes_kernel<Cost>), dim3(groups), dim3(256), 0, stream, b, (uint32_t*)temp);
  return (const uint32_t*)((const uint8_t*)temp + kHeaderBytes);
}

/* Clears the header, runs the two kernels on `stream`; returns the order array (device) or nullptr when the temp buffer
 * does not hold it (the decoder then hands the chunks out in the caller's order). The header's first word is the decoder's
 * ticket counter: cleared here too. */
template <class Cost>
inline const uint32_t* make_order(const lzl::Batch& b, void* temp, size_t bytes, hipStream_t stream)
{
  if (temp == nullptr || ((uintptr_t)temp & 15u) != 0 || bytes < temp_bytes(b.batch_size) || b.batch_size > 0x7fffffffu) {
    return nullptr;
  }
  if (hipMemsetAsync(temp, 0, kHeaderBytes, stream) != hipSuccess) {
    return nullptr;
  }
  const unsigned groups = (unsigned)((b.batch_size + 255) / 256);
  hipLaunchKernelGGL((cost_kernel<Cost>), dim3(groups), dim3(256), 0, stream, b, (uint32_t*)temp);
  hipLaunchKernelGGL((order_kernel<Cost>), dim3(groups), dim3(256), 0, stream, b.batch_size, (uint32_t*)temp);
  return (const uint32_t*)((const uint8_t*)temp + kHeaderBytes);
}

/* nvcompAmdBatched<Fmt>DecompressOrderAsync (include/nvcomp/amd_ext.h). */
template <class Cost>
inline nvcompStatus_t order_for_inspection(
    const void* const* comp_ptrs, const size_t* comp_bytes, size_t n, void* temp, size_t bytes, unsigned* out_order,
    unsigned char* out_class, hipStream_t stream)
{
  if (n == 0) {
    return nvcompSuccess;
  }
  if (comp_ptrs == nullptr || comp_bytes == nullptr || out_order == nullptr || out_class == nullptr) {
    return nvcompErrorInvalidValue;
  }
  const lzl::Batch b = {comp_ptrs, comp_bytes, nullptr, nullptr, n, nullptr, nullptr};
  const uint32_t* order = make_order<Cost>(b, temp, bytes, stream);
  if (order == nullptr) {
    return nvcompErrorInvalidValue; /* temp buffer missing, misaligned or smaller than temp_bytes(n) */
  }
  if (hipMemcpyAsync(out_order, order, 4 * n, hipMemcpyDeviceToDevice, stream) != hipSuccess
      || hipMemcpyAsync(out_class, (const uint8_t*)order + 4 * n, n, hipMemcpyDeviceToDevice, stream) != hipSuccess) {
    return nvcompErrorCudaError;
  }
  return nvcompSuccess;
}

} // namespace lzo
