/*
 * benchmark_snappy_synth -- batch of 64 KiB chunks of low-entropy bytes through the
 * low-level Snappy API: W warm-up + N timed back-to-back launches of compress, then of
 * decompress, every byte verified (reference program: benchmarks/benchmark_snappy_synth.cpp;
 * flags -g, -b/--batch_size (4000), -w/--warmup_count (10), -i/--iterations_count (10),
 * -m/--max_byte (3)). As there, the decompress call passes the SAME device array as the
 * buffer-capacity input and the actual-size output.
 */
#include <chrono>
#include <iomanip>

#include "benchmark_common.hpp"
#include "nvcomp/snappy.h"
#include "../examples/util.hpp"

int main(int argc, char** argv)
{
  try {
    int gpu = 0, max_byte = 3;
    size_t batch = 4000, warmup = 10, iters = 10;
    const size_t chunk = 1 << 16;
    for (int i = 1; i + 1 < argc; i += 2) {
      const std::string f = argv[i], v = argv[i + 1];
      if (f == "-g" || f == "--gpu") gpu = std::atoi(v.c_str());
      else if (f == "-b" || f == "--batch_size") batch = std::strtoull(v.c_str(), nullptr, 10);
      else if (f == "-w" || f == "--warmup_count") warmup = std::strtoull(v.c_str(), nullptr, 10);
      else if (f == "-i" || f == "--iterations_count") iters = std::strtoull(v.c_str(), nullptr, 10);
      else if (f == "-m" || f == "--max_byte") max_byte = std::atoi(v.c_str());
      else throw std::runtime_error("ERROR: unknown option " + f);
    }
    HIP_CHECK(hipSetDevice(gpu));
    hipStream_t stream;
    HIP_CHECK(hipStreamCreate(&stream));
    std::mt19937 rng(0);
    const std::vector<uint8_t> host = bench::gen_data(max_byte, batch * chunk, rng);
    uint8_t *d_in, *d_comp, *d_out;
    size_t max_out = 0, ctemp = 0, dtemp = 0;
    if (nvcompBatchedSnappyCompressGetMaxOutputChunkSize(chunk, nvcompBatchedSnappyDefaultOpts, &max_out) != nvcompSuccess
        || nvcompBatchedSnappyCompressGetTempSize(batch, chunk, nvcompBatchedSnappyDefaultOpts, &ctemp) != nvcompSuccess
        || nvcompBatchedSnappyDecompressGetTempSize(batch, chunk, &dtemp) != nvcompSuccess) {
      throw std::runtime_error("ERROR: size query failed");
    }
    HIP_CHECK(hipMalloc((void**)&d_in, batch * chunk));
    HIP_CHECK(hipMalloc((void**)&d_comp, batch * max_out));
    HIP_CHECK(hipMalloc((void**)&d_out, batch * chunk));
    HIP_CHECK(hipMemcpy(d_in, host.data(), batch * chunk, hipMemcpyHostToDevice));
    std::vector<void*> in_p(batch), comp_p(batch), out_p(batch);
    std::vector<size_t> in_s(batch, chunk);
    for (size_t i = 0; i < batch; ++i) {
      in_p[i] = d_in + i * chunk;
      comp_p[i] = d_comp + i * max_out;
      out_p[i] = d_out + i * chunk;
    }
    void **d_in_p, **d_comp_p, **d_out_p, *d_ctemp, *d_dtemp;
    size_t *d_in_s, *d_comp_s, *d_out_s;
    nvcompStatus_t* d_status;
    HIP_CHECK(hipMalloc((void**)&d_in_p, batch * 8));
    HIP_CHECK(hipMalloc((void**)&d_comp_p, batch * 8));
    HIP_CHECK(hipMalloc((void**)&d_out_p, batch * 8));
    HIP_CHECK(hipMalloc((void**)&d_in_s, batch * 8));
    HIP_CHECK(hipMalloc((void**)&d_comp_s, batch * 8));
    HIP_CHECK(hipMalloc((void**)&d_out_s, batch * 8));
    HIP_CHECK(hipMalloc((void**)&d_status, batch * sizeof(nvcompStatus_t)));
    HIP_CHECK(hipMalloc(&d_ctemp, ctemp ? ctemp : 1));
    HIP_CHECK(hipMalloc(&d_dtemp, dtemp ? dtemp : 1));
    HIP_CHECK(hipMemcpy(d_in_p, in_p.data(), batch * 8, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_comp_p, comp_p.data(), batch * 8, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_out_p, out_p.data(), batch * 8, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_in_s, in_s.data(), batch * 8, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_out_s, in_s.data(), batch * 8, hipMemcpyHostToDevice));
    auto compress = [&] {
      if (nvcompBatchedSnappyCompressAsync(d_in_p, d_in_s, chunk, batch, d_ctemp, ctemp, d_comp_p, d_comp_s,
                                           nvcompBatchedSnappyDefaultOpts, stream) != nvcompSuccess) {
        throw std::runtime_error("ERROR: nvcompBatchedSnappyCompressAsync failed");
      }
    };
    auto decompress = [&] {
      /* capacity array == actual-size array, as in the reference program */
      if (nvcompBatchedSnappyDecompressAsync(d_comp_p, d_comp_s, d_out_s, d_out_s, batch, d_dtemp, dtemp, d_out_p, d_status,
                                             stream) != nvcompSuccess) {
        throw std::runtime_error("ERROR: nvcompBatchedSnappyDecompressAsync failed");
      }
    };
    auto timed = [&](const std::function<void()>& fn) {
      for (size_t i = 0; i < warmup; ++i) fn();
      HIP_CHECK(hipStreamSynchronize(stream));
      const auto t0 = std::chrono::steady_clock::now();
      for (size_t i = 0; i < iters; ++i) fn();
      HIP_CHECK(hipStreamSynchronize(stream));
      return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / (double)iters;
    };
    const double tc = timed(compress);
    std::vector<size_t> comp_s(batch);
    HIP_CHECK(hipMemcpy(comp_s.data(), d_comp_s, batch * 8, hipMemcpyDeviceToHost));
    size_t comp_total = 0;
    for (size_t s : comp_s) comp_total += s;
    const double td = timed(decompress);
    std::vector<nvcompStatus_t> st(batch);
    std::vector<uint8_t> back(batch * chunk);
    HIP_CHECK(hipMemcpy(st.data(), d_status, batch * sizeof(nvcompStatus_t), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(back.data(), d_out, batch * chunk, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < batch; ++i) {
      if (st[i] != nvcompSuccess) throw std::runtime_error("ERROR: chunk " + std::to_string(i) + " failed to decompress");
    }
    if (back != host) throw std::runtime_error("ERROR: decompressed data differs from the input");
    const double bytes = (double)(batch * chunk);
    std::cout << std::fixed << std::setprecision(4);
    std::cout << "batch_size: " << batch << ", chunk (B): " << chunk << ", max_byte: " << max_byte << std::endl;
    std::cout << "compressed ratio: " << bytes / (double)comp_total << std::endl;
    std::cout << "compression throughput (GB/s): " << bytes / 1.0e9 / tc << std::endl;
    std::cout << "decompression throughput (GB/s): " << bytes / 1.0e9 / td << std::endl;
    return 0;
  } catch (const std::exception& e) {
    std::cerr << e.what() << std::endl;
    return 1;
  }
}
