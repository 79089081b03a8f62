/*
 * benchmark_allgather -- all-gather of a file's chunks across the GPUs of one node, with
 * and without LZ4 compression of the payload (reference program: benchmarks/benchmark_allgather.cpp).
 *
 *   -f, --filename F   input file (bytes)
 *   -g, --gpu G        number of GPUs (>= 2)
 *   -h, --chunks C     number of chunks, multiple of G (default G); chunk k lives on GPU k / (C/G)
 *   -c, --comp {none,lz4}
 *   --oversubscribe    map the G logical GPUs onto the devices present (testing on fewer GPUs)
 *
 * lz4: every GPU compresses its chunks with an LZ4Manager (64 KiB internal chunks), the
 * compressed buffers are copied device-to-device to every other GPU over xGMI
 * (hipMemcpyPeerAsync, one stream per GPU), every GPU decompresses all remote chunks, and
 * every GPU ends up with, and verifies, the whole file. Same process model as the
 * reference (one host thread, hipSetDevice per GPU) with its indexing defects removed
 * (SURVEY.md 3.5): managers are per GPU, the bytes moved are the actual compressed size,
 * any C that is a multiple of G works. Prints the reference's figures:
 *   per-GPU GB/s  = bytes * (G-1)/G / t        system GB/s = bytes * (G-1) / t
 * The last whitespace-separated token of stdout is the system throughput
 * (benchmarks/allgather_runall.py:56,62 parses exactly that).
 */
#include <chrono>
#include <cstring>
#include <iomanip>
#include <memory>

#include "nvcomp.hpp"
#include "../examples/util.hpp"

using namespace nvcomp;

struct Options
{
  std::string file;
  int gpus = 0;
  int chunks = 0;
  std::string comp = "none";
  bool oversubscribe = false;
};

static int device_of(const Options& o, int g)
{
  if (!o.oversubscribe) {
    return g;
  }
  int count = 1;
  HIP_CHECK(hipGetDeviceCount(&count));
  return g % count;
}

int main(int argc, char** argv)
{
  try {
    Options o;
    for (int i = 1; i < argc; ++i) {
      const std::string f = argv[i];
      if (f == "--oversubscribe") {
        o.oversubscribe = true;
        continue;
      }
      if (i + 1 >= argc) {
        throw std::runtime_error("ERROR: missing value for " + f);
      }
      const std::string v = argv[++i];
      if (f == "-f" || f == "--filename") o.file = v;
      else if (f == "-g" || f == "--gpu") o.gpus = std::atoi(v.c_str());
      else if (f == "-h" || f == "--chunks") o.chunks = std::atoi(v.c_str());
      else if (f == "-c" || f == "--comp" || f == "--compression") o.comp = v;
      else throw std::runtime_error("ERROR: unknown option " + f);
    }
    if (o.file.empty() || o.gpus < 2) {
      throw std::runtime_error("ERROR: need -f FILE and -g GPUS with GPUS >= 2");
    }
    if (o.chunks == 0) {
      o.chunks = o.gpus;
    }
    if (o.chunks % o.gpus != 0) {
      throw std::runtime_error("ERROR: the number of chunks must be a multiple of the number of GPUs");
    }
    if (o.comp != "none" && o.comp != "lz4") {
      throw std::runtime_error("ERROR: -c must be none or lz4");
    }
    const int G = o.gpus, C = o.chunks, per_gpu = C / G;
    const std::vector<char> data = util::read_file(o.file);
    const size_t n = data.size();
    const size_t chunk_bytes = 1 + (n - 1) / (size_t)C;
    auto chunk_len = [&](int k) { return std::min(chunk_bytes, n - std::min(n, (size_t)k * chunk_bytes)); };
    /* peer access between every pair that supports it */
    if (!o.oversubscribe) {
      for (int a = 0; a < G; ++a) {
        HIP_CHECK(hipSetDevice(a));
        for (int b = 0; b < G; ++b) {
          int can = 0;
          if (a != b && hipDeviceCanAccessPeer(&can, a, b) == hipSuccess && can) {
            (void)hipDeviceEnablePeerAccess(b, 0);
          }
        }
      }
    }
    std::vector<hipStream_t> streams(G);
    std::vector<uint8_t*> src(C, nullptr);                                 /* owner's copy of chunk k */
    std::vector<std::vector<uint8_t*>> recv(G, std::vector<uint8_t*>(C));  /* recv[g][k]: chunk k as received on g */
    std::vector<std::vector<uint8_t*>> full(G, std::vector<uint8_t*>(C));  /* full[g][k]: chunk k uncompressed on g */
    std::vector<std::unique_ptr<LZ4Manager>> managers(G);
    std::vector<CompressionConfig> ccfg(C);
    std::vector<uint8_t*> comp(C, nullptr);
    std::vector<size_t> comp_bytes(C, 0);
    const bool lz4 = o.comp == "lz4";
    for (int g = 0; g < G; ++g) {
      const int dev = device_of(o, g);
      HIP_CHECK(hipSetDevice(dev));
      HIP_CHECK(hipStreamCreateWithFlags(&streams[g], hipStreamNonBlocking));
      if (lz4) {
        managers[g].reset(new LZ4Manager(1 << 16, nvcompBatchedLZ4Opts_t{NVCOMP_TYPE_CHAR}, streams[g], dev));
      }
    }
    size_t recv_cap = chunk_bytes;
    for (int k = 0; k < C; ++k) {
      const int owner = k / per_gpu;
      HIP_CHECK(hipSetDevice(device_of(o, owner)));
      HIP_CHECK(hipMalloc((void**)&src[k], chunk_bytes));
      if (chunk_len(k)) {
        HIP_CHECK(hipMemcpy(src[k], data.data() + (size_t)k * chunk_bytes, chunk_len(k), hipMemcpyHostToDevice));
      }
      if (lz4) {
        ccfg[k] = managers[owner]->configure_compression(chunk_len(k));
        HIP_CHECK(hipMalloc((void**)&comp[k], ccfg[k].max_compressed_buffer_size));
        recv_cap = std::max(recv_cap, ccfg[k].max_compressed_buffer_size);
      }
    }
    for (int g = 0; g < G; ++g) {
      HIP_CHECK(hipSetDevice(device_of(o, g)));
      for (int k = 0; k < C; ++k) {
        HIP_CHECK(hipMalloc((void**)&recv[g][k], recv_cap));
        HIP_CHECK(hipMalloc((void**)&full[g][k], chunk_bytes));
      }
    }
    auto sync_all = [&] {
      for (int g = 0; g < G; ++g) {
        HIP_CHECK(hipSetDevice(device_of(o, g)));
        HIP_CHECK(hipStreamSynchronize(streams[g]));
      }
    };
    sync_all();
    const auto t0 = std::chrono::steady_clock::now();
    if (lz4) {
      for (int k = 0; k < C; ++k) {
        const int owner = k / per_gpu;
        HIP_CHECK(hipSetDevice(device_of(o, owner)));
        managers[owner]->compress(src[k], comp[k], ccfg[k]);
      }
      for (int k = 0; k < C; ++k) { /* actual sizes: each call waits for its owner's stream */
        const int owner = k / per_gpu;
        HIP_CHECK(hipSetDevice(device_of(o, owner)));
        comp_bytes[k] = managers[owner]->get_compressed_output_size(comp[k]);
      }
    }
    /* all-gather: chunk k goes from its owner to every other GPU; the owner keeps a plain copy */
    for (int k = 0; k < C; ++k) {
      const int owner = k / per_gpu;
      const uint8_t* payload = lz4 ? comp[k] : src[k];
      const size_t bytes = lz4 ? comp_bytes[k] : chunk_len(k);
      for (int g = 0; g < G; ++g) {
        HIP_CHECK(hipSetDevice(device_of(o, g)));
        if (g == owner) {
          HIP_CHECK(hipMemcpyAsync(full[g][k], src[k], chunk_len(k), hipMemcpyDeviceToDevice, streams[g]));
        } else {
          HIP_CHECK(hipMemcpyPeerAsync(lz4 ? recv[g][k] : full[g][k], device_of(o, g), payload, device_of(o, owner),
                                       bytes, streams[g]));
        }
      }
    }
    std::vector<std::vector<DecompressionConfig>> dcfg(G, std::vector<DecompressionConfig>(C));
    if (lz4) {
      for (int g = 0; g < G; ++g) {
        HIP_CHECK(hipSetDevice(device_of(o, g)));
        for (int k = 0; k < C; ++k) {
          if (k / per_gpu == g) {
            continue;
          }
          dcfg[g][k] = managers[g]->configure_decompression(ccfg[k]); /* no synchronisation */
          managers[g]->decompress(full[g][k], recv[g][k], dcfg[g][k]);
        }
      }
    }
    sync_all();
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    /* every GPU must now hold the whole file */
    std::vector<char> back(chunk_bytes);
    for (int g = 0; g < G; ++g) {
      HIP_CHECK(hipSetDevice(device_of(o, g)));
      for (int k = 0; k < C; ++k) {
        if (lz4 && k / per_gpu != g && *dcfg[g][k].get_status() != nvcompSuccess) {
          throw std::runtime_error("ERROR: decompression failed on GPU " + std::to_string(g));
        }
        HIP_CHECK(hipMemcpy(back.data(), full[g][k], chunk_len(k), hipMemcpyDeviceToHost));
        if (std::memcmp(back.data(), data.data() + (size_t)k * chunk_bytes, chunk_len(k)) != 0) {
          throw std::runtime_error("ERROR: GPU " + std::to_string(g) + " holds wrong data for chunk " + std::to_string(k));
        }
      }
    }
    size_t moved = 0;
    for (int k = 0; k < C; ++k) {
      moved += lz4 ? comp_bytes[k] : chunk_len(k);
    }
    std::cout << std::fixed << std::setprecision(4);
    std::cout << "----------" << std::endl;
    std::cout << "GPUs: " << G << ", chunks: " << C << ", compression: " << o.comp << std::endl;
    std::cout << "uncompressed (B): " << n << std::endl;
    if (lz4) {
      std::cout << "Compressed data size (B): " << moved << ", compression ratio: " << (double)n / (double)moved << std::endl;
    }
    std::cout << "Time (s): " << secs << std::endl;
    std::cout << "Per-GPU throughput (GB/s): " << (double)n * (G - 1) / G / 1.0e9 / secs << std::endl;
    std::cout << "Total system throughput (GB/s): " << (double)n * (G - 1) / 1.0e9 / secs << std::endl;
    return 0;
  } catch (const std::exception& e) {
    std::cerr << e.what() << std::endl;
    return 1;
  }
}
