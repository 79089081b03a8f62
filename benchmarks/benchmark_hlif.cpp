/*
 * benchmark_hlif <format> -f <file> -- one manager, one buffer: warm-up + N timed
 * compress calls, N timed decompress calls, byte-exact check
 * (reference program: benchmarks/benchmark_hlif.cpp + benchmark_hlif.hpp; same flags:
 * -f/--filename, -c/--chunk-size, -g/--gpu, -n/--num-iters, -t/--type, cascaded -r -d -b).
 */
#include <chrono>
#include <iomanip>
#include <memory>

#include "nvcomp.hpp"
#include "../examples/util.hpp"

using namespace nvcomp;

int main(int argc, char** argv)
{
  try {
    if (argc < 2) {
      std::cerr << "Usage: benchmark_hlif {lz4|snappy|cascaded|bitcomp|ans|deflate} -f FILE [-c chunk] [-g gpu] [-n iters] [-t type] [-r -d -b]"
                << std::endl;
      return 1;
    }
    const std::string format = argv[1];
    std::string file;
    size_t chunk = 65536, iters = 1;
    int gpu = 0;
    nvcompType_t type = NVCOMP_TYPE_CHAR;
    nvcompBatchedCascadedOpts_t casc = nvcompBatchedCascadedDefaultOpts;
    ChecksumPolicy policy = NoComputeNoVerify;
    for (int i = 2; i + 1 < argc; i += 2) {
      const std::string flag = argv[i], val = argv[i + 1];
      if (flag == "-f" || flag == "--filename") {
        file = val;
      } else if (flag == "-c" || flag == "--chunk-size") {
        chunk = std::strtoull(val.c_str(), nullptr, 10);
      } else if (flag == "-g" || flag == "--gpu") {
        gpu = std::atoi(val.c_str());
      } else if (flag == "-n" || flag == "--num-iters") {
        iters = std::strtoull(val.c_str(), nullptr, 10);
      } else if (flag == "-t" || flag == "--type") {
        type = val == "short" ? NVCOMP_TYPE_SHORT : val == "int" ? NVCOMP_TYPE_INT : val == "longlong" ? NVCOMP_TYPE_LONGLONG : NVCOMP_TYPE_CHAR;
      } else if (flag == "-r" || flag == "--num-rles") {
        casc.num_RLEs = std::atoi(val.c_str());
      } else if (flag == "-d" || flag == "--num-deltas") {
        casc.num_deltas = std::atoi(val.c_str());
      } else if (flag == "-b" || flag == "--num-bps") {
        casc.use_bp = std::atoi(val.c_str());
      } else if (flag == "--checksum") { /* extension: 0 NoComputeNoVerify (the reference's program), 1 ComputeAndNoVerify,
                                            2 NoComputeAndVerifyIfPresent, 3 ComputeAndVerifyIfPresent, 4 ComputeAndVerify */
        policy = (ChecksumPolicy)std::atoi(val.c_str());
      } else if (flag == "-m" || flag == "--memory") {
        /* accepted for compatibility; scratch is always managed by the manager */
      } else {
        throw std::runtime_error("ERROR: unknown option " + flag);
      }
    }
    if (file.empty()) {
      throw std::runtime_error("ERROR: Must specify a file with -f");
    }
    HIP_CHECK(hipSetDevice(gpu));
    hipStream_t stream;
    HIP_CHECK(hipStreamCreate(&stream));
    std::unique_ptr<nvcompManagerBase> manager;
    if (format == "lz4") {
      manager.reset(new LZ4Manager(chunk, nvcompBatchedLZ4Opts_t{type}, stream, gpu, policy));
    } else if (format == "snappy") {
      manager.reset(new SnappyManager(chunk, nvcompBatchedSnappyDefaultOpts, stream, gpu, policy));
    } else if (format == "cascaded") {
      casc.type = type;
      manager.reset(new CascadedManager(chunk, casc, stream, gpu, policy));
    } else if (format == "bitcomp") {
      manager.reset(new BitcompManager(chunk, nvcompBatchedBitcompFormatOpts{0 /* algo--fixed for now */, type}, stream, gpu,
                                       policy));
    } else if (format == "ans") {
      manager.reset(new ANSManager(chunk, nvcompBatchedANSOpts_t{}, stream, gpu, policy));
    } else if (format == "deflate") {
      manager.reset(new DeflateManager(chunk, nvcompBatchedDeflateDefaultOpts, stream, gpu, policy));
    } else {
      throw std::runtime_error("ERROR: unsupported format \"" + format + "\" (this build: lz4, snappy, cascaded, bitcomp, ans, deflate)");
    }
    const std::vector<char> data = util::read_file(file);
    const size_t n = data.size();
    std::cout << "----------" << std::endl;
    std::cout << "uncompressed (B): " << n << std::endl;
    uint8_t *d_in, *d_comp, *d_out;
    HIP_CHECK(hipMalloc((void**)&d_in, n ? n : 1));
    HIP_CHECK(hipMemcpy(d_in, data.data(), n, hipMemcpyHostToDevice));
    CompressionConfig cc = manager->configure_compression(n);
    HIP_CHECK(hipMalloc((void**)&d_comp, cc.max_compressed_buffer_size));
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    manager->compress(d_in, d_comp, cc); /* warm-up */
    HIP_CHECK(hipStreamSynchronize(stream));
    double comp_ms = 0;
    size_t comp_bytes = 0;
    for (size_t i = 0; i < iters; ++i) {
      HIP_CHECK(hipEventRecord(e0, stream));
      manager->compress(d_in, d_comp, cc);
      HIP_CHECK(hipEventRecord(e1, stream));
      comp_bytes = manager->get_compressed_output_size(d_comp); /* synchronises */
      float ms;
      HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
      comp_ms += ms;
    }
    comp_ms /= (double)iters;
    std::cout << "comp_size: " << comp_bytes << ", compressed ratio: " << std::fixed << std::setprecision(2)
              << (double)n / (double)comp_bytes << std::endl;
    std::cout << "compression throughput (GB/s): " << (double)n / 1.0e9 / (comp_ms * 1.0e-3) << std::endl;
    DecompressionConfig dc = manager->configure_decompression(d_comp);
    HIP_CHECK(hipMalloc((void**)&d_out, dc.decomp_data_size ? dc.decomp_data_size : 1));
    manager->decompress(d_out, d_comp, dc); /* warm-up */
    HIP_CHECK(hipStreamSynchronize(stream));
    double decomp_ms = 0;
    for (size_t i = 0; i < iters; ++i) {
      HIP_CHECK(hipEventRecord(e0, stream));
      manager->decompress(d_out, d_comp, dc);
      HIP_CHECK(hipEventRecord(e1, stream));
      HIP_CHECK(hipStreamSynchronize(stream));
      float ms;
      HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
      decomp_ms += ms;
    }
    decomp_ms /= (double)iters;
    std::cout << "decompression throughput (GB/s): " << (double)n / 1.0e9 / (decomp_ms * 1.0e-3) << std::endl;
    std::cout << "decompression time: " << decomp_ms << " ms." << std::endl;
    if (*dc.get_status() != nvcompSuccess || dc.decomp_data_size != n) {
      throw std::runtime_error("ERROR: decompression reported status " + std::to_string((int)*dc.get_status()));
    }
    std::vector<char> back(n);
    HIP_CHECK(hipMemcpy(back.data(), d_out, n, hipMemcpyDeviceToHost));
    if (back != data) {
      throw std::runtime_error("ERROR: decompressed data does not match the input");
    }
    (void)hipFree(d_in);
    (void)hipFree(d_comp);
    (void)hipFree(d_out);
    manager.reset();
    HIP_CHECK(hipStreamDestroy(stream));
    return 0;
  } catch (const std::exception& e) {
    std::cerr << e.what() << std::endl;
    return 1;
  }
}
