/* benchmark_deflate_chunked -- low-level DEFLATE round trip over files cut into chunks
 * (reference program: benchmarks/benchmark_deflate_chunked.cu: -a/--algorithm 0..2, chunks of at most 65536 bytes). */
#include "benchmark_template_chunked.hpp"

#include "nvcomp/deflate.h"

static nvcompBatchedDeflateOpts_t g_opts = {0};

static bool handle_arg(const std::string& arg, const std::string& val)
{
  if (arg == "--algorithm" || arg == "-a") {
    const int a = atoi(val.c_str());
    if (a < 0 || a > 2) {
      std::cerr << "ERROR: Deflate algorithm must be 0, 1, or 2, but it is " << a << std::endl;
      exit(1);
    }
    g_opts.algo = a;
    return true;
  }
  return false;
}

int main(int argc, char** argv)
{
  return bench::main_chunked(
      argc, argv, "  -a, --algorithm N        compressor effort 0..2 (default 0)\n", handle_arg, [](size_t) {
        bench::Codec c;
        const nvcompBatchedDeflateOpts_t o = g_opts;
        c.compress_temp_size = [o](size_t n, size_t m, size_t* out) { return nvcompBatchedDeflateCompressGetTempSize(n, m, o, out); };
        c.max_output_chunk_size = [o](size_t m, size_t* out) { return nvcompBatchedDeflateCompressGetMaxOutputChunkSize(m, o, out); };
        c.compress_async = [o](const void* const* ip, const size_t* is, size_t m, size_t n, void* t, size_t tb,
                               void* const* op, size_t* os, hipStream_t s) {
          return nvcompBatchedDeflateCompressAsync(ip, is, m, n, t, tb, op, os, o, s);
        };
        c.decompress_temp_size = nvcompBatchedDeflateDecompressGetTempSize;
        c.decompress_async = nvcompBatchedDeflateDecompressAsync;
        c.input_valid = [](const std::vector<std::vector<char>>& data) {
          for (const auto& chunk : data) {
            if (chunk.size() > 65536) {
              std::cerr << "ERROR: Deflate doesn't support chunk sizes larger than 65536 bytes." << std::endl;
              return false;
            }
          }
          return true;
        };
        return c;
      });
}
