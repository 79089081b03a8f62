/* benchmark_snappy_chunked -- low-level Snappy round trip over files cut into chunks
 * (reference program: benchmarks/benchmark_snappy_chunked.cu; no format options). */
#include "benchmark_template_chunked.hpp"

int main(int argc, char** argv)
{
  return bench::main_chunked(
      argc, argv, "", [](const std::string&, const std::string&) { return false; }, [](size_t) {
        bench::Codec c;
        const nvcompBatchedSnappyOpts_t o = nvcompBatchedSnappyDefaultOpts;
        c.compress_temp_size = [o](size_t n, size_t m, size_t* out) { return nvcompBatchedSnappyCompressGetTempSize(n, m, o, out); };
        c.max_output_chunk_size = [o](size_t m, size_t* out) { return nvcompBatchedSnappyCompressGetMaxOutputChunkSize(m, o, out); };
        c.compress_async = [o](const void* const* ip, const size_t* is, size_t m, size_t n, void* t, size_t tb,
                               void* const* op, size_t* os, hipStream_t s) {
          return nvcompBatchedSnappyCompressAsync(ip, is, m, n, t, tb, op, os, o, s);
        };
        c.decompress_temp_size = nvcompBatchedSnappyDecompressGetTempSize;
        c.decompress_async = nvcompBatchedSnappyDecompressAsync;
        c.input_valid = [](const std::vector<std::vector<char>>&) { return true; };
        return c;
      });
}
