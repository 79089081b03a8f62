/* benchmark_ans_chunked -- low-level ANS round trip over files cut into chunks
 * (reference program: benchmarks/benchmark_ans_chunked.cu; no format options, chunks must
 * stay below 2^32 bytes). */
#include "benchmark_template_chunked.hpp"

int main(int argc, char** argv)
{
  return bench::main_chunked(
      argc, argv, "", [](const std::string&, const std::string&) { return false; }, [](size_t) {
        bench::Codec c;
        const nvcompBatchedANSOpts_t o = nvcompBatchedANSDefaultOpts;
        c.compress_temp_size = [o](size_t n, size_t m, size_t* out) { return nvcompBatchedANSCompressGetTempSize(n, m, o, out); };
        c.max_output_chunk_size = [o](size_t m, size_t* out) { return nvcompBatchedANSCompressGetMaxOutputChunkSize(m, o, out); };
        c.compress_async = [o](const void* const* ip, const size_t* is, size_t m, size_t n, void* t, size_t tb,
                               void* const* op, size_t* os, hipStream_t s) {
          return nvcompBatchedANSCompressAsync(ip, is, m, n, t, tb, op, os, o, s);
        };
        c.decompress_temp_size = nvcompBatchedANSDecompressGetTempSize;
        c.decompress_async = nvcompBatchedANSDecompressAsync;
        c.input_valid = [](const std::vector<std::vector<char>>& chunks) {
          for (const auto& ch : chunks) {
            if (ch.size() > (1ull << 32) - 1) {
              std::cerr << "ERROR: ANS doesn't support chunk sizes larger than 2^32-1 bytes." << std::endl;
              return false;
            }
          }
          return true;
        };
        return c;
      });
}
