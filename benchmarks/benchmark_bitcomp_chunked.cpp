/* benchmark_bitcomp_chunked -- low-level Bitcomp round trip over files cut into chunks
 * (reference program: benchmarks/benchmark_bitcomp_chunked.cu: defaults {0, uchar},
 * options -t/--type, -a/--algorithm {0|1}). */
#include "benchmark_template_chunked.hpp"

static nvcompBatchedBitcompFormatOpts g_opts = {0, NVCOMP_TYPE_UCHAR};

static bool handle_extra(const std::string& flag, const std::string& val)
{
  if (flag == "-t" || flag == "--type") {
    static const struct { const char* name; nvcompType_t t; } kTypes[] = {
        {"char", NVCOMP_TYPE_CHAR}, {"uchar", NVCOMP_TYPE_UCHAR}, {"short", NVCOMP_TYPE_SHORT},
        {"ushort", NVCOMP_TYPE_USHORT}, {"int", NVCOMP_TYPE_INT}, {"uint", NVCOMP_TYPE_UINT},
        {"longlong", NVCOMP_TYPE_LONGLONG}, {"ulonglong", NVCOMP_TYPE_ULONGLONG}};
    for (const auto& k : kTypes) {
      if (val == k.name) {
        g_opts.data_type = k.t;
        return true;
      }
    }
    throw std::runtime_error("ERROR: Bitcomp data type must be char, uchar, short, ushort, int, uint, longlong or ulonglong");
  }
  if (flag == "-a" || flag == "--algorithm") {
    const int algo = std::atoi(val.c_str());
    if (algo < 0 || algo > 1) {
      throw std::runtime_error("ERROR: Bitcomp algorithm must be 0 or 1, but it is " + std::to_string(algo));
    }
    g_opts.algorithm_type = algo;
    return true;
  }
  return false;
}

static bool input_valid(const std::vector<std::vector<char>>& chunks)
{
  if ((int)g_opts.data_type < 0 || (int)g_opts.data_type > 7) {
    std::cerr << "ERROR: Bitcomp data type must be 0-7 (CHAR, UCHAR, SHORT, USHORT, INT, UINT, LONGLONG, or ULONGLONG), "
                 "but it is " << (int)g_opts.data_type << std::endl;
    return false;
  }
  const size_t width = (size_t)1 << ((unsigned)g_opts.data_type >> 1);
  for (const auto& c : chunks) {
    if (c.size() % width != 0) {
      std::cerr << "ERROR: Input data must have a length and chunk size that are a multiple of " << width
                << ", the size of the specified data type." << std::endl;
      return false;
    }
  }
  return true;
}

int main(int argc, char** argv)
{
  return bench::main_chunked(
      argc, argv, "  -t, --type T       element type (default uchar)\n  -a, --algorithm A  0 = default (delta), 1 = sparse\n",
      handle_extra, [](size_t) {
        bench::Codec c;
        c.compress_temp_size = [](size_t n, size_t m, size_t* out) { return nvcompBatchedBitcompCompressGetTempSize(n, m, g_opts, out); };
        c.max_output_chunk_size = [](size_t m, size_t* out) { return nvcompBatchedBitcompCompressGetMaxOutputChunkSize(m, g_opts, out); };
        c.compress_async = [](const void* const* ip, const size_t* is, size_t m, size_t n, void* t, size_t tb,
                              void* const* op, size_t* os, hipStream_t s) {
          return nvcompBatchedBitcompCompressAsync(ip, is, m, n, t, tb, op, os, g_opts, s);
        };
        c.decompress_temp_size = nvcompBatchedBitcompDecompressGetTempSize;
        c.decompress_async = nvcompBatchedBitcompDecompressAsync;
        c.input_valid = input_valid;
        return c;
      });
}
