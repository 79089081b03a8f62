/* benchmark_cascaded_chunked -- low-level Cascaded round trip over files cut into chunks
 * (reference program: benchmarks/benchmark_cascaded_chunked.cu: defaults {4096, uint, 2, 1, 1},
 * options -t/--type, -r/--num_rles, -d/--num_deltas, -b/--num_bps). */
#include "benchmark_template_chunked.hpp"

static nvcompBatchedCascadedOpts_t g_opts = {4096, NVCOMP_TYPE_UINT, 2, 1, 1};

static bool handle_extra(const std::string& flag, const std::string& val)
{
  if (flag == "-t" || flag == "--type") {
    static const struct { const char* name; nvcompType_t t; } kTypes[] = {
        {"char", NVCOMP_TYPE_CHAR}, {"uchar", NVCOMP_TYPE_UCHAR}, {"short", NVCOMP_TYPE_SHORT},
        {"ushort", NVCOMP_TYPE_USHORT}, {"int", NVCOMP_TYPE_INT}, {"uint", NVCOMP_TYPE_UINT},
        {"longlong", NVCOMP_TYPE_LONGLONG}, {"ulonglong", NVCOMP_TYPE_ULONGLONG}};
    for (const auto& k : kTypes) {
      if (val == k.name) {
        g_opts.type = k.t;
        return true;
      }
    }
    throw std::runtime_error("ERROR: Cascaded data type must be char, uchar, short, ushort, int, uint, longlong or ulonglong");
  }
  if (flag == "-r" || flag == "--num_rles") {
    g_opts.num_RLEs = std::atoi(val.c_str());
    return true;
  }
  if (flag == "-d" || flag == "--num_deltas") {
    g_opts.num_deltas = std::atoi(val.c_str());
    return true;
  }
  if (flag == "-b" || flag == "--num_bps") {
    g_opts.use_bp = std::atoi(val.c_str());
    if (g_opts.use_bp != 0 && g_opts.use_bp != 1) {
      throw std::runtime_error("ERROR: num_bps must be 0 or 1");
    }
    return true;
  }
  return false;
}

static bool input_valid(const std::vector<std::vector<char>>& chunks)
{
  if ((int)g_opts.type < 0 || (int)g_opts.type > 7) {
    std::cerr << "ERROR: Cascaded data type must be 0-7" << std::endl;
    return false;
  }
  const size_t width = (size_t)1 << ((unsigned)g_opts.type >> 1);
  for (const auto& c : chunks) {
    if (c.size() % width != 0) {
      std::cerr << "ERROR: every chunk must be a multiple of the element size (" << width << " B)" << std::endl;
      return false;
    }
  }
  return true;
}

int main(int argc, char** argv)
{
  return bench::main_chunked(
      argc, argv,
      "  -t, --type T       element type (default uint)\n  -r, --num_rles N   RLE layers (default 2)\n"
      "  -d, --num_deltas N delta layers (default 1)\n  -b, --num_bps 0|1  bit-packing (default 1)\n",
      handle_extra, [](size_t) {
        bench::Codec c;
        c.compress_temp_size = [](size_t n, size_t m, size_t* out) { return nvcompBatchedCascadedCompressGetTempSize(n, m, g_opts, out); };
        c.max_output_chunk_size = [](size_t m, size_t* out) { return nvcompBatchedCascadedCompressGetMaxOutputChunkSize(m, g_opts, out); };
        c.compress_async = [](const void* const* ip, const size_t* is, size_t m, size_t n, void* t, size_t tb,
                              void* const* op, size_t* os, hipStream_t s) {
          return nvcompBatchedCascadedCompressAsync(ip, is, m, n, t, tb, op, os, g_opts, s);
        };
        c.decompress_temp_size = nvcompBatchedCascadedDecompressGetTempSize;
        c.decompress_async = nvcompBatchedCascadedDecompressAsync;
        c.input_valid = input_valid;
        return c;
      });
}
