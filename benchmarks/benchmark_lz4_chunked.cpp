/* benchmark_lz4_chunked -- low-level LZ4 round trip over files cut into chunks
 * (reference program: benchmarks/benchmark_lz4_chunked.cu; option -t/--type as there). */
#include "benchmark_template_chunked.hpp"

static nvcompBatchedLZ4Opts_t g_opts = nvcompBatchedLZ4DefaultOpts;

static bool handle_extra(const std::string& flag, const std::string& val)
{
  if (flag != "-t" && flag != "--type") {
    return false;
  }
  static const struct { const char* name; nvcompType_t t; } kTypes[] = {
      {"bits", NVCOMP_TYPE_BITS}, {"char", NVCOMP_TYPE_CHAR}, {"uchar", NVCOMP_TYPE_UCHAR}, {"short", NVCOMP_TYPE_SHORT},
      {"ushort", NVCOMP_TYPE_USHORT}, {"int", NVCOMP_TYPE_INT}, {"uint", NVCOMP_TYPE_UINT}};
  for (const auto& k : kTypes) {
    if (val == k.name) {
      g_opts.data_type = k.t;
      return true;
    }
  }
  throw std::runtime_error("ERROR: LZ4 data type must be one of bits, char, uchar, short, ushort, int, uint");
}

static bool input_valid(const std::vector<std::vector<char>>& chunks)
{
  size_t width = 1;
  switch (g_opts.data_type) {
  case NVCOMP_TYPE_SHORT: case NVCOMP_TYPE_USHORT: width = 2; break;
  case NVCOMP_TYPE_INT: case NVCOMP_TYPE_UINT: width = 4; break;
  default: break;
  }
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
      argc, argv, "  -t, --type {bits,char,uchar,short,ushort,int,uint}  LZ4 element type hint (default char)\n",
      handle_extra, [](size_t) {
        bench::Codec c;
        c.compress_temp_size = [](size_t n, size_t m, size_t* out) { return nvcompBatchedLZ4CompressGetTempSize(n, m, g_opts, out); };
        c.max_output_chunk_size = [](size_t m, size_t* out) { return nvcompBatchedLZ4CompressGetMaxOutputChunkSize(m, g_opts, out); };
        c.compress_async = [](const void* const* ip, const size_t* is, size_t m, size_t n, void* t, size_t tb,
                              void* const* op, size_t* os, hipStream_t s) {
          return nvcompBatchedLZ4CompressAsync(ip, is, m, n, t, tb, op, os, g_opts, s);
        };
        c.decompress_temp_size = nvcompBatchedLZ4DecompressGetTempSize;
        c.decompress_async = nvcompBatchedLZ4DecompressAsync;
        c.input_valid = input_valid;
        return c;
      });
}
