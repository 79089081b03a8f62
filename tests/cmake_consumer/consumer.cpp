// Links the C API through the imported target and calls two host-only entry points (no GPU needed).
#include <cstdio>

#include "nvcomp/lz4.h"

int main()
{
  size_t bound = 0;
  if (nvcompBatchedLZ4CompressGetMaxOutputChunkSize(65536, nvcompBatchedLZ4DefaultOpts, &bound) != nvcompSuccess) {
    return 1;
  }
  std::printf("nvcomp::nvcomp linked, LZ4 bound for 64 KiB = %zu\n", bound);
  return bound == 65809 ? 0 : 2;
}
