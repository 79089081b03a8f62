/* nvcomp.hpp -- umbrella header of the C++ interface (reference include sites:
 * benchmarks/benchmark_allgather.cpp:33-34, benchmarks/benchmark_hlif.cpp:34-35). */
#pragma once

#include "nvcomp.h"
#include "nvcomp/ans.hpp"
#include "nvcomp/bitcomp.hpp"
#include "nvcomp/cascaded.hpp"
#include "nvcomp/deflate.hpp"
#include "nvcomp/lz4.hpp"
#include "nvcomp/nvcompManager.hpp"
#include "nvcomp/nvcompManagerFactory.hpp"
#include "nvcomp/snappy.hpp"

namespace nvcomp {

/* Element-type tag of a C++ type; callers add specialisations
 * (benchmarks/benchmark_common.h:136-140 maps float to NVCOMP_TYPE_INT). */
template <typename T>
inline nvcompType_t TypeOf();
template <> inline nvcompType_t TypeOf<int8_t>() { return NVCOMP_TYPE_CHAR; }
template <> inline nvcompType_t TypeOf<uint8_t>() { return NVCOMP_TYPE_UCHAR; }
template <> inline nvcompType_t TypeOf<int16_t>() { return NVCOMP_TYPE_SHORT; }
template <> inline nvcompType_t TypeOf<uint16_t>() { return NVCOMP_TYPE_USHORT; }
template <> inline nvcompType_t TypeOf<int32_t>() { return NVCOMP_TYPE_INT; }
template <> inline nvcompType_t TypeOf<uint32_t>() { return NVCOMP_TYPE_UINT; }
template <> inline nvcompType_t TypeOf<int64_t>() { return NVCOMP_TYPE_LONGLONG; }
template <> inline nvcompType_t TypeOf<uint64_t>() { return NVCOMP_TYPE_ULONGLONG; }

} // namespace nvcomp
