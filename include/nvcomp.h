/*
 * nvcomp.h -- umbrella header of the C API (MI355X build).
 * Version macros mirror the package version the reference links against
 * (CMakeLists.txt:18 `find_package(nvcomp 3.0.3 REQUIRED)`).
 */
#ifndef NVCOMP_H
#define NVCOMP_H

#define NVCOMP_MAJOR_VERSION 3
#define NVCOMP_MINOR_VERSION 0
#define NVCOMP_PATCH_VERSION 3

#include "nvcomp/shared_types.h"
#include "nvcomp/lz4.h"
#include "nvcomp/snappy.h"
#include "nvcomp/cascaded.h"
#include "nvcomp/bitcomp.h"
#include "nvcomp/ans.h"
#include "nvcomp/deflate.h"
#include "nvcomp/gzip.h"

#endif /* NVCOMP_H */
