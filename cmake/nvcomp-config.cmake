#[=======================================================================[
nvcomp-config.cmake -- CMake package of the MI355X-native build, so that a caller written against the reference's
package (CMakeLists.txt:18 `find_package(nvcomp 3.0.3 REQUIRED)`, cmake/nvcomp-config.cmake.in:25-26: target
`nvcomp::nvcomp`, variable `NVCOMP_FOUND`) configures unchanged:

    cmake -Dnvcomp_DIR=<this repo>/cmake ...        # or CMAKE_PREFIX_PATH=<this repo>
    find_package(nvcomp 3.0.3 REQUIRED)
    target_link_libraries(app PRIVATE nvcomp::nvcomp)

The target carries the include directories (include/ and ROCm's HIP headers), the HIP platform define the HIP
headers need under a plain host compiler, and links the HIP runtime. The library itself is built in-tree by
nvcomp_amd/csrc/Makefile (`python -c "import __graft_entry__ as g; g.build()"`); nothing is installed.
#]=======================================================================]

get_filename_component(_nvcomp_root "${CMAKE_CURRENT_LIST_DIR}/.." ABSOLUTE)
set(_nvcomp_lib "${_nvcomp_root}/nvcomp_amd/lib/libnvcomp.so")
if(NOT EXISTS "${_nvcomp_lib}")
  set(${CMAKE_FIND_PACKAGE_NAME}_FOUND FALSE)
  set(${CMAKE_FIND_PACKAGE_NAME}_NOT_FOUND_MESSAGE
      "libnvcomp.so has not been built: run `make -C ${_nvcomp_root}/nvcomp_amd/csrc` first")
  return()
endif()

if(NOT DEFINED ROCM_PATH)
  if(DEFINED ENV{ROCM_PATH})
    set(ROCM_PATH "$ENV{ROCM_PATH}")
  else()
    set(ROCM_PATH "/opt/rocm")
  endif()
endif()
find_library(NVCOMP_HIP_RUNTIME amdhip64 HINTS "${ROCM_PATH}/lib" REQUIRED)

if(NOT TARGET nvcomp::nvcomp)
  add_library(nvcomp::nvcomp SHARED IMPORTED)
  set_target_properties(nvcomp::nvcomp PROPERTIES
    IMPORTED_LOCATION "${_nvcomp_lib}"
    IMPORTED_NO_SONAME TRUE
    INTERFACE_INCLUDE_DIRECTORIES "${_nvcomp_root}/include;${ROCM_PATH}/include"
    INTERFACE_COMPILE_DEFINITIONS "__HIP_PLATFORM_AMD__"
    INTERFACE_LINK_LIBRARIES "${NVCOMP_HIP_RUNTIME}")
endif()

set(NVCOMP_FOUND TRUE)
set(${CMAKE_FIND_PACKAGE_NAME}_CONFIG "${CMAKE_CURRENT_LIST_FILE}")
include(FindPackageHandleStandardArgs)
find_package_handle_standard_args(${CMAKE_FIND_PACKAGE_NAME} CONFIG_MODE)
unset(_nvcomp_root)
unset(_nvcomp_lib)
