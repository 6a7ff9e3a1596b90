# CMake package file of the MI355X ICP engine (in-tree: no install step needed).
#
#   list(APPEND CMAKE_PREFIX_PATH "<this repository>/cmake")     # or -Dmi355_icp_DIR=<this repository>/cmake
#   find_package(mi355_icp REQUIRED)
#   target_link_libraries(opencv_surface_match mi355_icp::mi355_icp)
#
# gives the reference's `opencv_surface_match` target (pose_estimation/CMakeLists.txt:53-58) the C-ABI header
# (include/mi355_icp.h), the C++ mirror / OpenCV adapter headers (pose_estimation_amd/host/) and the shared library
# (pose_estimation_amd/lib/libmi355_icp.so, built by `make -C pose_estimation_amd/csrc` or __graft_entry__.build()).
get_filename_component(_mi355_icp_root "${CMAKE_CURRENT_LIST_DIR}/.." ABSOLUTE)
set(mi355_icp_INCLUDE_DIRS "${_mi355_icp_root}/include" "${_mi355_icp_root}")
set(mi355_icp_LIBRARY "${_mi355_icp_root}/pose_estimation_amd/lib/libmi355_icp.so")
if(NOT EXISTS "${mi355_icp_LIBRARY}")
  set(mi355_icp_FOUND FALSE)
  set(mi355_icp_NOT_FOUND_MESSAGE "libmi355_icp.so has not been built: run `make -C ${_mi355_icp_root}/pose_estimation_amd/csrc`")
  return()
endif()
if(NOT TARGET mi355_icp::mi355_icp)
  add_library(mi355_icp::mi355_icp SHARED IMPORTED)
  set_target_properties(mi355_icp::mi355_icp PROPERTIES
    IMPORTED_LOCATION "${mi355_icp_LIBRARY}"
    IMPORTED_NO_SONAME TRUE
    INTERFACE_INCLUDE_DIRECTORIES "${mi355_icp_INCLUDE_DIRS}"
    INTERFACE_COMPILE_FEATURES cxx_std_17)
endif()
set(mi355_icp_LIBRARIES mi355_icp::mi355_icp)
set(mi355_icp_FOUND TRUE)
