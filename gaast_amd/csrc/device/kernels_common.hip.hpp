// kernels_common.hip.hpp -- includes and vector types shared by the kernel headers
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

namespace gaast {
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef double double2v __attribute__((ext_vector_type(2)));
typedef float float16v __attribute__((ext_vector_type(16)));
}  // namespace gaast
