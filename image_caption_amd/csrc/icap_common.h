// Shared host/device helpers for the icap kernels (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "icap.h"

namespace icap {

// Per-thread error message behind icap_last_error().
std::string& last_error();
int fail(int code, const char* fmt, ...);

#define ICAP_REQUIRE(cond, ...)                                   \
  do {                                                            \
    if (!(cond)) return ::icap::fail(ICAP_ERR_INVALID, __VA_ARGS__); \
  } while (0)

#define ICAP_HIP(expr)                                                            \
  do {                                                                            \
    hipError_t e_ = (expr);                                                       \
    if (e_ != hipSuccess)                                                         \
      return ::icap::fail(ICAP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

// Launch check: a bad configuration is reported by hipGetLastError right after the launch.
#define ICAP_LAUNCH_CHECK(name)                                                        \
  do {                                                                                 \
    hipError_t e_ = hipGetLastError();                                                 \
    if (e_ != hipSuccess)                                                              \
      return ::icap::fail(ICAP_ERR_HIP, "launch of %s failed: %s", name, hipGetErrorString(e_)); \
  } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr float kF32Min = -3.4028234663852886e38f;  // tf.float32.min

#if defined(__HIPCC__)
// exp2-based transcendental helpers.  v_exp_f32 / v_rcp_f32 are accurate to about
// one ulp, which keeps tanh/sigmoid within ~2e-7 absolute of the libm values.
__device__ __forceinline__ float fast_exp(float x) {
  return __builtin_amdgcn_exp2f(x * 1.44269504088896341f);
}
__device__ __forceinline__ float fast_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + fast_exp(-x));
}
__device__ __forceinline__ float fast_tanh(float x) {
  // tanh(x) = 1 - 2 / (exp(2x) + 1); exp overflow -> 1, underflow -> -1.
  float e = __builtin_amdgcn_exp2f(x * 2.88539008177792681f);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
#endif

}  // namespace icap
