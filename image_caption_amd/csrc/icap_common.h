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

// Wave64 reductions on DPP (no LDS traffic, ~6 short VALU steps): an inclusive scan by
// row_shr 1,2,4,8 inside each row of 16 lanes, then row_bcast15 / row_bcast31 carry the
// row totals upward, so lane 63 ends with the reduction of all 64 lanes; v_readlane
// broadcasts it.  Lanes without a source take `identity` (bound_ctrl off).
#define ICAP_DPP_ROW_SHR(n) (0x110 + (n))
#define ICAP_DPP_BCAST15 0x142
#define ICAP_DPP_BCAST31 0x143

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float identity, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(v),
                                                    CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i(int identity, int v) {
  return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROW_MASK, 0xf, false);
}

__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f<ICAP_DPP_ROW_SHR(1), 0xf>(0.f, v);
  v += dpp_f<ICAP_DPP_ROW_SHR(2), 0xf>(0.f, v);
  v += dpp_f<ICAP_DPP_ROW_SHR(4), 0xf>(0.f, v);
  v += dpp_f<ICAP_DPP_ROW_SHR(8), 0xf>(0.f, v);
  v += dpp_f<ICAP_DPP_BCAST15, 0xa>(0.f, v);
  v += dpp_f<ICAP_DPP_BCAST31, 0xc>(0.f, v);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  const float ninf = -INFINITY;
  v = fmaxf(v, dpp_f<ICAP_DPP_ROW_SHR(1), 0xf>(ninf, v));
  v = fmaxf(v, dpp_f<ICAP_DPP_ROW_SHR(2), 0xf>(ninf, v));
  v = fmaxf(v, dpp_f<ICAP_DPP_ROW_SHR(4), 0xf>(ninf, v));
  v = fmaxf(v, dpp_f<ICAP_DPP_ROW_SHR(8), 0xf>(ninf, v));
  v = fmaxf(v, dpp_f<ICAP_DPP_BCAST15, 0xa>(ninf, v));
  v = fmaxf(v, dpp_f<ICAP_DPP_BCAST31, 0xc>(ninf, v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// XCD-aware tile order.  Workgroups b and b+8 share an XCD (and its L2), so the grid
// index is first remapped so that every XCD owns one contiguous run of tile numbers
// (bijective for any tile count), and tile numbers walk down a column of tiles (same
// B panel, successive M tiles) before moving to the next column.
__device__ __forceinline__ void tile_of_block(int pid, int tiles_m, int tiles_n, int& tm,
                                              int& tn) {
  const int total = tiles_m * tiles_n;
  const int q = total >> 3, r = total & 7;
  const int xcd = pid & 7;
  const int sid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (pid >> 3);
  tm = sid % tiles_m;
  tn = sid / tiles_m;
}

// Same XCD remap, but the tile numbers of an XCD sweep a band of ceil(tiles_n/8) column
// panels with the panel index fastest and the M tile slowest.  The band of B panels stays
// in the XCD's L2 while every A tile is fetched once per XCD and reused across the band
// by consecutive workgroups - the order for tall A operands that do not fit in one L2.
// `width` > 0 narrows the bands to that many panels (an XCD then sweeps several bands one
// after the other): for B operands whose eighth does not fit in an L2 either - a band of
// `width` panels does, and the A operand is fetched once per band instead of B once per M tile.
__device__ __forceinline__ void tile_of_block_band(int pid, int tiles_m, int tiles_n, int& tm,
                                                   int& tn, int width_limit = 0) {
  const int total = tiles_m * tiles_n;
  const int q = total >> 3, r = total & 7;
  const int xcd = pid & 7;
  const int sid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (pid >> 3);
  const int pw8 = (tiles_n + 7) >> 3;
  const int pw = width_limit > 0 ? min(width_limit, pw8) : pw8;  // panels per band
  const int band_tiles = tiles_m * pw;
  const int band = sid / band_tiles;
  const int in_band = sid - band * band_tiles;
  const int width = min(pw, tiles_n - band * pw);  // the last band may be narrower
  tm = in_band / width;
  tn = band * pw + in_band % width;
}

// Row-band variant: every XCD owns a band of ceil(tiles_m/8) tile rows and sweeps it with
// the tile row fastest, so the A operand is fetched by one XCD only and B by all eight -
// the order for small products whose A operand was just written by the previous kernel.
__device__ __forceinline__ void tile_of_block_rowband(int pid, int tiles_m, int tiles_n, int& tm,
                                                      int& tn) {
  tile_of_block_band(pid, tiles_n, tiles_m, tn, tm);
}
#endif

}  // namespace icap
