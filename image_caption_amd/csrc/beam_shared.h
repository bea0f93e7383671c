// Length-penalty table shared by the beam-search kernels.
#pragma once

#include <math.h>

#include "icap_common.h"

namespace icap {

constexpr int PEN_N = 64;  // host-computed entries (len 0..63); longer lengths use powf on device

// (5 + len)^w in f32 divided by the double 6^w folded to f32 (beam_search.py:127-128).
// The table comes from the host libm so that small lengths match numpy's float32 pow.
struct Penalty {
  float lpw;
  float den;
  float table[PEN_N];
};

inline Penalty make_penalty(double length_penalty_weight) {
  Penalty p;
  p.lpw = (float)length_penalty_weight;
  p.den = (float)pow(6.0, length_penalty_weight);
  for (int l = 0; l < PEN_N; ++l) p.table[l] = powf(5.0f + (float)l, p.lpw) / p.den;
  return p;
}

#if defined(__HIPCC__)
__device__ __forceinline__ float penalty_of(const Penalty& p, int len) {
  if (len >= 0 && len < PEN_N) return p.table[len];
  return powf(5.0f + (float)len, p.lpw) / p.den;
}
#endif

}  // namespace icap
