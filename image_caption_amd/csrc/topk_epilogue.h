// Per-row reduction used by the fused "projection + beam step" epilogues: a lane that holds
// NX consecutive logits of one row folds them into the row's record for those columns -
// maximum, sum of exp(x - max) and the TOPK largest values with their columns (equal
// values: lower column first).  Records are 2 + 2*TOPK floats:
//   [max, sum, x_0 .. x_{TOPK-1}, col_0 .. col_{TOPK-1} (int bits, TOPK_EMPTY = none)].
#pragma once

#include "icap_common.h"

namespace icap {

constexpr int TOPK_EMPTY = 0x7fffffff;

// Additive logit masks of the baseline beam searcher (utils/agent_utils.py:188-237), applied
// to the logits of a row before the log-softmax: float32.min on the start token's column at
// every step; with per-row target lengths, float32.min on every column but `eos` at the step
// where step1 == row_len[row], and on every column but `pad` where step1 > row_len[row].
// Disabled (the decode loop of recurrent_example.py) when start_col < 0 and row_len is null.
struct LogitMasks {
  int start_col = -1;
  const int32_t* row_len = nullptr;
  int step1 = 0;  // the searcher's loop index i plus one
  int eos = -1, pad = -1;
  __host__ __device__ bool any() const { return start_col >= 0 || row_len != nullptr; }
};

#if defined(__HIPCC__)
template <int TOPK>
struct TopkRow {
  float m;          // maximum (= x[0] of the list, -inf if no valid column)
  float sum;        // sum exp(x - m)
  float tx[TOPK];   // descending
  int tc[TOPK];
  float eos_v;      // the logit of column `eos` if it is among the scanned columns
  bool has_eos;
};

// x[e] is the logit of column cbase + e of row `row`: adds the masks in the reference's
// order (start, then eos, then pad; two hits overflow to -inf as they do there).
template <int NX>
__device__ __forceinline__ void apply_logit_masks(float (&x)[NX], int cbase, int row, int n_rows,
                                                  const LogitMasks& mk) {
  constexpr float FMIN = -3.4028234663852886e38f;  // tf.float32.min
  int keep = -1;  // the one column that stays unmasked in a forced row
  if (mk.row_len != nullptr && row < n_rows) {
    const int len = mk.row_len[row];
    keep = mk.step1 == len ? mk.eos : (mk.step1 > len ? mk.pad : -1);
  }
#pragma unroll
  for (int e = 0; e < NX; ++e) {
    const int c = cbase + e;
    float v = x[e];
    v = (c == mk.start_col) ? v + FMIN : v;
    v = (keep >= 0 && c != keep) ? v + FMIN : v;
    x[e] = v;
  }
}

// x[e] is the logit of column cbase + e; invalid columns must hold -inf.
// The list stays sorted, so "v beats entry k-1" implies "v beats entry k" and the new entry k
// is the median of (entry k-1, entry k, v): one v_med3_f32 per level for the values, two
// selects per level for the columns.  The eos logit is picked out only in the wave whose
// columns hold it (a wave-uniform branch), not by a compare per element.
template <int TOPK, int NX>
__device__ __forceinline__ TopkRow<TOPK> topk_scan(const float (&x)[NX], int cbase, int eos) {
  TopkRow<TOPK> r;
#pragma unroll
  for (int k = 0; k < TOPK; ++k) { r.tx[k] = -INFINITY; r.tc[k] = TOPK_EMPTY; }
  r.eos_v = 0.f;
  r.has_eos = false;
  {
    const int e_eos = eos - cbase;
    const bool hit = (unsigned)e_eos < (unsigned)NX;
    if (__builtin_amdgcn_ballot_w64(hit) != 0ull) {
      float ev = x[0];
#pragma unroll
      for (int e = 1; e < NX; ++e) ev = (e_eos == e) ? x[e] : ev;
      r.eos_v = ev;
      r.has_eos = hit;
    }
  }
#pragma unroll
  for (int e = 0; e < NX; ++e) {
    const float v = x[e];
    const int c = cbase + e;
    // insert (v, c) into the descending list; strict '>' keeps the lower column on ties
    bool gt[TOPK];
#pragma unroll
    for (int k = 0; k < TOPK; ++k) gt[k] = v > r.tx[k];
#pragma unroll
    for (int k = TOPK - 1; k >= 1; --k) {
      const int below = gt[k] ? c : r.tc[k];
      r.tc[k] = gt[k - 1] ? r.tc[k - 1] : below;
      r.tx[k] = __builtin_amdgcn_fmed3f(r.tx[k - 1], r.tx[k], v);
    }
    r.tc[0] = gt[0] ? c : r.tc[0];
    r.tx[0] = __builtin_fmaxf(r.tx[0], v);
  }
  r.m = r.tx[0];
  const float ms = (r.m > -INFINITY) ? r.m : 0.f;
  float sum = 0.f;
#pragma unroll
  for (int e = 0; e < NX; e += 4)
    sum += (fast_exp(x[e] - ms) + fast_exp(x[e + 1] - ms)) +
           (fast_exp(x[e + 2] - ms) + fast_exp(x[e + 3] - ms));
  r.sum = sum;
  return r;
}

// topk_scan for lanes whose NX <= 64 logits also lie in memory (`exact`: LDS, exact[e] == x[e]):
// the same record with less vector-unit work.  The TOPK + 1 largest are tracked on 32-bit keys -
// the logit with its low six mantissa bits replaced by the element index - so one
// v_med3_f32 per level replaces the compare and the two selects per level for the columns.
// Key order equals logit order wherever the logits differ above those six bits; if two
// neighbouring keys of the final list agree there (or one of them stands for an infinite or
// absent logit), the wave repeats the scan with topk_scan, so the record is the exact one in
// every case.  The values are then read back by element index.  x_differs (uniform over
// the wave): x[] no longer equals exact[] (masks were applied), take topk_scan at once.
template <int TOPK, int NX>
__device__ __forceinline__ TopkRow<TOPK> topk_scan_keyed(const float (&x)[NX], const float* exact,
                                                          int cbase, int eos, bool x_differs = false) {
  static_assert(NX <= 64 && NX > TOPK, "six bits of element index, more elements than list entries");
  constexpr float FMAX = 3.4028234663852886e38f;
  float t[TOPK + 1];
#pragma unroll
  for (int k = 0; k <= TOPK; ++k) t[k] = -INFINITY;
  const float top = INFINITY;
#pragma unroll
  for (int e = 0; e < NX; ++e) {
    // (clamped: an infinite logit with index bits would be a NaN)
    const float c = __builtin_amdgcn_fmed3f(x[e], -FMAX, FMAX);
    const float kf = __uint_as_float((__float_as_uint(c) & 0xffffffc0u) | (unsigned)e);
#pragma unroll
    for (int k = TOPK; k >= 1; --k) t[k] = __builtin_amdgcn_fmed3f(t[k - 1], t[k], kf);
    t[0] = __builtin_amdgcn_fmed3f(t[0], kf, top);  // max, as a median with +inf
  }
  bool redo = x_differs;
#pragma unroll
  for (int k = 0; k < TOPK; ++k)
    redo |= ((__float_as_uint(t[k]) ^ __float_as_uint(t[k + 1])) < 64u);
  // (a clamped or absent logit: the exponent field of its key is all ones but for the lowest bit)
#pragma unroll
  for (int k = 0; k <= TOPK; ++k) redo |= ((__float_as_uint(t[k]) & 0x7f000000u) == 0x7f000000u);
  if (__builtin_amdgcn_ballot_w64(redo) != 0ull) return topk_scan<TOPK, NX>(x, cbase, eos);
  TopkRow<TOPK> r;
  r.eos_v = 0.f;
  r.has_eos = false;
  {
    const int e_eos = eos - cbase;
    const bool hit = (unsigned)e_eos < (unsigned)NX;
    if (__builtin_amdgcn_ballot_w64(hit) != 0ull) {
      r.eos_v = exact[hit ? e_eos : 0];
      r.has_eos = hit;
    }
  }
#pragma unroll
  for (int k = 0; k < TOPK; ++k) {
    const int e = (int)(__float_as_uint(t[k]) & 63u);
    r.tx[k] = exact[e];
    r.tc[k] = cbase + e;
  }
  r.m = r.tx[0];
  float sum = 0.f;
#pragma unroll
  for (int e = 0; e < NX; e += 4)
    sum += (fast_exp(x[e] - r.m) + fast_exp(x[e + 1] - r.m)) +
           (fast_exp(x[e + 2] - r.m) + fast_exp(x[e + 3] - r.m));
  r.sum = sum;
  return r;
}

// Merges the record of the neighbouring lane (lane ^ 1, which holds the HIGHER columns when
// this lane is even) into this one; meaningful on even lanes.
template <int TOPK>
__device__ __forceinline__ TopkRow<TOPK> topk_merge_neighbour(const TopkRow<TOPK>& a) {
  constexpr int SWAP = 0xB1;  // quad_perm [1,0,3,2]
  const float om = dpp_f<SWAP, 0xf>(0.f, a.m);
  const float osum = dpp_f<SWAP, 0xf>(0.f, a.sum);
  float ox[TOPK];
  int oc[TOPK];
#pragma unroll
  for (int k = 0; k < TOPK; ++k) {
    ox[k] = dpp_f<SWAP, 0xf>(0.f, a.tx[k]);
    oc[k] = dpp_i<SWAP, 0xf>(0, a.tc[k]);
  }
  TopkRow<TOPK> r;
  r.eos_v = a.eos_v;
  r.has_eos = a.has_eos;
  r.m = fmaxf(a.m, om);
  const float ms = (r.m > -INFINITY) ? r.m : 0.f;
  r.sum = ((a.m > -INFINITY) ? a.sum * fast_exp(a.m - ms) : 0.f) +
          ((om > -INFINITY) ? osum * fast_exp(om - ms) : 0.f);
  // merge two descending lists (static indices only: each output picks by counting);
  // this lane's columns are the lower ones, so on equal values its entries come first
  int ia = 0, ib = 0;
#pragma unroll
  for (int k = 0; k < TOPK; ++k) {
    float av = -INFINITY, bv = -INFINITY;
    int ac = TOPK_EMPTY, bc = TOPK_EMPTY;
#pragma unroll
    for (int q = 0; q < TOPK; ++q) {
      if (q == ia) { av = a.tx[q]; ac = a.tc[q]; }
      if (q == ib) { bv = ox[q]; bc = oc[q]; }
    }
    const bool take_a = (ac != TOPK_EMPTY) && (bc == TOPK_EMPTY || av >= bv);
    r.tx[k] = take_a ? av : bv;
    r.tc[k] = take_a ? ac : bc;
    ia += take_a ? 1 : 0;
    ib += take_a ? 0 : 1;
  }
  return r;
}

template <int TOPK, bool DEVICE_COHERENT = false>
__device__ __forceinline__ void topk_store(float* dst, const TopkRow<TOPK>& r) {
  constexpr int RL = 2 + 2 * TOPK;
  float out[RL];
  out[0] = r.m;
  out[1] = r.sum;
#pragma unroll
  for (int k = 0; k < TOPK; ++k) {
    out[2 + k] = r.tx[k];
    out[2 + TOPK + k] = __int_as_float(r.tc[k]);
  }
  if constexpr (DEVICE_COHERENT) {
    // written through to where every XCD sees it (no cache maintenance needed by a reader
    // inside the same launch that loads with the same scope)
    unsigned long long* d2 = reinterpret_cast<unsigned long long*>(dst);
#pragma unroll
    for (int q = 0; q < RL / 2; ++q) {
      const unsigned long long v = (unsigned long long)__float_as_uint(out[2 * q]) |
                                   ((unsigned long long)__float_as_uint(out[2 * q + 1]) << 32);
      __hip_atomic_store(d2 + q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else {
    float4* d4 = reinterpret_cast<float4*>(dst);
#pragma unroll
    for (int q = 0; q < RL / 4; ++q)
      d4[q] = make_float4(out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]);
  }
}
#endif

}  // namespace icap
