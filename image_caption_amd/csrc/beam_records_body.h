// The beam-search step on the top-k records of the vocab projection, as a device function:
// the stand-alone kernel (beam.hip) and the projection kernel that carries the step in its
// own launch (gemm_f32.hip) share it.
#pragma once

#include <math.h>
#include <stdint.h>

#include "beam_shared.h"
#include "icap_common.h"

namespace icap {
namespace {

constexpr int EMPTY_IDX = 0x7fffffff;

__device__ __forceinline__ bool cand_better(float sa, int ia, float sb, int ib) {
  // a real candidate always beats an empty slot; otherwise higher score, then lower index
  if (ib == EMPTY_IDX) return ia != EMPTY_IDX;
  if (ia == EMPTY_IDX) return false;
  return sa > sb || (sa == sb && ia < ib);
}

// wave-wide argmax of (score, total, idx) in tf.nn.top_k order; every lane gets the result.
// Same DPP ladder as wave_max, carrying the triple; the combine keeps the better of
// (own, incoming), so lane 63 ends with the best of the wave.
#define ICAP_BEST_STEP(CTRL, MASK)                                    \
  {                                                                   \
    const float os = dpp_f<CTRL, MASK>(-INFINITY, bs);                \
    const float ot = dpp_f<CTRL, MASK>(0.f, bt);                      \
    const int oi = dpp_i<CTRL, MASK>(EMPTY_IDX, bi);                  \
    if (cand_better(os, oi, bs, bi)) { bs = os; bt = ot; bi = oi; }   \
  }
__device__ __forceinline__ void wave_best(float& bs, float& bt, int& bi) {
  ICAP_BEST_STEP(ICAP_DPP_ROW_SHR(1), 0xf)
  ICAP_BEST_STEP(ICAP_DPP_ROW_SHR(2), 0xf)
  ICAP_BEST_STEP(ICAP_DPP_ROW_SHR(4), 0xf)
  ICAP_BEST_STEP(ICAP_DPP_ROW_SHR(8), 0xf)
  ICAP_BEST_STEP(ICAP_DPP_BCAST15, 0xa)
  ICAP_BEST_STEP(ICAP_DPP_BCAST31, 0xc)
  bs = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bs), 63));
  bt = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bt), 63));
  bi = __builtin_amdgcn_readlane(bi, 63);
}

// Input: the per-row records written by the projection's top-k epilogue (for every
// 64-column half tile of a row: max, sum exp(x - max), the WL largest logits and their
// columns) and the row's eos logit.  One workgroup per image:
//   A. one wave per beam row folds the records into the row max and log-sum-exp;
//   B. every recorded logit becomes a candidate with the exact arithmetic of the reference
//      ((lp + ((x - max) - lse)) / penalty); finished beams contribute their synthetic
//      candidates; all of them go to LDS;
//   C. W rounds of a workgroup-wide argmax (score descending, lower flat index first)
//      pick the successors; threads 0..W-1 then write the outputs and the next state.
// A row's best W are always among its half tiles' best WL >= W, so nothing is lost.
struct RecArgs {
  const float* rec;    // [B*rpi][HT][2 + 2*WL]
  const float* eos_x;  // [B*rpi]
  int HT;
  int rpi;             // record rows per image: W, or 1 when only beam 0 was projected (time 0)
  const float* log_probs;
  const uint8_t* finished;
  const int32_t* lengths;
  int time, eos;
  float* scores;
  int32_t* ids;
  int32_t* parents;
  float* next_lp;
  uint8_t* next_fin;
  int32_t* next_len;
  int B, W, V;
  Penalty pen;
};

template <int WL, int NT>
struct RecLds {
  float s_m[NT / 64][WL], s_s[NT / 64][WL];
  float s_bs[NT / 64], s_bt[NT / 64];
  int s_bi[NT / 64];
};

// The step of image b by a workgroup of NT threads, each owning up to ITEMS records.
// COH: the records and eos logits were written by other workgroups of the same launch with
// device-scope stores and are loaded with device scope (never from a stale cache line).
template <int WL, int ITEMS, int NT, bool COH = false>
__device__ __forceinline__ void beam_records_body(const RecArgs& p, int b, RecLds<WL, NT>& lds) {
  constexpr int RL = 2 + 2 * WL;
  constexpr int NWV = NT / 64;
  constexpr int MAXW = WL;  // beam width served by this instantiation
  auto& s_m = lds.s_m;
  auto& s_s = lds.s_s;
  auto& s_bs = lds.s_bs;
  auto& s_bt = lds.s_bt;
  auto& s_bi = lds.s_bi;
  const int W = p.W, V = p.V, HT = p.HT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool first = p.time <= 0;
  const int n_rows = first ? 1 : W;  // time 0 scores beam 0 only
  const int n_rec = n_rows * HT;
  const size_t row0 = (size_t)b * W;

  // ---- every load of the workgroup is issued here ----
  // thread t owns records t, t + NT, ... of the image's n_rows * HT (ITEMS of them at most)
  float4 rr[ITEMS][RL / 4];
  int rrow[ITEMS];
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int idx = tid + it * NT;
    const bool ok = idx < n_rec;
    const int cl = ok ? idx : 0;
    rrow[it] = ok ? cl / HT : -1;
    const float4* src = reinterpret_cast<const float4*>(p.rec + ((size_t)b * p.rpi * HT + cl) * RL);
    if constexpr (COH) {
      const unsigned long long* s2 = reinterpret_cast<const unsigned long long*>(src);
#pragma unroll
      for (int q = 0; q < RL / 4; ++q) {
        const unsigned long long lo = __hip_atomic_load(s2 + 2 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long hi = __hip_atomic_load(s2 + 2 * q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        rr[it][q] = make_float4(__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)),
                                __uint_as_float((unsigned)hi), __uint_as_float((unsigned)(hi >> 32)));
      }
    } else {
#pragma unroll
      for (int q = 0; q < RL / 4; ++q) rr[it][q] = src[q];
    }
  }
  // per-row state (n_rows <= MAXW), and this thread's special candidate: thread
  // w*(W+2)+j makes the j-th synthetic candidate of row w if the row is finished, and the
  // eos candidate of a live row if j == 0
  float lp_r[MAXW];
  int len_r[MAXW];
  bool fin_r[MAXW];
#pragma unroll
  for (int w = 0; w < MAXW; ++w) {
    const bool ok = w < n_rows;
    lp_r[w] = p.log_probs[row0 + (ok ? w : 0)];
    len_r[w] = p.lengths[row0 + (ok ? w : 0)];
    fin_r[w] = ok ? p.finished[row0 + w] != 0 : true;
  }
  const int sp_w = tid / (W + 2), sp_j = tid - sp_w * (W + 2);
  const bool sp_on = sp_w < n_rows;
  const float* sp_eos_p = p.eos_x + (size_t)b * p.rpi + (sp_on ? sp_w : 0);
  const float sp_eos_x = COH ? __hip_atomic_load(sp_eos_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                             : *sp_eos_p;

  // ---- row statistics: (max, sum exp) folded per thread, per wave, then per workgroup ----
  float Mw[MAXW], Lw[MAXW];
#pragma unroll
  for (int w = 0; w < MAXW; ++w) {
    float m = -INFINITY;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it)
      if (rrow[it] == w) m = fmaxf(m, rr[it][0].x);
    const float mw = wave_max(m);
    float sm = 0.f;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it)
      if (rrow[it] == w && rr[it][0].x > -INFINITY) sm += rr[it][0].y * fast_exp(rr[it][0].x - mw);
    const float sw = wave_sum(sm);
    if (lane == 0) { s_m[wave][w] = mw; s_s[wave][w] = sw; }
  }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < MAXW; ++w) {
    float M = s_m[0][w];
#pragma unroll
    for (int q = 1; q < NWV; ++q) M = fmaxf(M, s_m[q][w]);
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < NWV; ++q) tot += (s_m[q][w] > -INFINITY) ? s_s[q][w] * fast_exp(s_m[q][w] - M) : 0.f;
    Mw[w] = M;
    Lw[w] = logf(tot);
  }

  // ---- this thread's candidates, exact scores ----
  constexpr int NCAND = ITEMS * WL + 1;
  float cs[NCAND], ct[NCAND];
  int ci[NCAND];
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int w = rrow[it];
    float lp = 0.f, M = 0.f, lse = 0.f;
    int len = 0;
    bool live = false;
#pragma unroll
    for (int q = 0; q < MAXW; ++q)
      if (q == w) { lp = lp_r[q]; len = len_r[q]; live = !fin_r[q]; M = Mw[q]; lse = Lw[q]; }
    const float pen_oth = penalty_of(p.pen, len + 1);
    const int flat0 = first ? 0 : w * V;
    const float* fl = reinterpret_cast<const float*>(&rr[it][0]);
#pragma unroll
    for (int k = 0; k < WL; ++k) {
      const float x = fl[2 + k];
      const int col = __float_as_int(fl[2 + WL + k]);
      const float tot = lp + ((x - M) - lse);
      const float sc = tot / pen_oth;
      // the eos column is scored by its row's special candidate (its own penalty)
      const bool ok = live && col >= 0 && col < V && col != p.eos && sc == sc;
      cs[it * WL + k] = ok ? sc : -INFINITY;
      ct[it * WL + k] = tot;
      ci[it * WL + k] = ok ? flat0 + col : EMPTY_IDX;
    }
  }
  {
    float sc = -INFINITY, tot = 0.f;
    int fi = EMPTY_IDX;
    if (sp_on) {
      float lp = 0.f, M = 0.f, lse = 0.f;
      int len = 0;
      bool fin = false;
#pragma unroll
      for (int q = 0; q < MAXW; ++q)
        if (q == sp_w) { lp = lp_r[q]; len = len_r[q]; fin = fin_r[q]; M = Mw[q]; lse = Lw[q]; }
      const int flat0 = first ? 0 : sp_w * V;
      const float pen = penalty_of(p.pen, len);
      if (fin) {
        // all mass on eos: only the first W+1 columns and the eos column can be selected
        const int ncol = min(V, W + 1);
        int col = -1;
        if (sp_j < ncol) col = sp_j;
        else if (sp_j == ncol && p.eos >= ncol && p.eos < V) col = p.eos;
        if (col >= 0) {
          tot = lp + ((col == p.eos) ? 0.f : kF32Min);
          sc = tot / pen;
          fi = flat0 + col;
        }
      } else if (sp_j == 0 && p.eos >= 0 && p.eos < V) {
        tot = lp + ((sp_eos_x - M) - lse);
        sc = tot / pen;
        fi = (sc == sc) ? flat0 + p.eos : EMPTY_IDX;
      }
    }
    cs[NCAND - 1] = sc;
    ct[NCAND - 1] = tot;
    ci[NCAND - 1] = fi;
  }

  // ---- W rounds of workgroup-wide argmax ----
  // Every thread keeps the best of its remaining candidates; a round compares those (one
  // value per thread), and only the winner's owner retires it and rescans its own list
  // (flat indices are unique: the eos column and finished rows are scored by the special
  // candidates only).
  float lb_s = cs[0], lb_t = ct[0];
  int lb_i = ci[0];
#pragma unroll
  for (int q = 1; q < NCAND; ++q)
    if (cand_better(cs[q], ci[q], lb_s, lb_i)) { lb_s = cs[q]; lb_t = ct[q]; lb_i = ci[q]; }
  float my_s = -INFINITY, my_t = 0.f;
  int my_i = EMPTY_IDX;
  for (int r = 0; r < W; ++r) {
    float bs = lb_s, bt = lb_t;
    int bi = lb_i;
    wave_best(bs, bt, bi);
    if (lane == 0) { s_bs[wave] = bs; s_bt[wave] = bt; s_bi[wave] = bi; }
    __syncthreads();
    bs = s_bs[0]; bt = s_bt[0]; bi = s_bi[0];
#pragma unroll
    for (int q = 1; q < NWV; ++q)
      if (cand_better(s_bs[q], s_bi[q], bs, bi)) { bs = s_bs[q]; bt = s_bt[q]; bi = s_bi[q]; }
    __syncthreads();
    if (lb_i == bi && bi != EMPTY_IDX) {
      lb_s = -INFINITY; lb_t = 0.f; lb_i = EMPTY_IDX;
#pragma unroll
      for (int q = 0; q < NCAND; ++q) {
        if (ci[q] == bi) { cs[q] = -INFINITY; ci[q] = EMPTY_IDX; }
        if (cand_better(cs[q], ci[q], lb_s, lb_i)) { lb_s = cs[q]; lb_t = ct[q]; lb_i = ci[q]; }
      }
    }
    if (tid == r) { my_s = bs; my_t = bt; my_i = bi; }
  }

  // Threads 0..W-1 finish one successor each.  The next-state buffers may be the state
  // inputs: every read of the state happened before the barriers of the rounds above.
  const bool act = tid < W;
  const int flat = (my_i == EMPTY_IDX) ? 0 : my_i;
  const int word = flat % V;
  const int parent = flat / V;
  // the parent's state is already in registers (rows beyond n_rows cannot be parents)
  int pfin = 0, plen = 0;
#pragma unroll
  for (int q = 0; q < MAXW; ++q)
    if (q == parent) { pfin = fin_r[q] ? 1 : 0; plen = len_r[q]; }
  const int nfin = (pfin || word == p.eos) ? 1 : 0;
  const int nlen = plen + ((word != p.eos && !nfin) ? 1 : 0);
  if (act) {
    const size_t o = row0 + tid;
    p.scores[o] = my_s;
    p.ids[o] = word;
    p.parents[o] = parent;
    p.next_lp[o] = my_t;
    p.next_fin[o] = (uint8_t)nfin;
    p.next_len[o] = nlen;
  }
}

}  // namespace
}  // namespace icap
