// Fused beam-search step (log-softmax + finished mask + length penalty + top-W over
// W*V candidates + gathers) and back-tracking.
//
// beam_step_f32 is two launches:
//
//   beam_row_kernel    one workgroup per (image, beam row).  The row's V logits are
//       read from HBM ONCE into registers (ITEMS float4 per thread, every load issued
//       before the first use).  One barrier gives the row max, the log-sum-exp and a
//       lower bound on the row's WL-th best score (the WL-th largest lane maximum), so
//       only a handful of "contenders" need the exact arithmetic of the reference
//       ((lp + ((x - max) - lse)) / penalty with an IEEE division); wave 0 ranks them with
//       DPP argmax rounds and writes the row's best WL candidates to a small workspace.
//       The global top-W is a subset of the union of the per-row top-W, so nothing is
//       lost.  Details at the kernel.
//   beam_merge_kernel  one wavefront per image: merges the W rows' candidates (and the
//       synthetic candidates of finished beams, which do not depend on the logits),
//       extracts the W winners in tf.nn.top_k order (score descending, lower flat
//       index first) and writes the step outputs and the next beam state.
//
// Rows too long for the register budget (V > 65536) take the same kernel with the
// logits streamed twice.
#include <math.h>

#include <mutex>

#include <stdlib.h>

#include "beam_records_body.h"
#include "beam_shared.h"
#include "icap_common.h"

namespace icap {
namespace {

constexpr int WL_MAX = 16;  // candidate slots per row in the workspace

struct BeamArgs {
  const float* logits;
  const float* log_probs;
  const uint8_t* finished;
  const int32_t* lengths;
  int time, start_offset, eos, input_is_log_prob;
  float* scores;
  int32_t* ids;
  int32_t* parents;
  float* next_lp;
  uint8_t* next_fin;
  int32_t* next_len;
  int B, W, V;
  int R;  // rows (beams) per image in the inputs; W successors per image come out
  // workspace: per (image,row) statistics, then candidate slots
  float2* ws_stats;  // [B*R] (max, lse)
  float* ws_score;   // [B*R*WL_MAX]
  float* ws_total;   // [B*R*WL_MAX]
  int* ws_idx;       // [B*R*WL_MAX]
  Penalty pen;
  // Device-side switch (engine only; both null otherwise): the launch does nothing unless
  // *run_flag != 0, and the merge kernel zeroes *clear_flag (a word no workgroup of this
  // launch tests) either way.
  const int* run_flag = nullptr;
  int* clear_flag = nullptr;
};


// The K largest of the values held one-per-slot by the 64 lanes (PERV slots per lane),
// largest first, as wave-uniform results; slots are consumed (set to -inf).  Values that
// are NaN never count.  Returns the K-th largest (-inf if fewer than K exist).
template <int K, int PERV>
__device__ __forceinline__ float wave_kth_largest(float (&v)[PERV], int lane, float* out_topk) {
  float kth = -INFINITY;
#pragma unroll
  for (int r = 0; r < K; ++r) {
    float m = v[0];
#pragma unroll
    for (int q = 1; q < PERV; ++q) m = fmaxf(m, v[q]);
    const float top = wave_max(m);
    kth = top;
    if (out_topk) out_topk[r] = top;
    // remove one instance of `top` from its first owner
    const unsigned long long owners = __ballot(m == top);
    if (owners != 0 && lane == (int)__builtin_ctzll(owners)) {
      bool done = false;
#pragma unroll
      for (int q = 0; q < PERV; ++q)
        if (!done && v[q] == top) { v[q] = -INFINITY; done = true; }
    }
  }
  return kth;
}

// Column of element e of item i for thread tid.  VEC: a thread owns 4 consecutive
// columns per item (one 16-byte load).  !VEC (V % 4 != 0 or unaligned rows): element
// (i, e) is its own coalesced 4-byte load, lanes on consecutive columns.
template <int NT, bool VEC>
__device__ __forceinline__ int col_of(int i, int e, int tid) {
  return VEC ? (i * NT + tid) * 4 + e : (i * 4 + e) * NT + tid;
}

// Loads one item; columns past the end read as -inf (branch-free: clamped address + select).
template <int NT, bool VEC>
__device__ __forceinline__ void load_item(const float* __restrict__ row, int i, int tid, int V,
                                          float (&x)[4]) {
  if (VEC) {
    const int c = (i * NT + tid) * 4;
    const bool ok = c < V;  // V % 4 == 0: all four or none
    const float4 v = *reinterpret_cast<const float4*>(row + (ok ? c : 0));
    x[0] = ok ? v.x : -INFINITY;
    x[1] = ok ? v.y : -INFINITY;
    x[2] = ok ? v.z : -INFINITY;
    x[3] = ok ? v.w : -INFINITY;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = (i * 4 + e) * NT + tid;
      const bool ok = c < V;
      const float v = row[ok ? c : 0];
      x[e] = ok ? v : -INFINITY;
    }
  }
}

constexpr int LIST_CAP = 256;  // contenders a row can hand to wave 0 through LDS

// One workgroup per (image, beam row).
//
// ITEMS > 0: the row fits in registers (V <= NT*4*ITEMS) and is read from memory once.
// ITEMS == 0: any V; the row is streamed twice (statistics, then contenders).
//
// 1. statistics: per-thread (max, sum exp) pairs and the lane maximum over the non-eos
//    columns; each wave reduces them with DPP and also extracts its WL largest lane
//    maxima.  One barrier later every thread knows the row max, the log-sum-exp and the
//    workgroup's WL-th largest lane maximum.  WL distinct elements reach that value, so
//    it bounds the row's WL-th best score from below: only logits at or above it (less a
//    few ulps for rounding in the later arithmetic) and the eos column, which has its own
//    length penalty, can be among the row's best WL.  These are the "contenders" - a
//    handful out of V.
// 2. contenders go to a small LDS list; after the second barrier wave 0 computes their
//    exact scores as the reference does ((lp + ((x - max) - lse)) / penalty, IEEE
//    division) and extracts the best WL in tf.nn.top_k order with DPP argmax rounds.
// 3. if the list overflows (massive ties, or fewer than WL finite logits) every wave
//    joins WL rounds of an exact workgroup-wide argmax instead.
// SOLO (WL == 1, one row per image, one successor: greedy decoding): the row's workgroup also
// does what beam_merge_kernel would do for its image - the finished row's synthetic candidates,
// the gathers and the next state - so the step is one launch.
template <int NT, int ITEMS, int WL, bool VEC, bool SOLO = false>
__global__ __launch_bounds__(NT) void beam_row_kernel(BeamArgs p) {
  static_assert(!SOLO || WL == 1, "the one-launch form serves one row and one successor per image");
  constexpr int NWV = NT / 64;
  constexpr bool CACHED = ITEMS > 0;
  constexpr int NX = CACHED ? ITEMS : 1;
  constexpr int PERB = (NWV * WL + 63) / 64;
  constexpr int PERL = LIST_CAP / 64;
  __shared__ float s_red[2 * NWV];
  __shared__ float s_top[64 * PERB];
  __shared__ float l_x[LIST_CAP];
  __shared__ int l_c[LIST_CAP];
  __shared__ int l_n;
  __shared__ float s_bs[NWV];
  __shared__ float s_bt[NWV];
  __shared__ int s_bi[NWV];

  const int R = p.R, V = p.V;
  const int b = blockIdx.x / R;
  const int w = blockIdx.x % R;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const size_t rw = (size_t)b * R + w;

  if (p.run_flag != nullptr && *p.run_flag == 0) return;
  const bool first = p.time <= 0;
  const bool scan = first ? (w == p.start_offset) : true;
  const bool stats_only = !scan && first && w == 0;  // beam 0 feeds the time-0 log-prob gather
  if (!scan && !stats_only) return;

  const float* row = p.logits + rw * V;
  const int n_items = CACHED ? ITEMS : (V + NT * 4 - 1) / (NT * 4);

  // The row loads are issued before the finished flag is looked at, so the flag's
  // latency is not in front of them (a finished row wastes its reads, nothing else).
  float xs[NX][4];
  if (CACHED) {
#pragma unroll
    for (int i = 0; i < NX; ++i) load_item<NT, VEC>(row, i, tid, V, xs[i]);
  }
  if (p.finished[rw]) {  // candidates of finished beams are made by the merge kernel
    if constexpr (SOLO) {
      if (tid == 0) {
        // as there: log-prob 0 at eos and f32.min elsewhere, so only the first WL + 1 columns
        // and the eos column can win (all other columns tie behind column 0)
        const float lp0 = p.log_probs[rw];
        const int len0 = p.lengths[rw];
        const float pen = penalty_of(p.pen, len0);
        const int ncol = min(V, 2);
        float bs = -INFINITY, bt = 0.f;
        int bi = EMPTY_IDX;
        for (int j = 0; j <= ncol; ++j) {
          int col = -1;
          if (j < ncol) col = j;
          else if (p.eos >= ncol && p.eos < V) col = p.eos;
          if (col < 0) continue;
          const float total = lp0 + ((col == p.eos) ? 0.f : kF32Min);
          const float sc = total / pen;
          if (cand_better(sc, col, bs, bi)) { bs = sc; bt = total; bi = col; }
        }
        const int word = (bi == EMPTY_IDX) ? 0 : bi;
        p.scores[rw] = bs;
        p.ids[rw] = word;
        p.parents[rw] = 0;
        p.next_lp[rw] = bt;
        p.next_fin[rw] = 1;
        p.next_len[rw] = len0;
      }
    }
    return;
  }
  if (tid == 0) l_n = 0;
  for (int q = NWV * WL + tid; q < 64 * PERB; q += NT) s_top[q] = -INFINITY;  // unused slots

  // which (item, element) of this thread is the eos column, if any
  const bool eos_in_row = p.eos >= 0 && p.eos < V;
  int eos_tid, eos_i, eos_e;
  if (VEC) { eos_e = p.eos & 3; eos_tid = (p.eos >> 2) % NT; eos_i = (p.eos >> 2) / NT; }
  else { eos_tid = p.eos % NT; eos_i = (p.eos / NT) >> 2; eos_e = (p.eos / NT) & 3; }
  const bool eos_mine = eos_in_row && tid == eos_tid;

  // ---- statistics ----
  float m = -INFINITY, sm = 0.f, lane_max = -INFINITY;
  if (CACHED) {
#pragma unroll
    for (int i = 0; i < NX; ++i)
      m = fmaxf(m, fmaxf(fmaxf(xs[i][0], xs[i][1]), fmaxf(xs[i][2], xs[i][3])));
    lane_max = m;
    if (eos_mine) {  // one thread of the row: its lane maximum must leave the eos column out
      lane_max = -INFINITY;
#pragma unroll
      for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (!(i == eos_i && e == eos_e)) lane_max = fmaxf(lane_max, xs[i][e]);
    }
    if (!p.input_is_log_prob) {
      const float mm = (m > -INFINITY) ? m : 0.f;
#pragma unroll
      for (int i = 0; i < NX; ++i)
        sm += (fast_exp(xs[i][0] - mm) + fast_exp(xs[i][1] - mm)) +
              (fast_exp(xs[i][2] - mm) + fast_exp(xs[i][3] - mm));
    }
  } else {
    for (int i = 0; i < n_items; ++i) {
      float x[4];
      load_item<NT, VEC>(row, i, tid, V, x);
      const float mn = fmaxf(m, fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3])));
      if (!p.input_is_log_prob && mn > -INFINITY) {
        sm = sm * fast_exp(m - mn) + (fast_exp(x[0] - mn) + fast_exp(x[1] - mn)) +
             (fast_exp(x[2] - mn) + fast_exp(x[3] - mn));
      }
      m = mn;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (!(eos_mine && i == eos_i && e == eos_e)) lane_max = fmaxf(lane_max, x[e]);
    }
  }
  {
    const float mw = wave_max(m);
    const float sw = wave_sum((m > -INFINITY) ? sm * fast_exp(m - mw) : 0.f);
    float lm[1] = {lane_max};
    float topk[WL];
    wave_kth_largest<WL, 1>(lm, lane, topk);
    if (lane == 0) {
      s_red[wave] = mw;
      s_red[NWV + wave] = sw;
#pragma unroll
      for (int r = 0; r < WL; ++r) s_top[wave * WL + r] = topk[r];
    }
  }
  __syncthreads();
  float mx = 0.f, lse = 0.f;
  if (!p.input_is_log_prob) {
    mx = s_red[0];
#pragma unroll
    for (int i = 1; i < NWV; ++i) mx = fmaxf(mx, s_red[i]);
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < NWV; ++i)
      tot += (s_red[i] > -INFINITY) ? s_red[NWV + i] * fast_exp(s_red[i] - mx) : 0.f;
    lse = logf(tot);
  }
  if (tid == 0) p.ws_stats[rw] = make_float2(mx, lse);
  if (stats_only) return;

  const float lp = p.log_probs[rw];
  const int len = p.lengths[rw];
  const int flat0 = first ? 0 : w * V;
  const float pen_eos = penalty_of(p.pen, len);
  const float pen_oth = penalty_of(p.pen, len + 1);
  // un-normalised total log-prob of logit x, exactly as the reference computes it
  auto total_of = [&](float x) -> float {
    const float pr = p.input_is_log_prob ? x : (x - mx) - lse;
    return lp + pr;
  };

  // ---- bound and contender threshold (in logit space) ----
  float thr_x;
  {
    float tv[PERB];
#pragma unroll
    for (int q = 0; q < PERB; ++q) tv[q] = s_top[lane + 64 * q];
    const float bound_x = wave_kth_largest<WL, PERB>(tv, lane, nullptr);
    // Two logits closer than a few ulps of the total can collide after the additions and
    // the division; the slack keeps every element that could tie with the bound.
    const float bound_t = total_of(bound_x);
    thr_x = bound_x - ((fabsf(bound_t) + fabsf(bound_x)) * 9.6e-7f + 1e-37f);
    if (!(thr_x == thr_x)) thr_x = -INFINITY;
  }

  // ---- contenders to the LDS list ----
  auto offer = [&](float x, int c) {
    const int pos = atomicAdd(&l_n, 1);
    if (pos < LIST_CAP) { l_x[pos] = x; l_c[pos] = c; }
  };
  if (CACHED) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const bool any4 = (xs[i][0] >= thr_x) || (xs[i][1] >= thr_x) || (xs[i][2] >= thr_x) ||
                        (xs[i][3] >= thr_x) || (eos_mine && i == eos_i);
      if (any4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = col_of<NT, VEC>(i, e, tid);
          if (c < V && (xs[i][e] >= thr_x || c == p.eos)) offer(xs[i][e], c);
        }
      }
    }
  } else {
    for (int i = 0; i < n_items; ++i) {
      float x[4];
      load_item<NT, VEC>(row, i, tid, V, x);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = col_of<NT, VEC>(i, e, tid);
        if (c < V && (x[e] >= thr_x || c == p.eos)) offer(x[e], c);
      }
    }
  }
  __syncthreads();
  const int n_list = l_n;

  float my_s = -INFINITY, my_t = 0.f;
  int my_i = EMPTY_IDX;
  if (n_list <= LIST_CAP) {
    // ---- common case: wave 0 ranks the listed contenders exactly ----
    if (wave != 0) return;
    float cs[PERL], ct[PERL];
    int ci[PERL];
#pragma unroll
    for (int q = 0; q < PERL; ++q) {
      const int j = lane + 64 * q;
      cs[q] = -INFINITY;
      ct[q] = 0.f;
      ci[q] = EMPTY_IDX;
      if (j < n_list) {
        const int c = l_c[j];
        ct[q] = total_of(l_x[j]);
        cs[q] = ct[q] / ((c == p.eos) ? pen_eos : pen_oth);
        ci[q] = flat0 + c;
      }
    }
    for (int r = 0; r < WL; ++r) {
      float bs = cs[0], bt = ct[0];
      int bi = ci[0];
#pragma unroll
      for (int q = 1; q < PERL; ++q)
        if (cand_better(cs[q], ci[q], bs, bi)) { bs = cs[q]; bt = ct[q]; bi = ci[q]; }
      wave_best(bs, bt, bi);
#pragma unroll
      for (int q = 0; q < PERL; ++q)
        if (ci[q] == bi && bi != EMPTY_IDX) { cs[q] = -INFINITY; ci[q] = EMPTY_IDX; }
      if (lane == r) { my_s = bs; my_t = bt; my_i = bi; }
    }
  } else {
    // ---- overflow: exact workgroup-wide argmax rounds over the whole row ----
    float last_s = INFINITY;
    int last_i = -1;  // the previous winner; eligible elements rank strictly behind it
    for (int r = 0; r < WL; ++r) {
      float bs = -INFINITY, bt = 0.f;
      int bi = EMPTY_IDX;
      auto consider = [&](float x, int c) {
        if (c >= V) return;
        const float t = total_of(x);
        const float sc = t / ((c == p.eos) ? pen_eos : pen_oth);
        const int fi = flat0 + c;
        const bool eligible = (last_i < 0) || sc < last_s || (sc == last_s && fi > last_i);
        if (eligible && cand_better(sc, fi, bs, bi)) { bs = sc; bt = t; bi = fi; }
      };
      // the row is re-read from memory (cache-hot) so this rare path stays small
      const int n_rounds = (V + NT * 4 - 1) / (NT * 4);
      for (int i = 0; i < n_rounds; ++i) {
        float x[4];
        load_item<NT, VEC>(row, i, tid, V, x);
#pragma unroll
        for (int e = 0; e < 4; ++e) consider(x[e], col_of<NT, VEC>(i, e, tid));
      }
      wave_best(bs, bt, bi);
      if (lane == 0) { s_bs[wave] = bs; s_bt[wave] = bt; s_bi[wave] = bi; }
      __syncthreads();
      bs = s_bs[0]; bt = s_bt[0]; bi = s_bi[0];
#pragma unroll
      for (int q = 1; q < NWV; ++q)
        if (cand_better(s_bs[q], s_bi[q], bs, bi)) { bs = s_bs[q]; bt = s_bt[q]; bi = s_bi[q]; }
      __syncthreads();
      if (bi == EMPTY_IDX) break;  // fewer than WL candidates in the row
      last_s = bs;
      last_i = bi;
      if (tid == r) { my_s = bs; my_t = bt; my_i = bi; }
    }
    if (wave != 0) return;
  }
  if constexpr (SOLO) {
    // (every thread of the workgroup loaded its state words before the barriers above, so
    // the next state may be written over the state inputs)
    if (lane == 0) {
      const int flat = (my_i == EMPTY_IDX) ? 0 : my_i;
      const int word = flat % V;
      const int nfin = (word == p.eos) ? 1 : 0;
      p.scores[rw] = my_s;
      p.ids[rw] = word;
      p.parents[rw] = flat / V;
      p.next_lp[rw] = my_t;
      p.next_fin[rw] = (uint8_t)nfin;
      p.next_len[rw] = len + ((word != p.eos && !nfin) ? 1 : 0);
    }
    return;
  }
  if (lane < WL) {
    p.ws_score[rw * WL_MAX + lane] = my_s;
    p.ws_total[rw * WL_MAX + lane] = my_t;
    p.ws_idx[rw * WL_MAX + lane] = my_i;
  }
}

// One wavefront per image, four images per workgroup.  WL >= max(R, W): the R rows'
// candidate slots and the W winners both fit the template.
template <int WL>
__global__ __launch_bounds__(256) void beam_merge_kernel(BeamArgs p) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p.clear_flag != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *p.clear_flag = 0;
  if (p.run_flag != nullptr && *p.run_flag == 0) return;
  if (b >= p.B) return;
  const int W = p.W, R = p.R, V = p.V;
  const bool first = p.time <= 0;
  const int row_lo = first ? p.start_offset : 0;
  const int row_hi = first ? p.start_offset + 1 : R;

  // candidate slots: row w owns slots [w*(WL+2), (w+1)*(WL+2))
  constexpr int SLOTS_PER_ROW = WL + 2;
  constexpr int PER = (WL * SLOTS_PER_ROW + 63) / 64;
  float cs[PER], ct[PER];
  int ci[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int slot = lane + 64 * k;
    const int w = slot / SLOTS_PER_ROW;
    const int j = slot % SLOTS_PER_ROW;
    cs[k] = -INFINITY;
    ct[k] = 0.f;
    ci[k] = EMPTY_IDX;
    if (w >= row_lo && w < row_hi) {
      const size_t rw = (size_t)b * R + w;
      const int flat0 = first ? 0 : w * V;
      if (p.finished[rw]) {
        // Finished beam: the masked log-prob is 0 at eos and f32.min elsewhere whatever
        // the logits are.  All non-eos candidates tie, so only the first WL+1 columns and
        // the eos column can ever be selected.
        const int ncol = min(V, WL + 1);
        int col = -1;
        if (j < ncol) col = j;
        else if (j == ncol && p.eos >= ncol && p.eos < V) col = p.eos;
        if (col >= 0) {
          const float total = p.log_probs[rw] + ((col == p.eos) ? 0.f : kF32Min);
          cs[k] = total / penalty_of(p.pen, p.lengths[rw]);
          ct[k] = total;
          ci[k] = flat0 + col;
        }
      } else if (j < WL) {
        cs[k] = p.ws_score[rw * WL_MAX + j];
        ct[k] = p.ws_total[rw * WL_MAX + j];
        ci[k] = p.ws_idx[rw * WL_MAX + j];
      }
    }
  }

  // W rounds of wave argmax; lane r keeps winner r
  float my_s = -INFINITY, my_t = 0.f;
  int my_i = EMPTY_IDX;
  for (int r = 0; r < W; ++r) {
    float bs = cs[0], bt = ct[0];
    int bi = ci[0];
#pragma unroll
    for (int k = 1; k < PER; ++k)
      if (cand_better(cs[k], ci[k], bs, bi)) { bs = cs[k]; bt = ct[k]; bi = ci[k]; }
    wave_best(bs, bt, bi);
#pragma unroll
    for (int k = 0; k < PER; ++k)
      if (ci[k] == bi && bi != EMPTY_IDX) { cs[k] = -INFINITY; ci[k] = EMPTY_IDX; }
    if (lane == r) { my_s = bs; my_t = bt; my_i = bi; }
  }

  // Lanes 0..W-1 finish one winner each.  With R == W the next-state buffers may be the
  // state inputs themselves: one wave owns the image and every load below precedes every
  // store in program order, so the in-place update is safe.
  const bool act = lane < W;
  const size_t o = (size_t)b * W + (act ? lane : 0);
  const int flat = (my_i == EMPTY_IDX) ? 0 : my_i;
  const int word = flat % V;
  const int parent = flat / V;
  float nlp = my_t;
  if (first && p.start_offset != 0) {
    // reference quirk: at time 0 the log-prob is gathered from beam 0 at the chosen word
    const size_t r0 = (size_t)b * R;
    float pr;
    if (p.finished[r0]) {
      pr = (word == p.eos) ? 0.f : kF32Min;
    } else {
      const float x = p.logits[r0 * V + word];
      const float2 st = p.ws_stats[r0];
      pr = p.input_is_log_prob ? x : (x - st.x) - st.y;
    }
    nlp = p.log_probs[r0] + pr;
  }
  const size_t rp = (size_t)b * R + parent;
  const int pfin = p.finished[rp] ? 1 : 0;
  const int plen = p.lengths[rp];
  const int nfin = (pfin || word == p.eos) ? 1 : 0;
  const int nlen = plen + ((word != p.eos && !nfin) ? 1 : 0);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // keep the loads above the stores
  if (act) {
    p.scores[o] = my_s;
    p.ids[o] = word;
    p.parents[o] = parent;
    p.next_lp[o] = nlp;
    p.next_fin[o] = (uint8_t)nfin;
    p.next_len[o] = nlen;
  }
}

// ---- second half of the fused vocab-projection + beam step ----
template <int WL, int ITEMS>
__global__ __launch_bounds__(256) void beam_records_kernel(RecArgs p) {
  __shared__ RecLds<WL, 256> lds;
  beam_records_body<WL, ITEMS, 256>(p, blockIdx.x, lds);
}

// ---- the same step with one wave per beam row ----
// Within a row the score is a non-decreasing function of the logit (same log-prob, same
// log-sum-exp, same positive penalty), so a row's successors are among its W largest recorded
// logits: wave w folds the records of row w into (max, log-sum-exp) and selects those W on
// 64-bit keys, (logit in an order-preserving integer form) << 32 | ~column, whose unsigned
// order is "higher logit first, then lower column".  A lane merges the sorted lists of its
// records into one sorted list of WL keys, and the DPP ladder of wave_max merges the lanes'
// lists (max of one list against the other reversed leaves the WL largest of both, a fixed
// sorting network orders them): straight-line code, no barrier, no rescans.  Only these W and
// the row's special candidates (the eos column with its own penalty; the synthetic
// candidates of a finished row) are scored with the exact division, 2W + 2 per row at most;
// after ONE barrier wave 0 ranks the image's W * (2W + 2) <= 60 candidates, one per lane, with
// the same network on (score, ~flat index) keys and writes the step's outputs.
typedef unsigned long long u64;

__device__ __forceinline__ unsigned ord_f32(float x) {  // unsigned order == float order (-0 as +0)
  unsigned bts = __float_as_uint(x);
  bts = (bts == 0x80000000u) ? 0u : bts;
  return (bts & 0x80000000u) ? ~bts : (bts | 0x80000000u);
}
__device__ __forceinline__ float unord_f32(unsigned o) {
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}
__device__ __forceinline__ u64 make_key(float x, int idx) {
  return ((u64)ord_f32(x) << 32) | (unsigned)~idx;
}
__device__ __forceinline__ void key_cswap(u64& hi, u64& lo) {  // descending compare-exchange
  const bool g = hi >= lo;
  const u64 h = g ? hi : lo, l = g ? lo : hi;
  hi = h;
  lo = l;
}
template <int N>
__device__ __forceinline__ void key_sort(u64 (&c)[N]) {
  static_assert(N == 1 || N == 3 || N == 5, "networks for 1, 3 and 5 keys");
  if constexpr (N == 3) {
    key_cswap(c[0], c[2]); key_cswap(c[0], c[1]); key_cswap(c[1], c[2]);
  } else if constexpr (N == 5) {
    key_cswap(c[0], c[1]); key_cswap(c[3], c[4]); key_cswap(c[2], c[4]);
    key_cswap(c[2], c[3]); key_cswap(c[1], c[4]); key_cswap(c[0], c[3]);
    key_cswap(c[0], c[2]); key_cswap(c[1], c[3]); key_cswap(c[1], c[2]);
  }
}
// a, b sorted descending -> a = the N largest of both, sorted descending
template <int N>
__device__ __forceinline__ void key_merge(u64 (&a)[N], const u64 (&b)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) a[i] = a[i] >= b[N - 1 - i] ? a[i] : b[N - 1 - i];
  key_sort<N>(a);
}
template <int CTRL, int MASK, int N>
__device__ __forceinline__ void key_ladder_step(u64 (&k)[N]) {
  u64 o[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const unsigned lo = (unsigned)dpp_i<CTRL, MASK>(0, (int)(unsigned)k[i]);
    const unsigned hi = (unsigned)dpp_i<CTRL, MASK>(0, (int)(unsigned)(k[i] >> 32));
    o[i] = ((u64)hi << 32) | lo;
  }
  key_merge<N>(k, o);
}
// every lane hands in a descending list; every lane gets the N largest keys of the wave
template <int N>
__device__ __forceinline__ void wave_top_keys(u64 (&k)[N]) {
  key_ladder_step<ICAP_DPP_ROW_SHR(1), 0xf, N>(k);
  key_ladder_step<ICAP_DPP_ROW_SHR(2), 0xf, N>(k);
  key_ladder_step<ICAP_DPP_ROW_SHR(4), 0xf, N>(k);
  key_ladder_step<ICAP_DPP_ROW_SHR(8), 0xf, N>(k);
  key_ladder_step<ICAP_DPP_BCAST15, 0xa, N>(k);
  key_ladder_step<ICAP_DPP_BCAST31, 0xc, N>(k);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)k[i], 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(k[i] >> 32), 63);
    k[i] = ((u64)hi << 32) | lo;
  }
}

template <int WL, int ITEMS>
__global__ __launch_bounds__(64 * WL) void beam_records_rows_kernel(RecArgs p) {
  constexpr int RL = 2 + 2 * WL;
  constexpr int NC = 2 * WL + 2;  // candidates a row hands over: W picks, then W + 2 special ones
  __shared__ float s_sc[WL][NC], s_tot[WL][NC];
  __shared__ int s_idx[WL][NC];
  __shared__ int s_fin[WL], s_len[WL];
  __shared__ float s_wt[WL];
  const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int W = p.W, V = p.V, HT = p.HT;
  const bool first = p.time <= 0;
  const int n_rows = first ? 1 : W;  // time 0 scores beam 0 only
  const size_t row0 = (size_t)b * W;

  if (w < n_rows) {  // (uniform over the wave)
    // ---- every load of the wave is issued here: lane l owns records l, l + 64, ... ----
    const float* rbase = p.rec + ((size_t)b * p.rpi + w) * HT * RL;
    float4 rr[ITEMS][RL / 4];
    bool okr[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int idx = lane + it * 64;
      okr[it] = idx < HT;
      const float4* src = reinterpret_cast<const float4*>(rbase + (size_t)(okr[it] ? idx : 0) * RL);
#pragma unroll
      for (int q = 0; q < RL / 4; ++q) rr[it][q] = src[q];
    }
    const float lp = p.log_probs[row0 + w];
    const int len = p.lengths[row0 + w];
    const bool fin = p.finished[row0 + w] != 0;
    const float eos_x = p.eos_x[(size_t)b * p.rpi + w];

    // ---- row statistics ----
    float m = -INFINITY;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) m = okr[it] ? fmaxf(m, rr[it][0].x) : m;
    const float M = wave_max(m);
    float sm = 0.f;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const float t = rr[it][0].y * fast_exp(rr[it][0].x - M);
      sm += (okr[it] && rr[it][0].x > -INFINITY) ? t : 0.f;
    }
    const float lse = logf(wave_sum(sm));

    // ---- the row's W largest logits (the eos column is scored apart, a finished row has none) ----
    u64 top[WL];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const float* fl = reinterpret_cast<const float*>(&rr[it][0]);
      u64 lst[WL];
#pragma unroll
      for (int k = 0; k < WL; ++k) {
        const float x = fl[2 + k];
        const int col = __float_as_int(fl[2 + WL + k]);
        const bool ok = okr[it] && !fin && col >= 0 && col < V && col != p.eos && x == x;
        lst[k] = ok ? make_key(x, col) : 0ull;
      }
      // a record is sorted; the one entry dropped from its middle (the eos column) sinks to the end
#pragma unroll
      for (int k = 0; k + 1 < WL; ++k) key_cswap(lst[k], lst[k + 1]);
      if (it == 0) {
#pragma unroll
        for (int k = 0; k < WL; ++k) top[k] = lst[k];
      } else {
        key_merge<WL>(top, lst);
      }
    }
    wave_top_keys<WL>(top);
    u64 mykey = 0ull;
#pragma unroll
    for (int k = 0; k < WL; ++k) mykey = (lane == k) ? top[k] : mykey;

    // ---- exact scores: lanes 0..W-1 the picks, lanes W..2W+1 the special candidates ----
    float sc = -INFINITY, tot = 0.f;
    int fi = EMPTY_IDX;
    const int flat0 = first ? 0 : w * V;
    if (lane < W) {
      if (mykey != 0ull) {
        const float x = unord_f32((unsigned)(mykey >> 32));
        const int col = (int)~(unsigned)mykey;
        tot = lp + ((x - M) - lse);
        const float s = tot / penalty_of(p.pen, len + 1);
        if (s == s) { sc = s; fi = flat0 + col; }
      }
    } else if (lane < 2 * W + 2) {
      const int j = lane - W;
      const float pen = penalty_of(p.pen, len);
      if (fin) {
        // all mass on eos: only the first W+1 columns and the eos column can be selected
        const int ncol = min(V, W + 1);
        int col = -1;
        if (j < ncol) col = j;
        else if (j == ncol && p.eos >= ncol && p.eos < V) col = p.eos;
        if (col >= 0) {
          tot = lp + ((col == p.eos) ? 0.f : kF32Min);
          sc = tot / pen;
          fi = flat0 + col;
        }
      } else if (j == 0 && p.eos >= 0 && p.eos < V) {
        tot = lp + ((eos_x - M) - lse);
        const float s = tot / pen;
        if (s == s) { sc = s; fi = flat0 + p.eos; }
      }
    }
    if (lane < NC) { s_sc[w][lane] = sc; s_tot[w][lane] = tot; s_idx[w][lane] = fi; }
    if (lane == 0) { s_fin[w] = fin ? 1 : 0; s_len[w] = len; }
  }
  __syncthreads();
  if (w != 0) return;

  // ---- wave 0: the image's candidates, one per lane; the W best by (score, flat index) ----
  float ct = 0.f;
  u64 ckey = 0ull;
  {
    const int r = lane / NC, q = lane - r * NC;
    if (r < n_rows && q < 2 * W + 2) {
      const int ci = s_idx[r][q];
      ct = s_tot[r][q];
      ckey = (ci == EMPTY_IDX) ? 0ull : make_key(s_sc[r][q], ci);
    }
  }
  u64 best[WL];
  best[0] = ckey;
#pragma unroll
  for (int k = 1; k < WL; ++k) best[k] = 0ull;
  wave_top_keys<WL>(best);
  // the winners' running log-probs come back through LDS (flat indices are unique)
  if (lane < WL) s_wt[lane] = 0.f;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
  for (int k = 0; k < WL; ++k)
    if (ckey != 0ull && ckey == best[k]) s_wt[k] = ct;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  u64 mine = 0ull;
#pragma unroll
  for (int k = 0; k < WL; ++k) mine = (lane == k) ? best[k] : mine;
  if (lane < W) {
    const bool none = mine == 0ull;
    const float my_s = none ? -INFINITY : unord_f32((unsigned)(mine >> 32));
    const int flat = none ? 0 : (int)~(unsigned)mine;
    const int word = flat % V;
    const int parent = flat / V;  // (< n_rows: flat indices come from scored rows only)
    const int pfin = s_fin[parent], plen = s_len[parent];
    const int nfin = (pfin || word == p.eos) ? 1 : 0;
    const int nlen = plen + ((word != p.eos && !nfin) ? 1 : 0);
    // (the next-state buffers may be the state inputs: every read of the state happened
    // before the barrier)
    const size_t o = row0 + lane;
    p.scores[o] = my_s;
    p.ids[o] = word;
    p.parents[o] = parent;
    p.next_lp[o] = s_wt[lane];
    p.next_fin[o] = (uint8_t)nfin;
    p.next_len[o] = nlen;
  }
}

// ---- the same step for wide beams (W up to 16) and any number of records per image ----
// The records hold each half tile's best WL (< W) logits, so a half tile can hide candidates:
// the kernel proves for every image that none of them matters, and raises *overflow
// otherwise (the caller then repeats the step on full logits).  One workgroup of 1024
// threads per image:
//   A. wave w folds the records of beam row w into the row max and log-sum-exp;
//   B. every thread scores the best logit of its records exactly; the W-th largest of the
//      thread maxima is a lower bound T on the W-th best score (W different candidates reach it);
//   C. every recorded logit whose exact score reaches T is a contender and goes to a list
//      in LDS together with the synthetic candidates (eos, finished rows).  If the LAST
//      stored logit of a record is a contender, logits the record does not hold could be
//      contenders too: overflow.  A full list: overflow.
//   D. wave 0 extracts the W winners from the list in tf.nn.top_k order.
constexpr int WIDE_NT = 1024;
constexpr int WIDE_CAP = 2048;  // contender slots

// ITEMS > 0 (records per lane of a row's wave, HT <= 64 * ITEMS): one pass over the records.
// Wave w folds row w's statistics and keeps every record's best logit in registers (it is the
// record's maximum); the bound T comes from the lanes' best scores, and only records whose
// best logit reaches T are read again in full (their other logits are lower).  ITEMS == 0:
// any HT, three passes (statistics, bound, contenders) - the records of 256 images do not
// stay in L2 between passes, so each pass is one over HBM.
template <int WL, int ITEMS>
__global__ __launch_bounds__(WIDE_NT) void beam_records_wide_kernel(RecArgs p, int* overflow) {
  constexpr int RL = 2 + 2 * WL;
  constexpr int NWV = WIDE_NT / 64;
  constexpr int MAXW = ICAP_MAX_BEAM_WIDTH;
  static_assert(NWV >= MAXW, "one wave per beam row");
  __shared__ float s_M[MAXW], s_L[MAXW], s_lp[MAXW];
  __shared__ int s_len[MAXW], s_fin[MAXW];
  __shared__ float s_top[NWV * MAXW];
  __shared__ float s_thr;
  __shared__ float l_s[WIDE_CAP], l_t[WIDE_CAP];
  __shared__ int l_i[WIDE_CAP];
  __shared__ int l_n, s_over;
  const int W = p.W, V = p.V, HT = p.HT;
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool first = p.time <= 0;
  const int n_rows = first ? 1 : W;  // time 0 scores beam 0 only
  const int n_rec = n_rows * HT;
  const size_t row0 = (size_t)b * W;
  const float* rec0 = p.rec + (size_t)b * p.rpi * HT * RL;

  if (tid < MAXW) {
    const bool ok = tid < n_rows;
    s_lp[tid] = p.log_probs[row0 + (ok ? tid : 0)];
    s_len[tid] = p.lengths[row0 + (ok ? tid : 0)];
    s_fin[tid] = ok ? (p.finished[row0 + tid] != 0 ? 1 : 0) : 1;
  }
  if (tid == 0) { l_n = 0; s_over = 0; }

  // the contender test of one record (all of its logits), shared by both forms
  auto contend = [&](int idx, int w, float thr) {
    const float4* r4 = reinterpret_cast<const float4*>(rec0 + (size_t)idx * RL);
    float4 q4[RL / 4];
#pragma unroll
    for (int q = 0; q < RL / 4; ++q) q4[q] = r4[q];
    const float* fl = reinterpret_cast<const float*>(&q4[0]);
    const float lp = s_lp[w], M = s_M[w], lse = s_L[w];
    const float pen_oth = penalty_of(p.pen, s_len[w] + 1);
    const int flat0 = first ? 0 : w * V;
#pragma unroll
    for (int k = 0; k < WL; ++k) {
      const float x = fl[2 + k];
      const int col = __float_as_int(fl[2 + WL + k]);
      const float tot = lp + ((x - M) - lse);
      const float sc = tot / pen_oth;
      const bool in_range = col >= 0 && col < V;
      // (the eos column is scored by its row's synthetic candidate, with its own penalty)
      if (in_range && sc == sc && sc >= thr) {
        if (k == WL - 1) s_over = 1;  // the record may hide logits that reach the bound too
        if (col != p.eos) {
          const int slot = atomicAdd(&l_n, 1);
          if (slot < WIDE_CAP) { l_s[slot] = sc; l_t[slot] = tot; l_i[slot] = flat0 + col; }
        }
      }
    }
  };
  // the W-th largest of every thread's `tmax`: a lower bound on the W-th best score (W
  // different candidates reach it)
  auto bound_of = [&](float tmax) -> float {
    {
      float v[1] = {tmax};
      float top[MAXW];
      wave_kth_largest<MAXW, 1>(v, lane, top);
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < MAXW; ++k) s_top[wave * MAXW + k] = top[k];
      }
    }
    __syncthreads();
    if (wave == 0) {
      constexpr int PERV = NWV * MAXW / 64;
      float v[PERV];
#pragma unroll
      for (int q = 0; q < PERV; ++q) v[q] = s_top[lane + 64 * q];
      // the W-th largest: W rounds, each removing one instance of the maximum
      float kth = -INFINITY;
      for (int r = 0; r < W; ++r) {
        float m = v[0];
#pragma unroll
        for (int q = 1; q < PERV; ++q) m = fmaxf(m, v[q]);
        const float top = wave_max(m);
        kth = top;
        const unsigned long long owners = __ballot(m == top);
        if (owners != 0 && lane == (int)__builtin_ctzll(owners)) {
          bool done = false;
#pragma unroll
          for (int q = 0; q < PERV; ++q)
            if (!done && v[q] == top) { v[q] = -INFINITY; done = true; }
        }
      }
      if (lane == 0) s_thr = kth;
    }
    __syncthreads();
    return s_thr;  // -inf when fewer than W live candidates exist: everything contends
  };

  if constexpr (ITEMS > 0) {
    // ---- A+B in one pass: wave w <- row w, lane <- records lane, lane + 64, ... ----
    float xs[ITEMS];   // the records' best logits (-inf: no record / no valid column / eos)
    float tbest = -INFINITY;
    __syncthreads();   // s_lp, s_len, s_fin
    if (wave < n_rows) {
      float2 ms[ITEMS];
      int c0[ITEMS];
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) {
        const int h = lane + 64 * j;
        const bool ok = h < HT;
        const float* r = rec0 + ((size_t)wave * HT + (ok ? h : 0)) * RL;
        ms[j] = *reinterpret_cast<const float2*>(r);
        c0[j] = __float_as_int(r[2 + WL]);
        if (!ok) { ms[j] = make_float2(-INFINITY, 0.f); c0[j] = -1; }
      }
      float m = -INFINITY, sm = 0.f;
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) {
        if (ms[j].x > -INFINITY) {
          const float nm = fmaxf(m, ms[j].x);
          sm = sm * fast_exp(m - nm) + ms[j].y * fast_exp(ms[j].x - nm);  // (m = -inf: sm is 0, exp gives 0)
          m = nm;
        }
        // a record's best logit is its maximum; kept while any of its logits could contend
        xs[j] = (c0[j] >= 0 && c0[j] < V) ? ms[j].x : -INFINITY;
        if (c0[j] >= 0 && c0[j] < V && c0[j] != p.eos) tbest = fmaxf(tbest, ms[j].x);
      }
      const float M = wave_max(m);
      const float tot = wave_sum((m > -INFINITY) ? sm * fast_exp(m - M) : 0.f);
      const float L = logf(tot);
      if (lane == 0) { s_M[wave] = M; s_L[wave] = L; }
      // the lane's best score: scores grow with the logit inside a row
      if (!s_fin[wave] && tbest > -INFINITY) {
        const float sc = (s_lp[wave] + ((tbest - M) - L)) / penalty_of(p.pen, s_len[wave] + 1);
        tbest = (sc == sc) ? sc : -INFINITY;
      } else {
        tbest = -INFINITY;
      }
    } else {
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) xs[j] = -INFINITY;
    }
    const float thr = bound_of(tbest);  // (its barriers also publish s_M, s_L)
    // ---- C: records whose best logit reaches the bound (or is not a number) ----
    if (wave < n_rows && !s_fin[wave]) {
      const float lp = s_lp[wave], M = s_M[wave], lse = s_L[wave];
      const float pen_oth = penalty_of(p.pen, s_len[wave] + 1);
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) {
        const float sc0 = (lp + ((xs[j] - M) - lse)) / pen_oth;
        if (xs[j] > -INFINITY && !(sc0 < thr)) contend(wave * HT + lane + 64 * j, wave, thr);
      }
    }
  } else {
  // ---- A: row statistics, wave w <- row w ----
  if (wave < n_rows) {
    float m = -INFINITY, sm = 0.f;
    for (int h = lane; h < HT; h += 64) {
      const float2 v = *reinterpret_cast<const float2*>(rec0 + ((size_t)wave * HT + h) * RL);
      if (v.x > -INFINITY) {
        const float nm = fmaxf(m, v.x);
        sm = sm * fast_exp(m - nm) + v.y * fast_exp(v.x - nm);  // (m = -inf: sm is 0, exp gives 0)
        m = nm;
      }
    }
    const float M = wave_max(m);
    const float tot = wave_sum((m > -INFINITY) ? sm * fast_exp(m - M) : 0.f);
    if (lane == 0) { s_M[wave] = M; s_L[wave] = logf(tot); }
  }
  __syncthreads();

  // ---- B: lower bound on the W-th best score from the records' best logits ----
  float tmax = -INFINITY;
  for (int idx = tid; idx < n_rec; idx += WIDE_NT) {
    const int w = idx / HT;
    if (s_fin[w]) continue;
    const float* r = rec0 + (size_t)idx * RL;
    const float x = r[2];
    const int col = __float_as_int(r[2 + WL]);
    const float tot = s_lp[w] + ((x - s_M[w]) - s_L[w]);
    const float sc = tot / penalty_of(p.pen, s_len[w] + 1);
    if (col >= 0 && col < V && col != p.eos && sc == sc) tmax = fmaxf(tmax, sc);
  }
  const float thr = bound_of(tmax);

  // ---- C: contenders ----
  for (int idx = tid; idx < n_rec; idx += WIDE_NT) {
    const int w = idx / HT;
    if (s_fin[w]) continue;
    contend(idx, w, thr);
  }
  }
  // synthetic candidates: thread w*(W+2)+j makes the j-th candidate of a finished row w (all
  // its mass on eos: only the first W+1 columns and eos can be selected) and, for j == 0, the
  // eos candidate of a live row
  {
    const int sp_w = tid / (W + 2), sp_j = tid - sp_w * (W + 2);
    if (sp_w < n_rows) {
      const float lp = s_lp[sp_w];
      const int flat0 = first ? 0 : sp_w * V;
      const float pen = penalty_of(p.pen, s_len[sp_w]);
      float sc = -INFINITY, tot = 0.f;
      int fi = EMPTY_IDX;
      if (s_fin[sp_w]) {
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
        const float ex = p.eos_x[(size_t)b * p.rpi + sp_w];
        tot = lp + ((ex - s_M[sp_w]) - s_L[sp_w]);
        sc = tot / pen;
        fi = (sc == sc) ? flat0 + p.eos : EMPTY_IDX;
      }
      if (fi != EMPTY_IDX) {
        const int slot = atomicAdd(&l_n, 1);
        if (slot < WIDE_CAP) { l_s[slot] = sc; l_t[slot] = tot; l_i[slot] = fi; }
      }
    }
  }
  __syncthreads();
  const int n_list = l_n;
  if (tid == 0 && (s_over != 0 || n_list > WIDE_CAP)) *overflow = 1;
  if (wave != 0) return;

  // ---- D: W winners (score descending, lower flat index first); lane r keeps winner r ----
  const int n = min(n_list, WIDE_CAP);
  float my_s = -INFINITY, my_t = 0.f;
  int my_i = EMPTY_IDX;
  for (int r = 0; r < W; ++r) {
    float bs = -INFINITY, bt = 0.f;
    int bi = EMPTY_IDX, bslot = -1;
    for (int q = lane; q < n; q += 64) {
      const float cs_ = l_s[q];
      const int ci_ = l_i[q];
      if (cand_better(cs_, ci_, bs, bi)) { bs = cs_; bt = l_t[q]; bi = ci_; bslot = q; }
    }
    const int own = bi;
    wave_best(bs, bt, bi);
    if (bi != EMPTY_IDX && own == bi && bslot >= 0) l_i[bslot] = EMPTY_IDX;  // flat indices are unique
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the next round reads the list again
    if (lane == r) { my_s = bs; my_t = bt; my_i = bi; }
  }
  const bool act = lane < W;
  const int flat = (my_i == EMPTY_IDX) ? 0 : my_i;
  const int word = flat % V;
  const int parent = flat / V;
  const int pfin = s_fin[parent < MAXW ? parent : 0];
  const int plen = s_len[parent < MAXW ? parent : 0];
  const int nfin = (pfin || word == p.eos) ? 1 : 0;
  const int nlen = plen + ((word != p.eos && !nfin) ? 1 : 0);
  if (act) {
    const size_t o = row0 + lane;
    p.scores[o] = my_s;
    p.ids[o] = word;
    p.parents[o] = parent;
    p.next_lp[o] = my_t;
    p.next_fin[o] = (uint8_t)nfin;
    p.next_len[o] = nlen;
  }
}

// ---- back-tracking: one thread per image ----
__global__ __launch_bounds__(256) void backtrack_kernel(const int32_t* __restrict__ ids,
                                                        const int32_t* __restrict__ parents,
                                                        const float* __restrict__ last_scores,
                                                        const uint8_t* __restrict__ last_finished,
                                                        int32_t* __restrict__ tokens,
                                                        int32_t* __restrict__ best_beam,
                                                        float* __restrict__ best_score, int T,
                                                        int B, int W,
                                                        const int* __restrict__ invalid) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  if (invalid != nullptr && *invalid != 0) {
    // an earlier kernel of the decode reported that it could not do its work: no caption
    if (best_beam) best_beam[b] = -1;
    if (best_score) best_score[b] = __int_as_float(0x7fc00000);
    for (int t = 0; t < T; ++t) tokens[(size_t)b * T + t] = -1;
    return;
  }
  bool any_fin = false;
  for (int k = 0; k < W; ++k) any_fin = any_fin || last_finished[(size_t)b * W + k];
  int best = 0;
  float best_s = 0.f;
  for (int k = 0; k < W; ++k) {
    float s = last_scores[(size_t)b * W + k];
    if (any_fin) s = s + (1.0f - (last_finished[(size_t)b * W + k] ? 1.0f : 0.0f)) * -1e9f;
    if (k == 0 || s > best_s) { best = k; best_s = s; }
  }
  if (best_beam) best_beam[b] = best;
  if (best_score) best_score[b] = best_s;
  int beam = best;
  for (int t = T - 1; t >= 0; --t) {
    const size_t o = ((size_t)t * B + b) * W + beam;
    tokens[(size_t)b * T + t] = ids[o];
    beam = parents[o];
    if (beam < 0 || beam >= W) beam = 0;  // corrupt parent: stay in range
  }
}

template <int NT, int ITEMS, int WL>
int launch_row(const BeamArgs& a, bool vec, hipStream_t s, bool solo = false) {
  const dim3 grid(a.B * a.R), block(NT);
  if constexpr (WL == 1) {
    if (solo) {
      if (vec)
        hipLaunchKernelGGL((beam_row_kernel<NT, ITEMS, 1, true, true>), grid, block, 0, s, a);
      else
        hipLaunchKernelGGL((beam_row_kernel<NT, ITEMS, 1, false, true>), grid, block, 0, s, a);
      ICAP_LAUNCH_CHECK("beam_row_kernel<solo>");
      return ICAP_OK;
    }
  }
  if (vec)
    hipLaunchKernelGGL((beam_row_kernel<NT, ITEMS, WL, true>), grid, block, 0, s, a);
  else
    hipLaunchKernelGGL((beam_row_kernel<NT, ITEMS, WL, false>), grid, block, 0, s, a);
  ICAP_LAUNCH_CHECK("beam_row_kernel");
  return ICAP_OK;
}

template <int WL>
int dispatch_beam(const BeamArgs& a, bool vec, hipStream_t s) {
  int rc;
  const int V = a.V;
  // greedy decoding (one row, one successor per image, no device-side switch): the row kernel
  // finishes the step itself.  Tuning knob ICAP_BEAM_SOLO=0: two launches.
  static const char* solo_env = getenv("ICAP_BEAM_SOLO");
  const bool solo = WL == 1 && a.R == 1 && a.W == 1 && !a.run_flag && !a.clear_flag &&
                    !(solo_env && solo_env[0] == '0');
  if (solo) {
    if (V <= 256 * 4 * 4) return launch_row<256, 4, WL>(a, vec, s, true);
    if (V <= 256 * 4 * 10) return launch_row<256, 10, WL>(a, vec, s, true);
    if (V <= 256 * 4 * 16) return launch_row<256, 16, WL>(a, vec, s, true);
    if (V <= 1024 * 4 * 13) return launch_row<1024, 13, WL>(a, vec, s, true);
    if (V <= 1024 * 4 * 16) return launch_row<1024, 16, WL>(a, vec, s, true);
    return launch_row<1024, 0, WL>(a, vec, s, true);
  }
  if (V <= 256 * 4 * 4) rc = launch_row<256, 4, WL>(a, vec, s);
  else if (V <= 256 * 4 * 10) rc = launch_row<256, 10, WL>(a, vec, s);
  else if (V <= 256 * 4 * 16) rc = launch_row<256, 16, WL>(a, vec, s);
  else if (V <= 1024 * 4 * 13) rc = launch_row<1024, 13, WL>(a, vec, s);
  else if (V <= 1024 * 4 * 16) rc = launch_row<1024, 16, WL>(a, vec, s);
  else rc = launch_row<1024, 0, WL>(a, vec, s);  // streamed rows
  if (rc != ICAP_OK) return rc;
  hipLaunchKernelGGL(beam_merge_kernel<WL>, dim3(ceil_div(a.B, 4)), dim3(256), 0, s, a);
  ICAP_LAUNCH_CHECK("beam_merge_kernel");
  return ICAP_OK;
}

}  // namespace

size_t beam_step_workspace_bytes(int B, int W) {
  if (B <= 0 || W <= 0) return 0;
  const size_t rows = (size_t)B * W;
  return rows * sizeof(float2) + 3 * rows * WL_MAX * sizeof(float);
}

// R input rows (beams) per image, W successors out: state inputs [B,R], outputs [B,W].  The
// reference's step takes any number of rows (beam_search.py:375-455 reshapes by vocab size
// only); its beam-search-optimisation trainer passes gold + search rows, R = W + 1
// (utils/train_utils.py:275-279).
int beam_step_rows_f32(const float* logits, const float* log_probs, const uint8_t* finished,
                       const int32_t* lengths, int time, int start_offset, int eos_token,
                       double length_penalty_weight, int input_is_log_prob, float* scores,
                       int32_t* predicted_ids, int32_t* parent_ids, float* next_log_probs,
                       uint8_t* next_finished, int32_t* next_lengths, int B, int R, int W, int V,
                       void* workspace, size_t workspace_bytes, hipStream_t s,
                       const int* run_flag, int* clear_flag) {
  ICAP_REQUIRE(logits && log_probs && finished && lengths, "beam_step_f32: null input");
  ICAP_REQUIRE(scores && predicted_ids && parent_ids && next_log_probs && next_finished &&
                   next_lengths,
               "beam_step_f32: null output");
  ICAP_REQUIRE(B > 0 && V > 0, "beam_step_f32: bad shape");
  ICAP_REQUIRE(W >= 1 && W <= ICAP_MAX_BEAM_WIDTH, "beam_step_f32: beam width %d outside [1,%d]",
               W, ICAP_MAX_BEAM_WIDTH);
  ICAP_REQUIRE(R >= 1 && R <= ICAP_MAX_BEAM_WIDTH, "beam_step_f32: %d rows per image outside [1,%d]",
               R, ICAP_MAX_BEAM_WIDTH);
  ICAP_REQUIRE(W <= V, "beam_step_f32: beam width %d exceeds vocab size %d", W, V);
  ICAP_REQUIRE(start_offset >= 0 && start_offset < R,
               "beam_step_f32: start_offset %d outside [0,%d)", start_offset, R);
  ICAP_REQUIRE((long long)R * V < (1LL << 31), "beam_step_f32: rows*V overflows int32");
  ICAP_REQUIRE(workspace && aligned16(workspace),
               "beam_step_f32: workspace must be a 16-byte aligned device pointer");
  ICAP_REQUIRE(workspace_bytes >= beam_step_workspace_bytes(B, R),
               "beam_step_f32: workspace of %zu bytes is smaller than the %zu needed",
               workspace_bytes, beam_step_workspace_bytes(B, R));
  BeamArgs a;
  a.logits = logits; a.log_probs = log_probs; a.finished = finished; a.lengths = lengths;
  a.time = time; a.start_offset = start_offset; a.eos = eos_token;
  a.input_is_log_prob = input_is_log_prob;
  a.scores = scores; a.ids = predicted_ids; a.parents = parent_ids;
  a.next_lp = next_log_probs; a.next_fin = next_finished; a.next_len = next_lengths;
  a.B = B; a.W = W; a.V = V; a.R = R;
  const size_t rows = (size_t)B * R;
  char* wsp = static_cast<char*>(workspace);
  a.ws_stats = reinterpret_cast<float2*>(wsp);
  a.ws_score = reinterpret_cast<float*>(wsp + rows * sizeof(float2));
  a.ws_total = a.ws_score + rows * WL_MAX;
  a.ws_idx = reinterpret_cast<int*>(a.ws_total + rows * WL_MAX);
  a.pen = make_penalty(length_penalty_weight);
  a.run_flag = run_flag;
  a.clear_flag = clear_flag;
  const bool vec = (V % 4 == 0) && aligned16(logits);
  const int wl = W > R ? W : R;  // candidate slots per row: the winners and the rows both fit
  if (wl <= 1) return dispatch_beam<1>(a, vec, s);
  if (wl <= 3) return dispatch_beam<3>(a, vec, s);
  if (wl <= 5) return dispatch_beam<5>(a, vec, s);
  if (wl <= 10) return dispatch_beam<10>(a, vec, s);
  return dispatch_beam<16>(a, vec, s);
}

int beam_step_f32(const float* logits, const float* log_probs, const uint8_t* finished,
                  const int32_t* lengths, int time, int start_offset, int eos_token,
                  double length_penalty_weight, int input_is_log_prob, float* scores,
                  int32_t* predicted_ids, int32_t* parent_ids, float* next_log_probs,
                  uint8_t* next_finished, int32_t* next_lengths, int B, int W, int V,
                  void* workspace, size_t workspace_bytes, hipStream_t s) {
  return beam_step_rows_f32(logits, log_probs, finished, lengths, time, start_offset, eos_token,
                            length_penalty_weight, input_is_log_prob, scores, predicted_ids,
                            parent_ids, next_log_probs, next_finished, next_lengths, B, W, W, V,
                            workspace, workspace_bytes, s, nullptr, nullptr);
}

// ---- the beam step for callers that cannot allocate -------------------------------------
// Same kernels as beam_step_f32; the candidate workspace is a fixed array in device memory
// instead of a caller buffer.  Every GPU the library runs on holds its own copy of the array
// (at its own address) until the process ends.
constexpr size_t STATIC_WS_BYTES =
    (size_t)ICAP_BEAM_STEP_STATIC_MAX_ROWS * (sizeof(float2) + 3 * WL_MAX * sizeof(float));
static __device__ __align__(256) unsigned char scratch[STATIC_WS_BYTES];

namespace {
constexpr int MAX_DEVICES = 64;
std::mutex scratch_mutex;
void* scratch_ptr[MAX_DEVICES] = {};  // device address of `scratch` per device ordinal

// The device address of the array on the device of stream `s`, looked up once per device (the
// array's name is only a host-side handle; kernels get this pointer as an argument).  The
// lookup is a runtime call that may load the code object on its first use, so it must not
// happen under stream capture: the first call on a device has to come from outside a capture.
int static_scratch(hipStream_t s, void** out) {
  int dev = -1, cur = -1;
  ICAP_HIP(hipGetDevice(&cur));
  if (s == nullptr || hipStreamGetDevice(s, &dev) != hipSuccess) {
    (void)hipGetLastError();
    dev = cur;
  }
  ICAP_REQUIRE(dev >= 0 && dev < MAX_DEVICES, "beam_step_static_f32: device ordinal %d not served",
               dev);
  ICAP_REQUIRE(dev == cur, "beam_step_static_f32: the stream belongs to device %d, the current device is %d",
               dev, cur);
  std::lock_guard<std::mutex> lock(scratch_mutex);
  if (!scratch_ptr[dev]) {
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &capture) != hipSuccess) (void)hipGetLastError();
    ICAP_REQUIRE(capture == hipStreamCaptureStatusNone,
                 "beam_step_static_f32: the first call on a device looks the static workspace's address up "
                 "and must be made outside stream capture (device %d)", dev);
    void* p = nullptr;
    ICAP_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    ICAP_REQUIRE(p != nullptr, "beam_step_static_f32: no device address for the static workspace");
    scratch_ptr[dev] = p;
  }
  *out = scratch_ptr[dev];
  return ICAP_OK;
}
}  // namespace

int beam_step_static_f32(const float* logits, const float* log_probs, const uint8_t* finished,
                         const int32_t* lengths, int time, int start_offset, int eos_token,
                         double length_penalty_weight, int input_is_log_prob, float* scores,
                         int32_t* predicted_ids, int32_t* parent_ids, float* next_log_probs,
                         uint8_t* next_finished, int32_t* next_lengths, int B, int W, int V,
                         hipStream_t s) {
  ICAP_REQUIRE(B > 0 && W > 0, "beam_step_static_f32: bad shape");
  if ((long long)B * W > ICAP_BEAM_STEP_STATIC_MAX_ROWS)
    return fail(ICAP_ERR_UNSUPPORTED,
                "beam_step_static_f32: %lld rows (B*W) exceed the static workspace of %d rows; use "
                "icap_beam_step_f32 with a caller workspace",
                (long long)B * W, ICAP_BEAM_STEP_STATIC_MAX_ROWS);
  void* ws = nullptr;
  const int rc = static_scratch(s, &ws);
  if (rc != ICAP_OK) return rc;
  // every slot the merge kernel reads is written by the row kernel of the same call (statistics
  // and all WL candidate slots of every row, empty ones marked), so nothing needs clearing
  return beam_step_rows_f32(logits, log_probs, finished, lengths, time, start_offset, eos_token,
                            length_penalty_weight, input_is_log_prob, scores, predicted_ids,
                            parent_ids, next_log_probs, next_finished, next_lengths, B, W, W, V, ws,
                            STATIC_WS_BYTES, s, nullptr, nullptr);
}

// Records one workgroup can hold in registers (8 per thread); more take the unfused kernels.
bool beam_records_supported(int W, int half_tiles, int top_k) {
  return W >= 1 && W <= top_k && half_tiles > 0 && (long long)W * half_tiles <= 8 * 256 &&
         256 >= W * (W + 2);
}

int beam_step_from_records_f32(const float* records, const float* eos_logit, int half_tiles,
                               int top_k, const float* log_probs, const uint8_t* finished,
                               const int32_t* lengths, int time, int eos_token,
                               double length_penalty_weight, float* scores,
                               int32_t* predicted_ids, int32_t* parent_ids, float* next_log_probs,
                               uint8_t* next_finished, int32_t* next_lengths, int B, int W, int V,
                               hipStream_t s, int compact_first) {
  ICAP_REQUIRE(records && eos_logit && log_probs && finished && lengths,
               "beam_step_from_records_f32: null input");
  ICAP_REQUIRE(scores && predicted_ids && parent_ids && next_log_probs && next_finished &&
                   next_lengths,
               "beam_step_from_records_f32: null output");
  ICAP_REQUIRE(B > 0 && V > 0 && half_tiles > 0, "beam_step_from_records_f32: bad shape");
  ICAP_REQUIRE(top_k == 1 || top_k == 3 || top_k == 5,
               "beam_step_from_records_f32: top_k must be 1, 3 or 5 (got %d)", top_k);
  ICAP_REQUIRE(W >= 1 && W <= top_k, "beam_step_from_records_f32: beam width %d exceeds top_k %d",
               W, top_k);
  ICAP_REQUIRE(W <= V && (long long)W * V < (1LL << 31), "beam_step_from_records_f32: bad W, V");
  ICAP_REQUIRE(beam_records_supported(W, half_tiles, top_k),
               "beam_step_from_records_f32: %d rows x %d records exceed one workgroup", W,
               half_tiles);
  RecArgs a;
  a.rec = records; a.eos_x = eos_logit; a.HT = half_tiles;
  ICAP_REQUIRE(!compact_first || time <= 0,
               "beam_step_from_records_f32: one record row per image only serves time 0");
  a.rpi = compact_first ? 1 : W;
  a.log_probs = log_probs; a.finished = finished; a.lengths = lengths;
  a.time = time; a.eos = eos_token;
  a.scores = scores; a.ids = predicted_ids; a.parents = parent_ids;
  a.next_lp = next_log_probs; a.next_fin = next_finished; a.next_len = next_lengths;
  a.B = B; a.W = W; a.V = V;
  a.pen = make_penalty(length_penalty_weight);
  // One wave per beam row while a lane holds at most 8 records of its row; the one-workgroup
  // form otherwise (and on request: ICAP_BEAM_RECORDS_FORM=wg, a tuning knob).
  const int row_items = ceil_div(half_tiles, 64);
  static const char* form_env = getenv("ICAP_BEAM_RECORDS_FORM");
  if (row_items <= 8 && !(form_env && form_env[0] == 'w')) {
    const dim3 grid(B);
#define ICAP_ROWS_LAUNCH(WL_)                                                                        \
  {                                                                                                  \
    const dim3 block(64 * WL_);                                                                      \
    if (row_items <= 2) hipLaunchKernelGGL((beam_records_rows_kernel<WL_, 2>), grid, block, 0, s, a);      \
    else if (row_items <= 3) hipLaunchKernelGGL((beam_records_rows_kernel<WL_, 3>), grid, block, 0, s, a); \
    else if (row_items <= 4) hipLaunchKernelGGL((beam_records_rows_kernel<WL_, 4>), grid, block, 0, s, a); \
    else hipLaunchKernelGGL((beam_records_rows_kernel<WL_, 8>), grid, block, 0, s, a);                     \
  }
    if (top_k == 1) ICAP_ROWS_LAUNCH(1)
    else if (top_k == 3) ICAP_ROWS_LAUNCH(3)
    else ICAP_ROWS_LAUNCH(5)
#undef ICAP_ROWS_LAUNCH
    ICAP_LAUNCH_CHECK("beam_records_rows_kernel");
    return ICAP_OK;
  }
  const int items = ceil_div((time <= 0 ? 1 : W) * half_tiles, 256);
  const dim3 grid(B), block(256);
#define ICAP_REC_LAUNCH(WL_)                                                                    \
  if (items <= 2) hipLaunchKernelGGL((beam_records_kernel<WL_, 2>), grid, block, 0, s, a);      \
  else if (items <= 4) hipLaunchKernelGGL((beam_records_kernel<WL_, 4>), grid, block, 0, s, a); \
  else hipLaunchKernelGGL((beam_records_kernel<WL_, 8>), grid, block, 0, s, a);
  if (top_k == 1) { ICAP_REC_LAUNCH(1) }
  else if (top_k == 3) { ICAP_REC_LAUNCH(3) }
  else { ICAP_REC_LAUNCH(5) }
#undef ICAP_REC_LAUNCH
  ICAP_LAUNCH_CHECK("beam_records_kernel");
  return ICAP_OK;
}

// The fused step for beams wider than the records are deep (W <= ICAP_MAX_BEAM_WIDTH on
// records of top_k = 5 logits per half tile).  *overflow (device word, zero on entry) is set
// when the records cannot prove the result for some image; the outputs are then to be
// recomputed from full logits (beam_step_f32).
int beam_step_from_records_wide_f32(const float* records, const float* eos_logit, int half_tiles,
                                    int top_k, const float* log_probs, const uint8_t* finished,
                                    const int32_t* lengths, int time, int eos_token,
                                    double length_penalty_weight, float* scores,
                                    int32_t* predicted_ids, int32_t* parent_ids,
                                    float* next_log_probs, uint8_t* next_finished,
                                    int32_t* next_lengths, int B, int W, int V, int* overflow,
                                    hipStream_t s, int compact_first) {
  ICAP_REQUIRE(records && eos_logit && log_probs && finished && lengths && overflow,
               "beam_step_from_records_wide_f32: null input");
  ICAP_REQUIRE(scores && predicted_ids && parent_ids && next_log_probs && next_finished &&
                   next_lengths,
               "beam_step_from_records_wide_f32: null output");
  ICAP_REQUIRE(B > 0 && V > 0 && half_tiles > 0, "beam_step_from_records_wide_f32: bad shape");
  ICAP_REQUIRE(top_k == 5, "beam_step_from_records_wide_f32: top_k must be 5 (got %d)", top_k);
  ICAP_REQUIRE(W >= 1 && W <= ICAP_MAX_BEAM_WIDTH,
               "beam_step_from_records_wide_f32: beam width %d outside [1,%d]", W,
               ICAP_MAX_BEAM_WIDTH);
  ICAP_REQUIRE(W <= V && (long long)W * V < (1LL << 31), "beam_step_from_records_wide_f32: bad W, V");
  ICAP_REQUIRE((long long)W * half_tiles < (1LL << 24),
               "beam_step_from_records_wide_f32: too many records per image");
  ICAP_REQUIRE(next_finished != finished && next_lengths != lengths,
               "beam_step_from_records_wide_f32: the next state must not alias the state (a "
               "recomputation after an overflow reads the old one)");
  ICAP_REQUIRE(!compact_first || time <= 0,
               "beam_step_from_records_wide_f32: one record row per image only serves time 0");
  RecArgs a;
  a.rec = records; a.eos_x = eos_logit; a.HT = half_tiles;
  a.rpi = compact_first ? 1 : W;
  a.log_probs = log_probs; a.finished = finished; a.lengths = lengths;
  a.time = time; a.eos = eos_token;
  a.scores = scores; a.ids = predicted_ids; a.parents = parent_ids;
  a.next_lp = next_log_probs; a.next_fin = next_finished; a.next_len = next_lengths;
  a.B = B; a.W = W; a.V = V;
  a.pen = make_penalty(length_penalty_weight);
  // (ICAP_WIDE_BEAM_FORM=passes: the three-pass form at any size; tuning / test knob)
  const char* form_env = getenv("ICAP_WIDE_BEAM_FORM");
  const bool passes = form_env && form_env[0] == 'p';
  if (!passes && half_tiles <= 64 * 4)
    hipLaunchKernelGGL((beam_records_wide_kernel<5, 4>), dim3(B), dim3(WIDE_NT), 0, s, a, overflow);
  else if (!passes && half_tiles <= 64 * 16)
    hipLaunchKernelGGL((beam_records_wide_kernel<5, 16>), dim3(B), dim3(WIDE_NT), 0, s, a, overflow);
  else
    hipLaunchKernelGGL((beam_records_wide_kernel<5, 0>), dim3(B), dim3(WIDE_NT), 0, s, a, overflow);
  ICAP_LAUNCH_CHECK("beam_records_wide_kernel");
  return ICAP_OK;
}

// invalid: optional device word; when it is not zero every caption comes out as -1 tokens with
// a NaN score (the engine's way of failing loudly from inside a stream).
int backtrack_guarded(const int32_t* ids, const int32_t* parents, const float* last_scores,
                      const uint8_t* last_finished, int32_t* tokens, int32_t* best_beam,
                      float* best_score, int T, int B, int W, const int* invalid, hipStream_t s) {
  ICAP_REQUIRE(ids && parents && last_scores && last_finished && tokens, "backtrack: null pointer");
  ICAP_REQUIRE(T > 0 && B > 0 && W > 0, "backtrack: bad shape");
  hipLaunchKernelGGL(backtrack_kernel, dim3(ceil_div(B, 256)), dim3(256), 0, s, ids, parents,
                     last_scores, last_finished, tokens, best_beam, best_score, T, B, W, invalid);
  ICAP_LAUNCH_CHECK("backtrack_kernel");
  return ICAP_OK;
}

int backtrack(const int32_t* ids, const int32_t* parents, const float* last_scores,
              const uint8_t* last_finished, int32_t* tokens, int32_t* best_beam,
              float* best_score, int T, int B, int W, hipStream_t s) {
  return backtrack_guarded(ids, parents, last_scores, last_finished, tokens, best_beam, best_score,
                           T, B, W, nullptr, s);
}

}  // namespace icap
