// The whole step loop of a greedy decode of a few dozen images in ONE launch.
//
// At beam width 1 and 32 images a decode step is five short kernels (cell, query, attention,
// vocab projection, beam step) of 8 to 15 us each, most of it launch and drain: the products
// are weight streams of 1 to 20 MB and the rest is a few hundred kilobytes.  Here one
// grid (one workgroup of 16 waves per compute unit) stays resident for all T steps
// (recurrent_example.py:517-558) and walks three phases per step, separated by grid barriers:
//
//   L   cell step            [ctx | h] . [K; R] + emb_proj[id] + b, LSTM update   H/8 blocks per 32 rows
//   QV  query + vocab        q = c . W2 + b2 (next step) and logits = h . Wout + bout   (U + V)/32 blocks
//   AB  attention + beam     attention of the next step (one image per block) beside the
//                            greedy beam step of this one (one row per block)
//
// The products use the small-batch scheme of the stand-alone kernels (a block owns 32 rows x 32
// columns, its sixteen waves split K, operands go from memory straight into the MFMA registers,
// partial blocks meet in LDS).  Between phases data changes hands between compute units on
// different XCDs, whose L2s are not coherent with each other.  What changes hands (h, c, q, ctx,
// the logits, ids and beam state: a few hundred kilobytes per step) is written through with
// device-scope stores and read with device-scope loads; the grid barrier is then only a count -
// every thread waits for its stores' acknowledgement, the workgroup adds itself to a device
// counter and one lane polls it.  No cache-wide release or acquire: those write back and empty
// a whole L2 per workgroup and barrier, and the 27 MB of weights that every step reads again
// (an eighth per XCD, which its 4 MB L2 holds) would come from memory every time.
// Every wait is bounded: a barrier that runs out raises *err, every workgroup leaves, and the
// caller turns the result into -1 tokens.
#include <stdlib.h>

#include "attention_body.h"
#include "beam_shared.h"
#include "icap_common.h"

namespace icap {

struct SmallLoopArgs {
  int B, C, H, U, L, D, V, T;  // B rows (= images: beam width 1); C = context width
  int eos, start_id;
  // weights (Keras layouts)
  const float *cell_kernel, *cell_recurrent, *cell_bias, *emb_proj;  // [C+E,4H] rows [0,C), [H,4H], [4H], [V,4H]
  const float *w2, *b2;                                              // [H,U], [U]
  const float* att_v;  float att_bv;
  const float* fc3_w;  float fc3_b;
  const float *wout, *bout;                                          // [H,V], [V]
  // per-decode operands
  const float *pf, *epf, *features;
  const int* pf_extreme;
  float *h0, *h1, *c, *q, *ctx, *logits;  // [B,H] x 3, [B,U], [B,D], [B,V]
  uint8_t* fin;                           // beam state [B], updated in place
  int32_t* len;
  int32_t *hist_ids, *hist_parents;       // [T,B]
  float *hist_scores, *hist_lp;
  Penalty pen;
  unsigned* bar;  // barrier count, zero at launch
  int* err;       // raised when a barrier runs out
};

#ifdef ICAP_TIMELINE
// Experiment builds only: workgroup 0 leaves the 100 MHz clock behind every phase and barrier,
// read back through icap_debug_small_loop_stamps.
__device__ long long icap_sl_stamps[4096];
#define ICAP_SL_STAMP(i) if (blockIdx.x == 0 && threadIdx.x == 0 && (i) < 4096) icap_sl_stamps[i] = wall_clock64();
#else
#define ICAP_SL_STAMP(i)
#endif

namespace {

constexpr int SL_NT = 1024;
constexpr int SL_LDS_BYTES = 72 * 1024;
#ifndef SL_POLL_SLEEP
#define SL_POLL_SLEEP 1  // x 64 clocks between two polls of a waiting workgroup
#endif
constexpr unsigned SL_POLL_LIMIT = 1u << 22;  // x the sleep: a few hundred milliseconds and more

template <typename T>
__device__ __forceinline__ T ld_dev(const T* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Data that changes hands between workgroups inside the launch (h, c, q, ctx, the logits, the
// beam state) is written with device-scope stores - through to where every XCD reads it - and
// read with device-scope loads, which do not take a line from the reader's L2.
template <typename T>
__device__ __forceinline__ void st_dev(T* p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float4 ld_dev4(const float* p) {  // 16-byte aligned
  const unsigned long long* s2 = reinterpret_cast<const unsigned long long*>(p);
  const unsigned long long lo = __hip_atomic_load(s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long hi = __hip_atomic_load(s2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_float4(__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)),
                     __uint_as_float((unsigned)hi), __uint_as_float((unsigned)(hi >> 32)));
}

// All workgroups of the grid meet here.  Returns false when the wait ran out (here or, as seen
// through *err, anywhere else): the caller leaves the kernel.
__device__ __forceinline__ bool grid_barrier(const SmallLoopArgs& p, unsigned& epoch, int* s_flag) {
  // every thread's device-scope stores are acknowledged (written through) before the workgroup
  // counts itself; no cache-wide fence on either side of the wait: a release would write a whole
  // L2 back and an acquire would empty it - of the weights, which every phase of every step
  // reads again and which nothing in the launch writes
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  epoch += 1;
  if (threadIdx.x == 0) {
    const unsigned target = epoch * gridDim.x;
    __hip_atomic_fetch_add(p.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool ok = true;
    unsigned polls = 0;
    while (ld_dev(p.bar) < target) {
      __builtin_amdgcn_s_sleep(SL_POLL_SLEEP);
      if (++polls > SL_POLL_LIMIT || ((polls & 255u) == 0 && ld_dev(p.err) != 0)) { ok = false; break; }
    }
    if (!ok) __hip_atomic_store(p.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *s_flag = ok ? 1 : 0;
  }
  __syncthreads();
  const bool ok = *s_flag != 0;
  __syncthreads();  // (the word is part of the phases' LDS)
  return ok;
}

// ---- phase L: one block = 32 rows x 8 units (4 gates), K = C + H split over the 16 waves ----
template <int NG>
__device__ __forceinline__ void cell_block(const SmallLoopArgs& p, int blk, int tiles_m, const float* h_in,
                                           float* h_out, int t, char* smem) {
  float (*Zs)[32][33] = reinterpret_cast<float (*)[32][33]>(smem);            // [16][32][33]
  float (*Z2)[33] = reinterpret_cast<float (*)[33]>(smem + 16 * 32 * 33 * 4);  // [32][33]
  const int tile_m = blk % tiles_m, tile_j = blk / tiles_m;
  const int m0 = tile_m * 32, j0 = tile_j * 8;
  const int H4 = 4 * p.H;
  const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int gm = min(m0 + l31, p.B - 1);  // clamped rows are computed, never stored
  const int col = (l31 >> 3) * p.H + j0 + (l31 & 7);
  const int k_begin = q * ((p.C + p.H) / 16);
  float4 a[NG];
  float b[NG][4];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int k = k_begin + 8 * g + 4 * half;
    const bool in_ctx = k < p.C;  // C % 4 == 0: a 4-k slice never straddles the two sources
    const float* ap = in_ctx ? p.ctx + (size_t)gm * p.C + k : h_in + (size_t)gm * p.H + (k - p.C);
    a[g] = ld_dev4(ap);
    const float* bp = (in_ctx ? p.cell_kernel + (size_t)k * H4 : p.cell_recurrent + (size_t)(k - p.C) * H4) + col;
#pragma unroll
    for (int s = 0; s < 4; ++s) b[g][s] = bp[(size_t)s * H4];
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].x, b[g][0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].y, b[g][1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].z, b[g][2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].w, b[g][3], acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) Zs[q][(r & 3) + 8 * (r >> 2) + 4 * half][l31] = acc[r];
  __syncthreads();
  {
    const int row = tid >> 5, cc = tid & 31;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += Zs[i][row][cc];
    Z2[row][cc] = sum;
  }
  __syncthreads();
  if (tid < 256) {
    const int row = tid >> 3, u = tid & 7;
    const int m = m0 + row, j = j0 + u;
    if (m < p.B) {
      int id = t == 0 ? p.start_id : ld_dev(p.hist_ids + (size_t)(t - 1) * p.B + m);
      const bool tok = id >= 0 && id < p.V;
      id = tok ? id : 0;
      float z[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float e = p.emb_proj[(size_t)id * H4 + g * p.H + j];
        z[g] = (Z2[row][g * 8 + u] + (tok ? e : 0.f)) + p.cell_bias[g * p.H + j];
      }
      const size_t o = (size_t)m * p.H + j;
      const float c_new = fast_sigmoid(z[1]) * ld_dev(p.c + o) + fast_sigmoid(z[0]) * fast_tanh(z[2]);
      st_dev(p.c + o, c_new);
      st_dev(h_out + o, fast_sigmoid(z[3]) * fast_tanh(c_new));
    }
  }
  __syncthreads();  // (the next block of this workgroup reuses the LDS)
}

// ---- phase QV: y[M,N] = x[M,K] . w[K,N] + bias, one block = 32 rows x 32 columns ----
template <int NG>
__device__ __forceinline__ void dense_block(const float* x, const float* __restrict__ w,
                                            const float* __restrict__ bias, float* y, int M,
                                            int K, int N, int blk, int tiles_m, char* smem) {
  float (*Zs)[32][33] = reinterpret_cast<float (*)[32][33]>(smem);
  const int tile_m = blk % tiles_m, tile_n = blk / tiles_m;
  const int m0 = tile_m * 32, n0 = tile_n * 32;
  const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int gm = min(m0 + l31, M - 1);
  const int col = min(n0 + l31, N - 1);  // clamped rows / columns are computed, never stored
  const int k_begin = q * (K / 16);
  float4 a[NG];
  float b[NG][4];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int k = k_begin + 8 * g + 4 * half;
    a[g] = ld_dev4(x + (size_t)gm * K + k);
    const float* bp = w + (size_t)k * N + col;
#pragma unroll
    for (int s = 0; s < 4; ++s) b[g][s] = bp[(size_t)s * N];
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].x, b[g][0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].y, b[g][1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].z, b[g][2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].w, b[g][3], acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) Zs[q][(r & 3) + 8 * (r >> 2) + 4 * half][l31] = acc[r];
  __syncthreads();
  {
    const int row = tid >> 5, cc = tid & 31;
    const int m = m0 + row, n = n0 + cc;
    if (m < M && n < N) {
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) sum += Zs[i][row][cc];
      st_dev(y + (size_t)m * N + n, sum + (bias ? bias[n] : 0.f));
    }
  }
  __syncthreads();
}

// ---- phase AB, beam half: the greedy step of one row (beam_search.py:360-543 at beam width 1:
// log-softmax, finished-row mask, length penalty with its own value for the eos column, first
// maximum, next state).  The row is staged in LDS; every logit that could tie with the best one
// after rounding gets the reference's exact score arithmetic, ties go to the lower column. ----
struct Cand {
  float s, t;
  int i;
};
__device__ __forceinline__ bool cand_better(float s, int i, float bs, int bi) {
  return (bi < 0) ? (i >= 0) : (i >= 0 && (s > bs || (s == bs && i < bi)));
}
__device__ __forceinline__ Cand wave_best(Cand c) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    Cand o;
    o.s = __shfl_xor(c.s, d);
    o.t = __shfl_xor(c.t, d);
    o.i = __shfl_xor(c.i, d);
    if (cand_better(o.s, o.i, c.s, c.i)) c = o;
  }
  return c;
}

__device__ __forceinline__ void greedy_row(const SmallLoopArgs& p, int b, int t, char* smem) {
  const int V = p.V;
  float* xs = reinterpret_cast<float*>(smem);             // [V] (V <= 16384)
  float* s_red = xs + 16384;                              // [3][16]
  int* s_int = reinterpret_cast<int*>(s_red + 48);        // [16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t o = (size_t)t * p.B + b;
  const float lp = t == 0 ? 0.f : ld_dev(p.hist_lp + (size_t)(t - 1) * p.B + b);
  const int len = ld_dev(p.len + b);
  if (ld_dev(p.fin + b)) {
    // log-prob 0 at eos and f32.min elsewhere: only columns 0, 1 and eos can win
    if (tid == 0) {
      const float pen = penalty_of(p.pen, len);
      float bs = 0.f, bt = 0.f;
      int bi = -1;
      const int ncol = min(V, 2);
      for (int j = 0; j <= ncol; ++j) {
        int col = -1;
        if (j < ncol) col = j;
        else if (p.eos >= ncol && p.eos < V) col = p.eos;
        if (col < 0) continue;
        const float total = lp + ((col == p.eos) ? 0.f : kF32Min);
        const float sc = total / pen;
        if (cand_better(sc, col, bs, bi)) { bs = sc; bt = total; bi = col; }
      }
      p.hist_scores[o] = bs;
      st_dev(p.hist_ids + o, bi < 0 ? 0 : bi);
      p.hist_parents[o] = 0;
      st_dev(p.hist_lp + o, bt);
      // (finished stays set, the length stays)
    }
    return;
  }
  const float* row = p.logits + (size_t)b * V;
  // statistics: max, sum of exp, and the best logit outside the eos column
  float m = -INFINITY, lm = -INFINITY;
  for (int c = tid * 4; c < V; c += SL_NT * 4) {
    const float4 v = ld_dev4(row + c);
    *reinterpret_cast<float4*>(xs + c) = v;
    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      m = fmaxf(m, e[k]);
      if (c + k != p.eos) lm = fmaxf(lm, e[k]);
    }
  }
  __syncthreads();
  float sm = 0.f;
  {
    const float mm = (m > -INFINITY) ? m : 0.f;
    for (int c = tid * 4; c < V; c += SL_NT * 4) {
      const float4 v = *reinterpret_cast<const float4*>(xs + c);
      sm += (fast_exp(v.x - mm) + fast_exp(v.y - mm)) + (fast_exp(v.z - mm) + fast_exp(v.w - mm));
    }
  }
  {
    const float mw = wave_max(m);
    const float sw = wave_sum((m > -INFINITY) ? sm * fast_exp(m - mw) : 0.f);
    const float lw = wave_max(lm);
    if (lane == 0) { s_red[wave] = mw; s_red[16 + wave] = sw; s_red[32 + wave] = lw; }
  }
  __syncthreads();
  float mx = s_red[0], bound_x = s_red[32];
#pragma unroll
  for (int i = 1; i < 16; ++i) { mx = fmaxf(mx, s_red[i]); bound_x = fmaxf(bound_x, s_red[32 + i]); }
  float tot = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) tot += (s_red[i] > -INFINITY) ? s_red[16 + i] * fast_exp(s_red[i] - mx) : 0.f;
  const float lse = logf(tot);
  const float pen_eos = penalty_of(p.pen, len), pen_oth = penalty_of(p.pen, len + 1);
  auto total_of = [&](float x) -> float { return lp + ((x - mx) - lse); };
  // logits within a few ulps of the best one can collide with it after the additions and the division
  float thr_x;
  {
    const float bound_t = total_of(bound_x);
    thr_x = bound_x - ((fabsf(bound_t) + fabsf(bound_x)) * 9.6e-7f + 1e-37f);
    if (!(thr_x == thr_x)) thr_x = -INFINITY;
  }
  Cand best{0.f, 0.f, -1};
  for (int c = tid * 4; c < V; c += SL_NT * 4) {
    const float4 v = *reinterpret_cast<const float4*>(xs + c);
    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int col = c + k;
      if (e[k] >= thr_x || col == p.eos) {
        const float tt = total_of(e[k]);
        const float sc = tt / ((col == p.eos) ? pen_eos : pen_oth);
        if (cand_better(sc, col, best.s, best.i)) { best.s = sc; best.t = tt; best.i = col; }
      }
    }
  }
  best = wave_best(best);
  __syncthreads();  // (s_red is read above by every thread)
  if (lane == 0) { s_red[wave] = best.s; s_red[16 + wave] = best.t; s_int[wave] = best.i; }
  __syncthreads();
  if (tid == 0) {
    Cand w{s_red[0], s_red[16], s_int[0]};
    for (int i = 1; i < 16; ++i)
      if (cand_better(s_red[i], s_int[i], w.s, w.i)) { w.s = s_red[i]; w.t = s_red[16 + i]; w.i = s_int[i]; }
    const int word = w.i < 0 ? 0 : w.i;
    const int nfin = (word == p.eos) ? 1 : 0;
    p.hist_scores[o] = w.i < 0 ? -INFINITY : w.s;
    st_dev(p.hist_ids + o, word);
    p.hist_parents[o] = 0;
    st_dev(p.hist_lp + o, w.t);
    st_dev(p.fin + b, (uint8_t)nfin);
    st_dev(p.len + b, len + ((word != p.eos && !nfin) ? 1 : 0));
  }
  __syncthreads();  // (whatever runs next in this workgroup reuses the LDS)
}

template <int NG_CELL, int NG_DENSE, int UCH>
__global__ __launch_bounds__(SL_NT, 1) void small_loop_kernel(SmallLoopArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[SL_LDS_BYTES];
  int* s_flag = reinterpret_cast<int*>(smem + SL_LDS_BYTES - 16);
  const int nb = gridDim.x, bid = blockIdx.x;
  const int tiles_m = (p.B + 31) / 32;
  const int q_blocks = tiles_m * (p.U / 32), v_blocks = tiles_m * ((p.V + 31) / 32);
  const int l_blocks = tiles_m * (p.H / 8);
  unsigned epoch = 0;
  AttArgs a{p.pf, p.features, p.q, p.att_v, p.att_bv, nullptr, p.c, p.fc3_w, p.fc3_b, p.ctx, nullptr,
            p.B, 1, 1, 1, p.L, p.U, p.D, p.H, nb, nullptr, nullptr, nullptr};
  a.epf = p.epf;
  a.pf_extreme = p.pf_extreme;
  // the attention of step 0: its query first
  ICAP_SL_STAMP(0)
  for (int blk = bid; blk < q_blocks; blk += nb)
    dense_block<NG_DENSE>(p.c, p.w2, p.b2, p.q, p.B, p.H, p.U, blk, tiles_m, smem);
  ICAP_SL_STAMP(1)
  if (!grid_barrier(p, epoch, s_flag)) return;
  ICAP_SL_STAMP(2)
  if (bid < p.B) attention_body<SL_NT, 1, UCH, true, ATT_ACT_TANH, true, false, true>(a, smem, bid, nb);
  ICAP_SL_STAMP(3)
  if (!grid_barrier(p, epoch, s_flag)) return;
  ICAP_SL_STAMP(4)
  float* h_in = p.h0;
  float* h_out = p.h1;
  for (int t = 0; t < p.T; ++t) {
    for (int blk = bid; blk < l_blocks; blk += nb) cell_block<NG_CELL>(p, blk, tiles_m, h_in, h_out, t, smem);
    ICAP_SL_STAMP(8 + 6 * t + 0)
    if (!grid_barrier(p, epoch, s_flag)) return;
    ICAP_SL_STAMP(8 + 6 * t + 1)
    const bool more = t + 1 < p.T;
    // (the few query blocks first in the walk: the workgroups that get them also get vocab blocks)
    for (int blk = bid; blk < (more ? q_blocks : 0) + v_blocks; blk += nb) {
      const int qb = more ? q_blocks : 0;
      if (blk < qb) dense_block<NG_DENSE>(p.c, p.w2, p.b2, p.q, p.B, p.H, p.U, blk, tiles_m, smem);
      else dense_block<NG_DENSE>(h_out, p.wout, p.bout, p.logits, p.B, p.H, p.V, blk - qb, tiles_m, smem);
    }
    ICAP_SL_STAMP(8 + 6 * t + 2)
    if (!grid_barrier(p, epoch, s_flag)) return;
    ICAP_SL_STAMP(8 + 6 * t + 3)
    // rows from the top of the grid, images from the bottom: different compute units while B <= nb / 2
    for (int b = bid; b < p.B; b += nb) greedy_row(p, b, t, smem);
    if (more) {
      const int rb = nb - 1 - bid;
      if (rb < p.B) attention_body<SL_NT, 1, UCH, true, ATT_ACT_TANH, true, false, true>(a, smem, rb, nb);
    }
    ICAP_SL_STAMP(8 + 6 * t + 4)
    if (!grid_barrier(p, epoch, s_flag)) return;
    ICAP_SL_STAMP(8 + 6 * t + 5)
    float* tmp = h_in; h_in = h_out; h_out = tmp;
  }
}

}  // namespace

// The instances built: (K splits of the cell product, of the dense products, 256-wide chunks of U)
static const void* small_loop_instance(int ngc, int ngd, int uch) {
#define ICAP_SL_PICK(GC, GD, UC) \
  if (ngc == GC && ngd == GD && uch == UC) return reinterpret_cast<const void*>(&small_loop_kernel<GC, GD, UC>);
  ICAP_SL_PICK(6, 4, 2)  // the reference's sizes: C = 256, H = U = 512
  ICAP_SL_PICK(2, 1, 1)  // C = H = U = 128
  ICAP_SL_PICK(4, 2, 1)  // C = H = U = 256
#undef ICAP_SL_PICK
  return nullptr;
}

// Shapes the one-launch loop serves: beam width 1 with at most 64 rows, one of the built K
// splits ((C + H) / 128 for the cell product, H / 128 for the dense ones), U % 32 == 0, V % 4 == 0
// up to 16384 (the row is staged in LDS), the attention's LDS within the kernel's.
bool small_loop_supported(int B, int W, int C, int H, int U, int L, int D, int V) {
  if (!(W == 1 && B >= 1 && B <= 64 && C > 0 && H > 0 && U > 0 && L >= 1 && D > 0)) return false;
  if (C % 4 || (C + H) % 128 || H % 128 || U % 32 || D % 4 || V % 4 || V < 4 || V > 16384) return false;
  if (!small_loop_instance((C + H) / 128, H / 128, (U + 255) / 256)) return false;
  return sizeof(float) * attention_lds_floats(SL_NT, 1, 1, L, D) <= (size_t)SL_LDS_BYTES - 64;
}

// The grid: one workgroup per compute unit (a workgroup of 16 waves and 72 KB of LDS is alone
// on its unit).  An ordinary launch, not a cooperative one: the runtime keeps a hardware queue
// of its own for cooperative launches, and on this driver one more queue in the process is
// what stalls event waits between the legacy default stream and other streams later on.
// Without the runtime's guarantee that the whole grid is resident at once the barriers lean
// on the dispatcher placing every workgroup as units fall free - true whenever no other
// resident kernel waits for this one; two of these loops enqueued on different streams of
// one GPU can each hold part of the units and then both run into the bounded wait, which ends
// them with *err raised (-1 tokens, reported by the engine's decode status) instead of a hang.
int small_loop_f32(const SmallLoopArgs& args, int compute_units, hipStream_t s) {
  const SmallLoopArgs& p = args;
  ICAP_REQUIRE(small_loop_supported(p.B, 1, p.C, p.H, p.U, p.L, p.D, p.V), "small_loop_f32: shape not served");
  ICAP_REQUIRE(p.T >= 1 && compute_units >= 1, "small_loop_f32: bad step count or grid");
  ICAP_REQUIRE(aligned16(p.ctx) && aligned16(p.h0) && aligned16(p.h1) && aligned16(p.c) && aligned16(p.q) &&
                   aligned16(p.logits) && aligned16(p.pf) && aligned16(p.features) && aligned16(p.att_v) &&
                   (!p.epf || aligned16(p.epf)),
               "small_loop_f32: operands must be 16-byte aligned");
  const void* fn = small_loop_instance((p.C + p.H) / 128, p.H / 128, (p.U + 255) / 256);
  ICAP_REQUIRE(fn != nullptr, "small_loop_f32: no instance for these sizes");
  SmallLoopArgs copy = args;
  void* kargs[] = {&copy};
  static const char* grid_env = getenv("ICAP_SMALL_LOOP_GRID");  // tuning knob: fewer workgroups
  const int grid = grid_env && atoi(grid_env) > 0 && atoi(grid_env) < compute_units ? atoi(grid_env) : compute_units;
  ICAP_REQUIRE(grid >= p.B, "small_loop_f32: the grid must hold one workgroup per row");
  ICAP_HIP(hipLaunchKernel(fn, dim3(grid), dim3(SL_NT), kargs, 0, s));
  return ICAP_OK;
}

// The engine's view: fills the argument block from its configuration, weights and workspace.
int small_loop_decode(const icap_config& c, const icap_weights& w, const float* wout, const float* bout,
                      const float* emb_proj, float att_v_bias, float fc3_bias, const float* pf,
                      const float* epf, const int* pf_extreme, const float* features, float* h0, float* h1,
                      float* cst, float* q, float* ctx, float* logits, uint8_t* fin, int32_t* len,
                      int32_t* hist_ids, int32_t* hist_parents, float* hist_scores, float* hist_lp,
                      unsigned* bar, int* err, int compute_units, hipStream_t s) {
  SmallLoopArgs a{};
  a.B = c.B; a.C = c.D; a.H = c.H; a.U = c.U; a.L = c.L; a.D = c.D; a.V = c.V; a.T = c.T;
  a.eos = c.eos_id; a.start_id = c.start_id;
  a.cell_kernel = w.lstm_kernel; a.cell_recurrent = w.lstm_recurrent_kernel; a.cell_bias = w.lstm_bias;
  a.emb_proj = emb_proj;
  a.w2 = w.att_w2_kernel; a.b2 = w.att_w2_bias;
  a.att_v = w.att_v_kernel; a.att_bv = att_v_bias;
  a.fc3_w = w.fc3_kernel; a.fc3_b = fc3_bias;
  a.wout = wout; a.bout = bout;
  a.pf = pf; a.epf = epf; a.features = features; a.pf_extreme = pf_extreme;
  a.h0 = h0; a.h1 = h1; a.c = cst; a.q = q; a.ctx = ctx; a.logits = logits;
  a.fin = fin; a.len = len;
  a.hist_ids = hist_ids; a.hist_parents = hist_parents; a.hist_scores = hist_scores; a.hist_lp = hist_lp;
  a.pen = make_penalty(c.length_penalty_weight);
  a.bar = bar; a.err = err;
  return small_loop_f32(a, compute_units, s);
}

}  // namespace icap

#ifdef ICAP_TIMELINE
extern "C" int icap_debug_small_loop_stamps(long long* out, int n) {
  if (n > 4096) n = 4096;
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(icap::icap_sl_stamps), (size_t)n * sizeof(long long));
}
#endif
