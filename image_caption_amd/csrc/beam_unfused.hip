// Unfused pieces of the beam-search step, for callers that bring their own successor
// choice (BeamSearchConfig.choose_successors_fn) or call the helper functions directly:
//
//   beam_scores_f32    log-softmax + finished mask + running total + length-penalised
//                      score for every (image, beam, word): materialises the [B,W,V]
//                      `scores` and `total` tensors of beam_search.py:396-445.
//   beam_gather_f32    everything after the successor choice (beam_search.py:475-529):
//                      word / parent split, log-prob gather, finished and length update.
//   mask_probs_f32, length_penalty_f32, hyp_score_f32, batch_gather_f32
//                      the stand-alone helpers of beam_search.py:115-141,165-197,347-358.
//   logit_masks_f32    the additive start / forced-eos / forced-pad logit masks of the
//                      baseline beam searcher (utils/agent_utils.py:188-237), in place.
//   expand_rows_i32    per-image int32 values to beam rows (tf.tile of the target lengths,
//                      utils/agent_utils.py:183-184).
//
// All of it is streaming element-wise work: one 16-byte access per lane where the row
// layout allows it, two reads of a logit row at most (the second one is L2-hot).
#include "beam_shared.h"
#include "icap_common.h"
#include "topk_epilogue.h"

namespace icap {
namespace {

constexpr int NT = 256;

struct ScoreArgs {
  const float* logits;
  const float* log_probs;
  const uint8_t* finished;
  const int32_t* lengths;
  float* scores;
  float* totals;
  int eos, input_is_log_prob, V;
  Penalty pen;
};

// One workgroup per (image, beam row).
template <bool VEC>
__global__ __launch_bounds__(NT) void beam_scores_kernel(ScoreArgs p) {
  __shared__ float s_m[NT / 64], s_s[NT / 64];
  const int V = p.V;
  const size_t rw = blockIdx.x;
  const int tid = threadIdx.x;
  const float* row = p.logits + rw * V;
  float* out_s = p.scores + rw * V;
  float* out_t = p.totals + rw * V;
  const bool fin = p.finished[rw] != 0;
  const float lp = p.log_probs[rw];
  const int len = p.lengths[rw];

  float mx = 0.f, lse = 0.f;
  if (!fin && !p.input_is_log_prob) {
    // streaming (max, sum exp) per thread, then wave and workgroup reductions
    float m = -INFINITY, sm = 0.f;
    if (VEC) {
      for (int c = tid * 4; c < V; c += NT * 4) {
        const float4 v = *reinterpret_cast<const float4*>(row + c);
        const float mn = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
        if (mn > -INFINITY)
          sm = sm * fast_exp(m - mn) + (fast_exp(v.x - mn) + fast_exp(v.y - mn)) +
               (fast_exp(v.z - mn) + fast_exp(v.w - mn));
        m = mn;
      }
    } else {
      for (int c = tid; c < V; c += NT) {
        const float v = row[c];
        const float mn = fmaxf(m, v);
        if (mn > -INFINITY) sm = sm * fast_exp(m - mn) + fast_exp(v - mn);
        m = mn;
      }
    }
    const float mw = wave_max(m);
    const float sw = wave_sum((m > -INFINITY) ? sm * fast_exp(m - mw) : 0.f);
    if ((tid & 63) == 0) { s_m[tid >> 6] = mw; s_s[tid >> 6] = sw; }
    __syncthreads();
    mx = s_m[0];
#pragma unroll
    for (int i = 1; i < NT / 64; ++i) mx = fmaxf(mx, s_m[i]);
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) tot += (s_m[i] > -INFINITY) ? s_s[i] * fast_exp(s_m[i] - mx) : 0.f;
    lse = logf(tot);
  }
  // A finished beam keeps its length for every word; a live one grows by one except at eos
  // (beam_search.py:416-437).
  const float pen_eos = penalty_of(p.pen, len);
  const float pen_oth = fin ? pen_eos : penalty_of(p.pen, len + 1);
  auto one = [&](float x, int c, float& t, float& s) {
    float pr;
    if (fin) pr = (c == p.eos) ? 0.f : kF32Min;
    else pr = p.input_is_log_prob ? x : (x - mx) - lse;
    t = lp + pr;
    s = t / ((c == p.eos) ? pen_eos : pen_oth);
  };
  if (VEC) {
    for (int c = tid * 4; c < V; c += NT * 4) {
      const float4 v = *reinterpret_cast<const float4*>(row + c);
      float4 t, s;
      one(v.x, c, t.x, s.x);
      one(v.y, c + 1, t.y, s.y);
      one(v.z, c + 2, t.z, s.z);
      one(v.w, c + 3, t.w, s.w);
      *reinterpret_cast<float4*>(out_t + c) = t;
      *reinterpret_cast<float4*>(out_s + c) = s;
    }
  } else {
    for (int c = tid; c < V; c += NT) {
      float t, s;
      one(row[c], c, t, s);
      out_t[c] = t;
      out_s[c] = s;
    }
  }
}

// One thread per chosen successor.
__global__ __launch_bounds__(NT) void beam_gather_kernel(
    const float* __restrict__ totals, const uint8_t* __restrict__ finished,
    const int32_t* __restrict__ lengths, const int32_t* __restrict__ word_indices, int eos,
    int32_t* __restrict__ ids, int32_t* __restrict__ parents, float* __restrict__ next_lp,
    uint8_t* __restrict__ next_fin, int32_t* __restrict__ next_len, int B, int W, int V, int K) {
  const int o = blockIdx.x * NT + threadIdx.x;
  if (o >= B * K) return;
  const int b = o / K;
  // An index outside [0, W*V) is not an error here: like the gathers of the reference on a
  // GPU (tf.gather stores 0 for an out-of-range index) it yields log-prob 0 and, where the
  // parent beam is out of range too, an unfinished parent of length 0; word and parent are
  // the floor quotient / remainder the reference computes from the raw index.
  const long long flat = word_indices[o];
  const bool in_range = flat >= 0 && flat < (long long)W * V;
  long long parent_l = flat / V, word_l = flat % V;
  if (word_l < 0) { word_l += V; parent_l -= 1; }
  const int word = (int)word_l;
  const int parent = (int)parent_l;
  const bool parent_ok = parent_l >= 0 && parent_l < W;
  const size_t rp = (size_t)b * W + (parent_ok ? parent : 0);
  const int pfin = (parent_ok && finished[rp]) ? 1 : 0;
  const int nfin = (pfin || word == eos) ? 1 : 0;
  ids[o] = word;
  parents[o] = parent;
  next_lp[o] = in_range ? totals[(size_t)b * W * V + flat] : 0.f;
  next_fin[o] = (uint8_t)nfin;
  next_len[o] = (parent_ok ? lengths[rp] : 0) + ((word != eos && !nfin) ? 1 : 0);
}

// (1 - fm) * finished_row + fm * probs with fm = 1 - finished, as floats like the reference
// (a finished row therefore turns an infinite input into NaN exactly as it does there).
__global__ __launch_bounds__(NT) void mask_probs_kernel(const float* __restrict__ probs,
                                                        const uint8_t* __restrict__ finished,
                                                        float* __restrict__ out, int eos,
                                                        long long total, int V) {
  const long long i = (long long)blockIdx.x * NT + threadIdx.x;
  if (i >= total) return;
  const long long r = i / V;
  const int c = (int)(i - r * V);
  const float fm = 1.0f - (finished[r] ? 1.0f : 0.0f);
  const float rowv = (c == eos) ? 0.f : kF32Min;
  out[i] = (1.0f - fm) * rowv + fm * probs[i];
}

__global__ __launch_bounds__(NT) void hyp_score_kernel(const float* __restrict__ log_probs,
                                                       const int32_t* __restrict__ lengths,
                                                       float* __restrict__ out, long long n,
                                                       Penalty pen) {
  const long long i = (long long)blockIdx.x * NT + threadIdx.x;
  if (i >= n) return;
  const float pn = penalty_of(pen, lengths[i]);
  out[i] = log_probs ? log_probs[i] / pn : pn;
}

__global__ __launch_bounds__(NT) void batch_gather_kernel(const float* __restrict__ a,
                                                          const int32_t* __restrict__ idx,
                                                          float* __restrict__ out, int B, int X,
                                                          int Y) {
  const int o = blockIdx.x * NT + threadIdx.x;
  if (o >= B * Y) return;
  const int b = o / Y;
  const int j = idx[o];
  // (an out-of-range index stores 0, as tf.gather does on a GPU)
  out[o] = (j >= 0 && j < X) ? a[(size_t)b * X + j] : 0.f;
}

// One thread per four consecutive logits of a row (V % 4 == 0) or per logit.
template <bool VEC>
__global__ __launch_bounds__(NT) void logit_masks_kernel(float* __restrict__ logits, int N, int V,
                                                         LogitMasks mk) {
  constexpr int E = VEC ? 4 : 1;
  const long long per_row = V / E;
  const long long i = (long long)blockIdx.x * NT + threadIdx.x;
  if (i >= per_row * N) return;
  const int row = (int)(i / per_row);
  const int c0 = (int)(i - row * per_row) * E;
  float* q = logits + (size_t)row * V + c0;
  float x[E];
  if constexpr (VEC) {
    const float4 v = *reinterpret_cast<const float4*>(q);
    x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
  } else {
    x[0] = *q;
  }
  apply_logit_masks<E>(x, c0, row, N, mk);
  if constexpr (VEC)
    *reinterpret_cast<float4*>(q) = make_float4(x[0], x[1], x[2], x[3]);
  else
    *q = x[0];
}

__global__ __launch_bounds__(NT) void expand_rows_i32_kernel(const int32_t* __restrict__ src,
                                                             int32_t* __restrict__ dst, int B, int W,
                                                             int tile_mode) {
  const int n = blockIdx.x * NT + threadIdx.x;
  if (n >= B * W) return;
  dst[n] = src[tile_mode == ICAP_TILE_REPEAT ? n / W : n % B];
}

// One workgroup per row: a sample from softmax(logits + token masks) by inversion of the
// cumulative distribution at the caller's uniform number u[row] in [0,1), then the reference's
// forcing by position (eos where step + 1 == length, pad where step >= length).
//   pass 1  row maximum and the exp-sums of the NT contiguous column chunks (one per thread);
//   scan    exclusive prefix of the chunk sums (wave scans + one LDS round);
//   pass 2  the thread whose chunk holds the target mass u * total walks it.
// Masked columns carry f32.min: their exp is exactly 0, they are never drawn.
struct SampleArgs {
  const float* logits;   // [N,V]
  const float* u;        // [N]
  const int32_t* row_len;  // [N] or null
  int32_t* out;          // [N]
  int V, step, start, eos, pad;
};

__global__ __launch_bounds__(NT) void sample_categorical_kernel(SampleArgs p) {
  __shared__ float s_red[NT / 64];
  __shared__ float s_pre[NT / 64];
  __shared__ int s_pick;
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* x = p.logits + (size_t)row * p.V;
  const int chunk = (p.V + NT - 1) / NT;
  const int c_lo = min(tid * chunk, p.V), c_hi = min(c_lo + chunk, p.V);
  auto masked = [&](int c) -> float {
    float v = x[c];
    if (c == p.start) v += kF32Min;
    if (c == p.eos) v += kF32Min;
    if (c == p.pad) v += kF32Min;
    return v;
  };
  float m = -INFINITY;
  for (int c = c_lo; c < c_hi; ++c) m = fmaxf(m, masked(c));
  m = wave_max(m);
  if (lane == 0) s_red[wave] = m;
  if (tid == 0) s_pick = -1;
  __syncthreads();
  m = s_red[0];
#pragma unroll
  for (int w = 1; w < NT / 64; ++w) m = fmaxf(m, s_red[w]);
  __syncthreads();
  float mine = 0.f;
  for (int c = c_lo; c < c_hi; ++c) mine += expf(masked(c) - m);
  // inclusive scan of the chunk sums over the workgroup
  float inc = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  if (lane == 63) s_red[wave] = inc;
  __syncthreads();
  if (tid == 0) {
    float run = 0.f;
    for (int w = 0; w < NT / 64; ++w) { s_pre[w] = run; run += s_red[w]; }
  }
  __syncthreads();
  // [before, upto): this thread's share of the cumulative mass; a thread's `upto` is the next
  // thread's `before` bit for bit, so the shares tile [0, total) without gaps
  const float prev = __shfl_up(inc, 1, 64);
  const float before = s_pre[wave] + (lane == 0 ? 0.f : prev);
  const float upto = s_pre[wave] + inc;
  const float total = s_pre[NT / 64 - 1] + s_red[NT / 64 - 1];
  const float target = p.u[row] * total;
  if (target >= before && target < upto && c_hi > c_lo) {
    float run = before;
    int pick = -1;
    for (int c = c_lo; c < c_hi; ++c) {
      const float e = expf(masked(c) - m);
      run += e;
      if (e > 0.f) {
        pick = c;                 // (the last column with mass, should rounding overshoot)
        if (target < run) break;
      }
    }
    if (pick >= 0) atomicMax(&s_pick, pick);
  }
  __syncthreads();
  if (tid == 0) {
    int sample = s_pick;
    if (sample < 0) {  // rounding left the target beyond the last chunk: the last column with mass
      for (int c = p.V - 1; c >= 0 && sample < 0; --c)
        if (expf(masked(c) - m) > 0.f) sample = c;
      if (sample < 0) sample = 0;
    }
    if (p.row_len) {
      const int len = p.row_len[row];
      if (p.step + 1 == len) sample = p.eos;
      if (p.step >= len) sample = p.pad;
    }
    p.out[row] = sample;
  }
}

}  // namespace

int logit_masks_f32(float* logits, int N, int V, const LogitMasks& mk, hipStream_t s) {
  ICAP_REQUIRE(logits, "logit_masks_f32: null pointer");
  ICAP_REQUIRE(N > 0 && V > 0, "logit_masks_f32: bad shape");
  ICAP_REQUIRE(mk.start_col < V && mk.eos < V && mk.pad < V, "logit_masks_f32: token outside the vocabulary");
  ICAP_REQUIRE(!mk.row_len || (mk.eos >= 0 && mk.pad >= 0),
               "logit_masks_f32: eos and pad tokens are required with row lengths");
  if (!mk.any()) return ICAP_OK;
  const bool vec = (V % 4 == 0) && aligned16(logits);
  const long long threads = (long long)N * (vec ? V / 4 : V);
  ICAP_REQUIRE(threads / NT < (1LL << 31), "logit_masks_f32: too many elements");
  const dim3 grid((unsigned)((threads + NT - 1) / NT));
  if (vec)
    hipLaunchKernelGGL(logit_masks_kernel<true>, grid, dim3(NT), 0, s, logits, N, V, mk);
  else
    hipLaunchKernelGGL(logit_masks_kernel<false>, grid, dim3(NT), 0, s, logits, N, V, mk);
  ICAP_LAUNCH_CHECK("logit_masks_kernel");
  return ICAP_OK;
}

// sample[n] ~ softmax(logits[n] + f32.min at the start / eos / pad columns), drawn by inversion
// at u[n]; with row_len: eos where step + 1 == row_len[n], pad where step >= row_len[n].
int sample_categorical_f32(const float* logits, const float* u, const int32_t* row_len, int step,
                           int start_token, int eos_token, int pad_token, int32_t* samples, int N,
                           int V, hipStream_t s) {
  ICAP_REQUIRE(logits && u && samples, "sample_categorical_f32: null pointer");
  ICAP_REQUIRE(N > 0 && V > 0, "sample_categorical_f32: bad shape");
  ICAP_REQUIRE(start_token < V && eos_token < V && pad_token < V,
               "sample_categorical_f32: token outside the vocabulary");
  ICAP_REQUIRE(!row_len || (eos_token >= 0 && pad_token >= 0),
               "sample_categorical_f32: eos and pad tokens are required with row lengths");
  SampleArgs a{logits, u, row_len, samples, V, step, start_token, eos_token, pad_token};
  hipLaunchKernelGGL(sample_categorical_kernel, dim3(N), dim3(NT), 0, s, a);
  ICAP_LAUNCH_CHECK("sample_categorical_kernel");
  return ICAP_OK;
}

int expand_rows_i32(const int32_t* src, int32_t* dst, int B, int W, int tile_mode, hipStream_t s) {
  ICAP_REQUIRE(src && dst, "expand_rows_i32: null pointer");
  ICAP_REQUIRE(B > 0 && W > 0, "expand_rows_i32: bad shape");
  ICAP_REQUIRE(tile_mode == ICAP_TILE_REFERENCE || tile_mode == ICAP_TILE_REPEAT,
               "expand_rows_i32: unknown tile_mode %d", tile_mode);
  hipLaunchKernelGGL(expand_rows_i32_kernel, dim3((B * W + NT - 1) / NT), dim3(NT), 0, s, src, dst, B,
                     W, tile_mode);
  ICAP_LAUNCH_CHECK("expand_rows_i32_kernel");
  return ICAP_OK;
}

int beam_scores_f32(const float* logits, const float* log_probs, const uint8_t* finished,
                    const int32_t* lengths, int eos_token, double length_penalty_weight,
                    int input_is_log_prob, float* scores, float* totals, int B, int W, int V,
                    hipStream_t s) {
  ICAP_REQUIRE(logits && log_probs && finished && lengths && scores && totals,
               "beam_scores_f32: null pointer");
  ICAP_REQUIRE(B > 0 && W > 0 && V > 0, "beam_scores_f32: bad shape");
  ICAP_REQUIRE((long long)W * V < (1LL << 31), "beam_scores_f32: W*V overflows int32");
  ScoreArgs a;
  a.logits = logits; a.log_probs = log_probs; a.finished = finished; a.lengths = lengths;
  a.scores = scores; a.totals = totals;
  a.eos = eos_token; a.input_is_log_prob = input_is_log_prob; a.V = V;
  a.pen = make_penalty(length_penalty_weight);
  const bool vec = (V % 4 == 0) && aligned16(logits) && aligned16(scores) && aligned16(totals);
  const dim3 grid((unsigned)((size_t)B * W)), block(NT);
  if (vec) hipLaunchKernelGGL(beam_scores_kernel<true>, grid, block, 0, s, a);
  else hipLaunchKernelGGL(beam_scores_kernel<false>, grid, block, 0, s, a);
  ICAP_LAUNCH_CHECK("beam_scores_kernel");
  return ICAP_OK;
}

int beam_gather_f32(const float* totals, const uint8_t* finished, const int32_t* lengths,
                    const int32_t* word_indices, int eos_token, int32_t* predicted_ids,
                    int32_t* parent_ids, float* next_log_probs, uint8_t* next_finished,
                    int32_t* next_lengths, int B, int W, int V, int K, hipStream_t s) {
  ICAP_REQUIRE(totals && finished && lengths && word_indices, "beam_gather_f32: null input");
  ICAP_REQUIRE(predicted_ids && parent_ids && next_log_probs && next_finished && next_lengths,
               "beam_gather_f32: null output");
  ICAP_REQUIRE(B > 0 && W > 0 && V > 0 && K > 0, "beam_gather_f32: bad shape");
  ICAP_REQUIRE((long long)W * V < (1LL << 31), "beam_gather_f32: W*V overflows int32");
  ICAP_REQUIRE((long long)B * K < (1LL << 31), "beam_gather_f32: B*K overflows int32");
  hipLaunchKernelGGL(beam_gather_kernel, dim3(ceil_div(B * K, NT)), dim3(NT), 0, s, totals,
                     finished, lengths, word_indices, eos_token, predicted_ids, parent_ids,
                     next_log_probs, next_finished, next_lengths, B, W, V, K);
  ICAP_LAUNCH_CHECK("beam_gather_kernel");
  return ICAP_OK;
}

int mask_probs_f32(const float* probs, const uint8_t* finished, float* out, int eos_token, int N,
                   int V, hipStream_t s) {
  ICAP_REQUIRE(probs && finished && out, "mask_probs_f32: null pointer");
  ICAP_REQUIRE(N > 0 && V > 0, "mask_probs_f32: bad shape");
  const long long total = (long long)N * V;
  ICAP_REQUIRE(total / NT < (1LL << 31), "mask_probs_f32: too many elements");
  hipLaunchKernelGGL(mask_probs_kernel, dim3((unsigned)((total + NT - 1) / NT)), dim3(NT), 0, s,
                     probs, finished, out, eos_token, total, V);
  ICAP_LAUNCH_CHECK("mask_probs_kernel");
  return ICAP_OK;
}

int hyp_score_f32(const float* log_probs, const int32_t* lengths, double length_penalty_weight,
                  float* out, long long n, hipStream_t s) {
  ICAP_REQUIRE(lengths && out, "hyp_score_f32: null pointer");
  ICAP_REQUIRE(n > 0 && n / NT < (1LL << 31), "hyp_score_f32: bad element count");
  hipLaunchKernelGGL(hyp_score_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, s,
                     log_probs, lengths, out, n, make_penalty(length_penalty_weight));
  ICAP_LAUNCH_CHECK("hyp_score_kernel");
  return ICAP_OK;
}

int batch_gather_f32(const float* a, const int32_t* idx, float* out, int B, int X, int Y,
                     hipStream_t s) {
  ICAP_REQUIRE(a && idx && out, "batch_gather_f32: null pointer");
  ICAP_REQUIRE(B > 0 && X > 0 && Y > 0, "batch_gather_f32: bad shape");
  ICAP_REQUIRE((long long)B * Y < (1LL << 31), "batch_gather_f32: B*Y overflows int32");
  hipLaunchKernelGGL(batch_gather_kernel, dim3(ceil_div(B * Y, NT)), dim3(NT), 0, s, a, idx, out,
                     B, X, Y);
  ICAP_LAUNCH_CHECK("batch_gather_kernel");
  return ICAP_OK;
}

}  // namespace icap
