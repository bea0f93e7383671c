// Fused beam-search step (log-softmax + finished mask + length penalty + top-W over
// W*V candidates + gathers) and back-tracking.
//
// beam_step_f32: one workgroup per image.
//   pass A  per unfinished beam row: online (max, sum exp) over the V logits; when the
//           image's W*V logits fit in LDS they are staged there so HBM is read once.
//   pass C  every wave scans its share of the rows in increasing flat index and keeps
//           a wave-uniform, register-resident list of its best WL candidates
//           (score, total log-prob, flat index).  A cheap conservative pre-filter on
//           the un-normalised log-prob skips the IEEE division for almost every
//           element; survivors are scored exactly as the reference does
//           ((lp + ((x - max) - lse)) / penalty) and enter the list best-first by
//           wave-wide argmax rounds.
//   merge   the wave lists meet in LDS; wave 0 extracts the W winners in order
//           (score descending, lower flat index first on ties = tf.nn.top_k) and
//           writes the outputs and the next beam state.
#include <math.h>

#include "icap_common.h"

namespace icap {
namespace {

constexpr int PEN_N = 64;  // host-computed length-penalty table entries (len 0..63)
constexpr int EMPTY_IDX = 0x7fffffff;

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
  float lpw;
  float pen_den;
  float pen_table[PEN_N];
};

__device__ __forceinline__ float penalty_of(const BeamArgs& p, int len) {
  if (len >= 0 && len < PEN_N) return p.pen_table[len];
  return powf(5.0f + (float)len, p.lpw) / p.pen_den;
}

template <int WL>
struct CandList {
  float s[WL];
  float t[WL];
  int i[WL];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int j = 0; j < WL; ++j) {
      s[j] = -INFINITY;
      t[j] = 0.f;
      i[j] = EMPTY_IDX;
    }
  }
  // candidate (sc, ii) ranks ahead of entry j: higher score, or equal score and lower
  // flat index (the tf.nn.top_k order); an empty slot ranks behind everything
  __device__ __forceinline__ bool ahead(float sc, int ii, int j) const {
    return i[j] == EMPTY_IDX || sc > s[j] || (sc == s[j] && ii < i[j]);
  }
  __device__ __forceinline__ bool accepts(float sc, int ii) const { return ahead(sc, ii, WL - 1); }
  __device__ __forceinline__ void insert(float sc, float tt, int ii) {
#pragma unroll
    for (int j = WL - 1; j >= 0; --j) {
      const int jm = j > 0 ? j - 1 : 0;
      if (j > 0 && ahead(sc, ii, jm)) {
        s[j] = s[jm];
        t[j] = t[jm];
        i[j] = i[jm];
      } else if (ahead(sc, ii, j)) {
        s[j] = sc;
        t[j] = tt;
        i[j] = ii;
      }
    }
  }
};

__device__ __forceinline__ bool cand_better(float sa, int ia, float sb, int ib) {
  // a real candidate always beats an empty slot; otherwise higher score, then lower index
  if (ib == EMPTY_IDX) return ia != EMPTY_IDX;
  if (ia == EMPTY_IDX) return false;
  return sa > sb || (sa == sb && ia < ib);
}

template <int NT, int WL, bool VEC>
__global__ __launch_bounds__(NT) void beam_step_kernel(BeamArgs p) {
  constexpr int NWV = NT / 64;
  constexpr bool STAGE = (NT == 1024);  // the image's logits are staged in LDS
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // carve (all offsets multiples of 16 bytes)
  float* s_lp = reinterpret_cast<float*>(smem);            // [16]
  float* s_max = s_lp + 16;                                // [16]
  float* s_lse = s_max + 16;                               // [16]
  int* s_len = reinterpret_cast<int*>(s_lse + 16);         // [16]
  int* s_fin = s_len + 16;                                 // [16]
  float* s_red = reinterpret_cast<float*>(s_fin + 16);     // [2][NWV]
  float* s_cs = s_red + 2 * NWV + ((2 * NWV) % 4 ? 4 - (2 * NWV) % 4 : 0);  // [NWV*WL]
  float* s_ct = s_cs + NWV * WL + ((NWV * WL) % 4 ? 4 - (NWV * WL) % 4 : 0);
  int* s_ci = reinterpret_cast<int*>(s_ct + NWV * WL + ((NWV * WL) % 4 ? 4 - (NWV * WL) % 4 : 0));
  float* s_logits = reinterpret_cast<float*>(s_ci + NWV * WL + ((NWV * WL) % 4 ? 4 - (NWV * WL) % 4 : 0));

  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int W = p.W, V = p.V;
  const float* lg = p.logits + (size_t)b * W * V;

  if (tid < W) {
    s_lp[tid] = p.log_probs[(size_t)b * W + tid];
    s_len[tid] = p.lengths[(size_t)b * W + tid];
    s_fin[tid] = p.finished[(size_t)b * W + tid] ? 1 : 0;
  }
  __syncthreads();

  const bool first = p.time <= 0;
  // rows whose candidates are scored
  const int row_lo = first ? p.start_offset : 0;
  const int row_hi = first ? p.start_offset + 1 : W;

  // ---- pass A: per-row (max, lse); stage logits in LDS when they fit ----
  for (int w = 0; w < W; ++w) {
    const bool needed = (w >= row_lo && w < row_hi) || (first && w == 0);
    if (!needed || s_fin[w]) continue;  // block-uniform
    const float* row = lg + (size_t)w * V;
    float* srow = s_logits + (size_t)w * V;
    if (p.input_is_log_prob) {
      if (STAGE) {
        for (int c = tid * 4; c < V; c += NT * 4) {
          if (VEC) {
            *reinterpret_cast<float4*>(srow + c) = *reinterpret_cast<const float4*>(row + c);
          } else {
            for (int e = 0; e < 4 && c + e < V; ++e) srow[c + e] = row[c + e];
          }
        }
      }
      if (tid == 0) { s_max[w] = 0.f; s_lse[w] = 0.f; }
      continue;
    }
    float m = -INFINITY, s = 0.f;
#pragma unroll 4
    for (int c = tid * 4; c < V; c += NT * 4) {
      float x0, x1, x2, x3;
      if (VEC) {
        const float4 x = *reinterpret_cast<const float4*>(row + c);
        x0 = x.x; x1 = x.y; x2 = x.z; x3 = x.w;
        if (STAGE) *reinterpret_cast<float4*>(srow + c) = x;
      } else {
        x0 = row[c];
        x1 = c + 1 < V ? row[c + 1] : -INFINITY;
        x2 = c + 2 < V ? row[c + 2] : -INFINITY;
        x3 = c + 3 < V ? row[c + 3] : -INFINITY;
        if (STAGE) {
          srow[c] = x0;
          if (c + 1 < V) srow[c + 1] = x1;
          if (c + 2 < V) srow[c + 2] = x2;
          if (c + 3 < V) srow[c + 3] = x3;
        }
      }
      const float mn = fmaxf(fmaxf(m, fmaxf(x0, x1)), fmaxf(x2, x3));
      if (mn > -INFINITY) {
        s = s * fast_exp(m - mn) + fast_exp(x0 - mn) + fast_exp(x1 - mn) + fast_exp(x2 - mn) +
            fast_exp(x3 - mn);
        m = mn;
      }
    }
    // wave then block reduction of (m, s)
    const float mw = wave_max(m);
    float sw = (m > -INFINITY) ? s * fast_exp(m - mw) : 0.f;
    sw = wave_sum(sw);
    if (lane == 0) { s_red[wave] = mw; s_red[NWV + wave] = sw; }
    __syncthreads();
    if (tid == 0) {
      float mb = -INFINITY;
      for (int i = 0; i < NWV; ++i) mb = fmaxf(mb, s_red[i]);
      float sb = 0.f;
      for (int i = 0; i < NWV; ++i)
        if (s_red[i] > -INFINITY) sb += s_red[NWV + i] * fast_exp(s_red[i] - mb);
      s_max[w] = mb;
      s_lse[w] = logf(sb);
    }
    __syncthreads();
  }
  __syncthreads();

  // ---- pass C: wave-level candidate lists ----
  CandList<WL> list;
  list.init();
  float thr = -INFINITY;  // conservative bound on `total` below which nothing can enter

  for (int w = row_lo; w < row_hi; ++w) {
    const float lp = s_lp[w];
    const int len = s_len[w];
    const int flat0 = first ? 0 : w * V;
    if (s_fin[w]) {
      // Finished beam: masked log-prob is 0 at eos and f32.min elsewhere, whatever the
      // logits are.  All non-eos candidates tie, so only the first WL+1 columns and
      // the eos column can ever be selected.  Handled by wave 0 in scan order.
      if (wave == 0) {
        const float pen = penalty_of(p, len);
        const int ncol = min(V, WL + 1);
        for (int step = 0; step <= ncol; ++step) {
          int col;
          if (step < ncol) col = step;
          else if (p.eos >= ncol && p.eos < V) col = p.eos;
          else break;
          const float mp = (col == p.eos) ? 0.f : kF32Min;
          const float total = lp + mp;
          const float sc = total / pen;
          if (list.accepts(sc, flat0 + col)) list.insert(sc, total, flat0 + col);
        }
      }
      continue;
    }
    const float mx = s_max[w], lse = s_lse[w];
    const float pen_eos = penalty_of(p, len);
    const float pen_oth = penalty_of(p, len + 1);
    const float* row_g = lg + (size_t)w * V;
    const float* row_s = s_logits + (size_t)w * V;

    auto refresh_thr = [&]() {
      if (list.i[WL - 1] == EMPTY_IDX) { thr = -INFINITY; return; }
      const float tau = list.s[WL - 1];
      const float a = tau * pen_eos, c = tau * pen_oth;
      const float lo = fminf(a, c);
      thr = lo - fabsf(lo) * 4.8e-7f;  // a few ulps of slack; NaN/-inf fall through to exact
      if (!(thr == thr)) thr = -INFINITY;
    };
    refresh_thr();

    for (int c0 = wave * 256; c0 < V; c0 += NWV * 256) {
      const int c = c0 + lane * 4;
      float x[4];
      if (VEC) {
        if (c < V) {
          const float4 v4 = STAGE ? *reinterpret_cast<const float4*>(row_s + c)
                                  : *reinterpret_cast<const float4*>(row_g + c);
          x[0] = v4.x; x[1] = v4.y; x[2] = v4.z; x[3] = v4.w;
        } else {
          x[0] = x[1] = x[2] = x[3] = 0.f;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          x[e] = (c + e < V) ? (STAGE ? row_s[c + e] : row_g[c + e]) : 0.f;
      }
      float tot[4];
      bool pass = false;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float pr = p.input_is_log_prob ? x[e] : (x[e] - mx) - lse;
        tot[e] = lp + pr;
        pass = pass || ((c + e < V) && (tot[e] >= thr));
      }
      if (!__any(pass)) continue;
      // Exact scores.  Every element that would enter the list is a contender; each
      // round moves the wave's best contender into the list (argmax by shuffles), which
      // raises the bar for the rest, so a batch of 256 elements needs at most WL rounds.
      float sc[4];
      bool ok[4];
      bool any_ok = false;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float pen = (c + e == p.eos) ? pen_eos : pen_oth;
        sc[e] = tot[e] / pen;
        ok[e] = (c + e < V) && list.accepts(sc[e], flat0 + c + e);
        any_ok = any_ok || ok[e];
      }
      while (__any(any_ok)) {
        float bs = -INFINITY, bt = 0.f;
        int bi = EMPTY_IDX;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (ok[e] && cand_better(sc[e], flat0 + c + e, bs, bi)) {
            bs = sc[e];
            bt = tot[e];
            bi = flat0 + c + e;
          }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const float os = __shfl_xor(bs, o, 64);
          const float ot = __shfl_xor(bt, o, 64);
          const int oi = __shfl_xor(bi, o, 64);
          if (cand_better(os, oi, bs, bi)) { bs = os; bt = ot; bi = oi; }
        }
        list.insert(bs, bt, bi);
        any_ok = false;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ok[e] = ok[e] && (flat0 + c + e != bi) && list.accepts(sc[e], flat0 + c + e);
          any_ok = any_ok || ok[e];
        }
      }
      refresh_thr();
    }
  }

  // ---- merge the wave lists ----
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < WL; ++j) {
      s_cs[wave * WL + j] = list.s[j];
      s_ct[wave * WL + j] = list.t[j];
      s_ci[wave * WL + j] = list.i[j];
    }
  }
  __syncthreads();
  if (wave != 0) return;

  constexpr int NC = NWV * WL;
  constexpr int PER = (NC + 63) / 64;
  float cs[PER], ct[PER];
  int ci[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int e = lane + 64 * k;
    cs[k] = e < NC ? s_cs[e] : -INFINITY;
    ct[k] = e < NC ? s_ct[e] : 0.f;
    ci[k] = e < NC ? s_ci[e] : EMPTY_IDX;
  }
  for (int r = 0; r < W; ++r) {
    // lane-local best, then wave argmax
    float bs = cs[0], bt = ct[0];
    int bi = ci[0];
#pragma unroll
    for (int k = 1; k < PER; ++k)
      if (cand_better(cs[k], ci[k], bs, bi)) { bs = cs[k]; bt = ct[k]; bi = ci[k]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float os = __shfl_xor(bs, o, 64);
      const float ot = __shfl_xor(bt, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (cand_better(os, oi, bs, bi)) { bs = os; bt = ot; bi = oi; }
    }
    // remove the winner from its owner
#pragma unroll
    for (int k = 0; k < PER; ++k)
      if (ci[k] == bi && bi != EMPTY_IDX) { cs[k] = -INFINITY; ci[k] = EMPTY_IDX; }
    if (lane == 0) {
      const size_t o = (size_t)b * W + r;
      const int flat = (bi == EMPTY_IDX) ? 0 : bi;
      const int word = flat % V;
      const int parent = flat / V;
      float nlp = bt;
      if (first && p.start_offset != 0) {
        // reference quirk: the log-prob is gathered from beam 0 at the chosen word
        const float lp0 = s_lp[0];
        float pr;
        if (s_fin[0]) pr = (word == p.eos) ? 0.f : kF32Min;
        else {
          const float x = lg[word];
          pr = p.input_is_log_prob ? x : (x - s_max[0]) - s_lse[0];
        }
        nlp = lp0 + pr;
      }
      const int pfin = s_fin[parent];
      const int nfin = (pfin || word == p.eos) ? 1 : 0;
      const int nlen = s_len[parent] + ((word != p.eos && !nfin) ? 1 : 0);
      p.scores[o] = bs;
      p.ids[o] = word;
      p.parents[o] = parent;
      p.next_lp[o] = nlp;
      p.next_fin[o] = (uint8_t)nfin;
      p.next_len[o] = nlen;
    }
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
                                                        int B, int W) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
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

constexpr size_t beam_fixed_smem(int nwv, int wl) {
  auto pad4 = [](size_t n) { return (n + 3) / 4 * 4; };
  return sizeof(float) * (16 * 5 + pad4(2 * nwv) + 3 * pad4((size_t)nwv * wl));
}

template <int NT, int WL, bool VEC>
int launch_beam_v(const BeamArgs& a, size_t smem, hipStream_t s) {
  // more than 64 KiB of dynamic LDS must be opted into once per kernel
  static bool opted = false;
  if (smem > 64 * 1024 && !opted) {
    ICAP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(beam_step_kernel<NT, WL, VEC>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    opted = true;
  }
  hipLaunchKernelGGL((beam_step_kernel<NT, WL, VEC>), dim3(a.B), dim3(NT), smem, s, a);
  return ICAP_OK;
}

template <int NT, int WL>
int launch_beam(const BeamArgs& a, bool vec, size_t smem, hipStream_t s) {
  const int rc = vec ? launch_beam_v<NT, WL, true>(a, smem, s)
                     : launch_beam_v<NT, WL, false>(a, smem, s);
  if (rc != ICAP_OK) return rc;
  ICAP_LAUNCH_CHECK("beam_step_kernel");
  return ICAP_OK;
}

template <int WL>
int dispatch_beam(BeamArgs& a, bool vec, hipStream_t s) {
  // Stage the image's logits in LDS (one 1024-thread workgroup per CU) when they fit;
  // otherwise pass C re-reads them through L2 / Infinity Cache with 256-thread groups.
  const size_t need = (size_t)a.W * a.V * sizeof(float);
  const size_t fixed_big = beam_fixed_smem(16, WL);
  if (need + fixed_big <= 150 * 1024) {
    return launch_beam<1024, WL>(a, vec, fixed_big + need, s);
  }
  return launch_beam<256, WL>(a, vec, beam_fixed_smem(4, WL), s);
}

}  // namespace

int beam_step_f32(const float* logits, const float* log_probs, const uint8_t* finished,
                  const int32_t* lengths, int time, int start_offset, int eos_token,
                  double length_penalty_weight, int input_is_log_prob, float* scores,
                  int32_t* predicted_ids, int32_t* parent_ids, float* next_log_probs,
                  uint8_t* next_finished, int32_t* next_lengths, int B, int W, int V,
                  hipStream_t s) {
  ICAP_REQUIRE(logits && log_probs && finished && lengths, "beam_step_f32: null input");
  ICAP_REQUIRE(scores && predicted_ids && parent_ids && next_log_probs && next_finished &&
                   next_lengths,
               "beam_step_f32: null output");
  ICAP_REQUIRE(B > 0 && V > 0, "beam_step_f32: bad shape");
  ICAP_REQUIRE(W >= 1 && W <= ICAP_MAX_BEAM_WIDTH, "beam_step_f32: beam width %d outside [1,%d]",
               W, ICAP_MAX_BEAM_WIDTH);
  ICAP_REQUIRE(W <= V, "beam_step_f32: beam width %d exceeds vocab size %d", W, V);
  ICAP_REQUIRE(start_offset >= 0 && start_offset < W, "beam_step_f32: start_offset %d outside [0,%d)",
               start_offset, W);
  ICAP_REQUIRE((long long)W * V < (1LL << 31), "beam_step_f32: W*V overflows int32");
  BeamArgs a;
  a.logits = logits; a.log_probs = log_probs; a.finished = finished; a.lengths = lengths;
  a.time = time; a.start_offset = start_offset; a.eos = eos_token;
  a.input_is_log_prob = input_is_log_prob;
  a.scores = scores; a.ids = predicted_ids; a.parents = parent_ids;
  a.next_lp = next_log_probs; a.next_fin = next_finished; a.next_len = next_lengths;
  a.B = B; a.W = W; a.V = V;
  // (5 + len)^w in f32, divided by the double 6^w folded to f32: beam_search.py:127-128
  a.lpw = (float)length_penalty_weight;
  a.pen_den = (float)pow(6.0, length_penalty_weight);
  for (int l = 0; l < PEN_N; ++l) a.pen_table[l] = powf(5.0f + (float)l, a.lpw) / a.pen_den;
  const bool vec = (V % 4 == 0) && aligned16(logits);
  if (W <= 1) return dispatch_beam<1>(a, vec, s);
  if (W <= 2) return dispatch_beam<2>(a, vec, s);
  if (W <= 3) return dispatch_beam<3>(a, vec, s);
  if (W <= 4) return dispatch_beam<4>(a, vec, s);
  if (W <= 5) return dispatch_beam<5>(a, vec, s);
  if (W <= 8) return dispatch_beam<8>(a, vec, s);
  if (W <= 10) return dispatch_beam<10>(a, vec, s);
  return dispatch_beam<16>(a, vec, s);
}

int backtrack(const int32_t* ids, const int32_t* parents, const float* last_scores,
              const uint8_t* last_finished, int32_t* tokens, int32_t* best_beam,
              float* best_score, int T, int B, int W, hipStream_t s) {
  ICAP_REQUIRE(ids && parents && last_scores && last_finished && tokens, "backtrack: null pointer");
  ICAP_REQUIRE(T > 0 && B > 0 && W > 0, "backtrack: bad shape");
  hipLaunchKernelGGL(backtrack_kernel, dim3(ceil_div(B, 256)), dim3(256), 0, s, ids, parents,
                     last_scores, last_finished, tokens, best_beam, best_score, T, B, W);
  ICAP_LAUNCH_CHECK("backtrack_kernel");
  return ICAP_OK;
}

}  // namespace icap
