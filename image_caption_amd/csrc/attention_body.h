// Device body of the additive (Bahdanau) attention step, shared by the stand-alone kernel
// (attention.hip) and by the vocab-projection launch that carries the next step's attention
// in its first workgroups (gemm_f32.hip).
//
// One work item = one feature block ("group": one image) and a block of RB of its rows (its
// beams), which share one pass over PF = W1(features)+b1 and over the features.
//   phase 1  score[k][l] = sum_u v[u]*tanh(PF[l][u] + q_k[u])  : each wave takes locations
//            l = wave, wave+WAVES, ...; a lane owns 4 consecutive u per 256-wide chunk, q_k
//            and v stay in registers; wavefront shuffles reduce over u, scores go to LDS.
//            tanh(a + b) = 1 - 2 / (1 + exp(2a) * exp(2b)): with exp(2 * PF) tabulated once per
//            decode (PF does not change over the steps) and exp(2 * q_k) formed once per work
//            item, an element costs one reciprocal instead of an exponential and a
//            reciprocal - the two quarter-rate instructions that bound this phase.  The
//            factored form is used while |PF| and |q| are below ATT_EXP_LIMIT (an accuracy
//            bound, see there); beyond it the work item takes the plain form on PF itself.
//   phase 2  softmax over l per row (one wave per row), weights kept in LDS.
//   phase 3  ctx[k][d] = sum_l w[k][l]*F[l][d] : waves split l, lanes own 4 d each; partial
//            sums meet in LDS; optional sigmoid gate (beta) applied on store.
#pragma once

#include "icap_common.h"

namespace icap {

constexpr int ATT_MAX_ROWS = ICAP_MAX_BEAM_WIDTH;

struct AttArgs {
  const float* pf;        // [G,L,U]
  const float* features;  // [G,L,D]
  const float* q;         // [N,U]  (unused by attention_fq_body)
  const float* v;         // [U]
  float bv;
  const float* score_bias;  // [N,L] or null
  const float* gate_state;  // [N,H] or null
  const float* gate_w;      // [H]
  float gate_b;
  float* ctx;      // [N,D]
  float* weights;  // [N,L]
  int G, R, stride_g, stride_k, L, U, D, H;
  int grid;  // workgroups launched (<= G)
  // attention_fq_body: q = q_state . q_w + q_b computed per pass, never stored
  const float* q_state;  // [N,H]
  const float* q_w;      // [H,U]
  const float* q_b;      // [U] or null
  int high_prio = 0;     // the kernel's waves raise their issue priority
  // Factored form of the tanh score (vector path): epf = exp(2 * PF) [G,L,U], built once per
  // decode, and *pf_extreme != 0 if any |PF| reached ATT_EXP_LIMIT (then the plain form runs).
  const float* epf = nullptr;
  const int* pf_extreme = nullptr;
  // The same table and the features as bf16 (reduced-precision mode, X16 bodies): half the bytes
  // of the two streams that make up the step's memory traffic.  The plain form (table flagged
  // or |q| beyond the limit) still reads pf in float32.
  const uint16_t* epf16 = nullptr;       // [G,L,U]
  const uint16_t* features16 = nullptr;  // [G,L,D]
};

// The factored form is used while |PF| and |q| stay below this.  exp(2x) is exp2(x * 2/ln 2);
// the rounding of that product alone gives the factor a relative error of about 1.2e-7 * |x|,
// and where PF and q cancel, tanh picks up half the relative error of the product of the two
// factors: at most about 1.2e-6 at the limit (1e-7 for tanh evaluated directly), and in
// proportion below it - a few 1e-7 at the pre-activations of a trained model.  (Overflow is
// far away: the factors stay finite and normal up to |x| = 40.)
constexpr float ATT_EXP_LIMIT = 10.f;

#if defined(__HIPCC__)
template <bool VEC>
__device__ __forceinline__ float4 att_ld4(const float* __restrict__ p, int i, int n) {
  if (VEC) {
    if (i + 3 < n) return *reinterpret_cast<const float4*>(p + i);
    return make_float4(0.f, 0.f, 0.f, 0.f);  // VEC implies n % 4 == 0 and i % 4 == 0
  }
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i + 0 < n) v.x = p[i + 0];
  if (i + 1 < n) v.y = p[i + 1];
  if (i + 2 < n) v.z = p[i + 2];
  if (i + 3 < n) v.w = p[i + 3];
  return v;
}
#endif

// LDS floats of one workgroup: score/weight [R][L] | beta [16] | partial ctx [WAVES][RB][Dpad].
inline size_t attention_lds_floats(int nt, int R, int rb, int L, int D) {
  const size_t sw = (size_t)((R * L + 3) / 4) * 4;
  const size_t dpad = (size_t)((D + 3) / 4) * 4;
  return sw + ATT_MAX_ROWS + (size_t)(nt / 64) * rb * dpad;
}

#if defined(__HIPCC__)
struct ATT_TRUE { static constexpr bool value = true; };
struct ATT_FALSE { static constexpr bool value = false; };
// The activation inside the score: tanh (the reference default, attention/attention.py:31)
// or one of the other named forms a caller may construct the layer with.
constexpr int ATT_ACT_TANH = 0, ATT_ACT_LINEAR = 1, ATT_ACT_RELU = 2, ATT_ACT_SIGMOID = 3;
template <int ACT>
__device__ __forceinline__ float att_act(float x) {
  if constexpr (ACT == ATT_ACT_LINEAR) return x;
  else if constexpr (ACT == ATT_ACT_RELU) return fmaxf(x, 0.f);
  else if constexpr (ACT == ATT_ACT_SIGMOID) return fast_sigmoid(x);
  else return fast_tanh(x);
}

// Batched, branch-free loads for the vector path: LB locations' rows in flight per wave.  Lanes
// whose columns lie beyond n read the row's first vector instead (finite values that only meet
// zero weights or never reach a store), so no load sits behind a branch and its wait.
template <int LB, int CH, int WAVES>
__device__ __forceinline__ void att_load_batch(float4 (&x)[LB][CH], const float* __restrict__ base,
                                               int row_floats, const int (&off)[CH], int l0, int L) {
#pragma unroll
  for (int b = 0; b < LB; ++b) {
    const int l = min(l0 + b * WAVES, L - 1);
#pragma unroll
    for (int ch = 0; ch < CH; ++ch)
      x[b][ch] = *reinterpret_cast<const float4*>(base + (size_t)l * row_floats + off[ch]);
  }
}

// The same batch from a bf16 array: four consecutive bf16 per lane and chunk (8 bytes), widened
// to float32 (a bf16 is the upper half of the float32 it rounds).
template <int LB, int CH, int WAVES>
__device__ __forceinline__ void att_load_batch16(float4 (&x)[LB][CH], const uint16_t* __restrict__ base,
                                                 int row_elems, const int (&off)[CH], int l0, int L) {
  uint2 raw[LB][CH];
#pragma unroll
  for (int b = 0; b < LB; ++b) {
    const int l = min(l0 + b * WAVES, L - 1);
#pragma unroll
    for (int ch = 0; ch < CH; ++ch)
      raw[b][ch] = *reinterpret_cast<const uint2*>(base + (size_t)l * row_elems + off[ch]);
  }
#pragma unroll
  for (int b = 0; b < LB; ++b)
#pragma unroll
    for (int ch = 0; ch < CH; ++ch)
      x[b][ch] = make_float4(__uint_as_float(raw[b][ch].x << 16), __uint_as_float(raw[b][ch].x & 0xffff0000u),
                             __uint_as_float(raw[b][ch].y << 16), __uint_as_float(raw[b][ch].y & 0xffff0000u));
}

// RB rows are held in registers at a time; UCH = number of 256-wide chunks of U.  The
// workgroup (NT threads) walks work items first, first + stride, ...
// QCOH: the queries were written by other workgroups of the same launch with device-scope
// stores and are read with device scope (VEC only).
// X16 (VEC, tanh): the factored scores read the bf16 table p.epf16 and the context the bf16
// features p.features16.
// SCOH: the gate state was written, and the context will be read, by other workgroups of the
// same launch: device-scope loads and stores for both.
template <int NT, int RB, int UCH, bool VEC, int ACT = ATT_ACT_TANH, bool QCOH = false, bool X16 = false,
          bool SCOH = false>
__device__ __forceinline__ void attention_body(const AttArgs& p, char* smem, int first, int stride) {
  static_assert(!X16 || (VEC && ACT == ATT_ACT_TANH), "bf16 operands serve the vector path with tanh");
  constexpr int WAVES = NT / 64;
  float* s_w = reinterpret_cast<float*>(smem);
  const int sw_elems = ((p.R * p.L + 3) / 4) * 4;
  float* s_beta = s_w + sw_elems;
  float* s_part = s_beta + ATT_MAX_ROWS;
  const int dpad = ((p.D + 3) / 4) * 4;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  float4 v4[UCH];
#pragma unroll
  for (int ch = 0; ch < UCH; ++ch) v4[ch] = att_ld4<VEC>(p.v, ch * 256 + lane * 4, p.U);

  // every block of RB rows of a group is its own work item: rows only share reads
  const int n_blk = (p.R + RB - 1) / RB;
  for (int wi = first; wi < p.G * n_blk; wi += stride) {
  const int g = wi / n_blk;
  const int kb = (wi - g * n_blk) * RB;             // first row of this work item
  const int ke = min(kb + RB, p.R);                 // one past its last row
  const float* pf_g = p.pf + (size_t)g * p.L * p.U;
  const float* f_g = p.features + (size_t)g * p.L * p.D;

  // beta gate per row (one wave per row)
  for (int k = kb + wave; k < ke; k += WAVES) {
    float beta = 1.f;
    if (p.gate_state) {
      const int n = g * p.stride_g + k * p.stride_k;
      const float* cs = p.gate_state + (size_t)n * p.H;
      float part = 0.f;
      for (int i = lane; i < p.H; i += 64) {
        float sv;
        if constexpr (SCOH) sv = __hip_atomic_load(cs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else sv = cs[i];
        part += sv * p.gate_w[i];
      }
      part = wave_sum(part);
      beta = fast_sigmoid(part + p.gate_b);
    }
    if (lane == 0) s_beta[k] = beta;
  }

  // ---- phase 1: scores ----
  {
    const int k0 = kb;
    float4 q4[RB][UCH];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int k = min(k0 + r, ke - 1);
      const int n = g * p.stride_g + k * p.stride_k;
      const float* qrow = p.q + (size_t)n * p.U;
#pragma unroll
      for (int ch = 0; ch < UCH; ++ch) {
        if constexpr (QCOH) {
          const int i = ch * 256 + lane * 4;
          const bool ok = i + 3 < p.U;
          const unsigned long long* s2 = reinterpret_cast<const unsigned long long*>(qrow + (ok ? i : 0));
          const unsigned long long lo = __hip_atomic_load(s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned long long hi = __hip_atomic_load(s2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          q4[r][ch] = ok ? make_float4(__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)),
                                       __uint_as_float((unsigned)hi), __uint_as_float((unsigned)(hi >> 32)))
                         : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
          q4[r][ch] = att_ld4<VEC>(qrow, ch * 256 + lane * 4, p.U);
        }
      }
    }
    // factored form: q4 becomes exp(2 q) when this work item may use the table (every wave
    // holds the same q4, so the decision is the same in all of them)
    bool factored = false;
    if constexpr (VEC && ACT == ATT_ACT_TANH) {
      if ((X16 ? p.epf16 != nullptr : p.epf != nullptr) && *p.pf_extreme == 0) {
        float mx = 0.f;
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
          for (int ch = 0; ch < UCH; ++ch)
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(q4[r][ch].x), fabsf(q4[r][ch].y))),
                       fmaxf(fabsf(q4[r][ch].z), fabsf(q4[r][ch].w)));
        factored = __builtin_amdgcn_ballot_w64(!(mx < ATT_EXP_LIMIT)) == 0ull;
        if (factored) {
          constexpr float C2 = 2.88539008177792681f;  // 2 / ln 2
#pragma unroll
          for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int ch = 0; ch < UCH; ++ch) {
              q4[r][ch].x = __builtin_amdgcn_exp2f(q4[r][ch].x * C2);
              q4[r][ch].y = __builtin_amdgcn_exp2f(q4[r][ch].y * C2);
              q4[r][ch].z = __builtin_amdgcn_exp2f(q4[r][ch].z * C2);
              q4[r][ch].w = __builtin_amdgcn_exp2f(q4[r][ch].w * C2);
            }
        }
      }
    }
    const float* x_g = (factored && !X16) ? p.epf + (size_t)g * p.L * p.U : pf_g;
    const uint16_t* x16_g = X16 ? p.epf16 + (size_t)g * p.L * p.U : nullptr;
    // FAST: x holds exp(2 PF) and q4 exp(2 q); tanh = 1 - 2 / (1 + product)
    // FAST, two elements per reciprocal (the reciprocal runs at a quarter of the vector rate
    // and is half of an element's cost): with d = 1 + exp(2 PF) exp(2 q),
    //   sum v (1 - 2 / d) = sum v + sum_pairs ((-2 v1) d2 + (-2 v2) d1) / (d1 d2)
    // d <= 1 + e^40 below the limit, so the product of two stays far inside the float range.
    float vsum = 0.f;
#pragma unroll
    for (int ch = 0; ch < UCH; ++ch) vsum += (v4[ch].x + v4[ch].y) + (v4[ch].z + v4[ch].w);
    auto score_location = [&](const float4 (&x)[UCH], int l, auto fast) {
      constexpr bool FAST = decltype(fast)::value;
      auto pair = [&](float x1, float q1, float v1, float x2, float q2, float v2, float acc) -> float {
        const float d1 = __builtin_fmaf(x1, q1, 1.0f), d2 = __builtin_fmaf(x2, q2, 1.0f);
        const float n = __builtin_fmaf(-2.0f * v2, d1, (-2.0f * v1) * d2);
        return __builtin_fmaf(n, __builtin_amdgcn_rcpf(d1 * d2), acc);
      };
      float a[RB];
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        if constexpr (FAST) {
          a[r] = vsum;
#pragma unroll
          for (int ch = 0; ch < UCH; ++ch) {
            a[r] = pair(x[ch].x, q4[r][ch].x, v4[ch].x, x[ch].y, q4[r][ch].y, v4[ch].y, a[r]);
            a[r] = pair(x[ch].z, q4[r][ch].z, v4[ch].z, x[ch].w, q4[r][ch].w, v4[ch].w, a[r]);
          }
        } else {
          a[r] = 0.f;
#pragma unroll
          for (int ch = 0; ch < UCH; ++ch) {
            a[r] += v4[ch].x * att_act<ACT>(x[ch].x + q4[r][ch].x);
            a[r] += v4[ch].y * att_act<ACT>(x[ch].y + q4[r][ch].y);
            a[r] += v4[ch].z * att_act<ACT>(x[ch].z + q4[r][ch].z);
            a[r] += v4[ch].w * att_act<ACT>(x[ch].w + q4[r][ch].w);
          }
        }
      }
      float part[RB];
#pragma unroll
      for (int r = 0; r < RB; ++r) part[r] = (k0 + r < ke) ? wave_sum(a[r]) : 0.f;  // (uniform)
      if (lane == 0) {
#pragma unroll
        for (int r = 0; r < RB; ++r) {
          const int k = k0 + r;
          if (k < ke) {
            float sc = part[r] + p.bv;
            if (p.score_bias) {
              const int n = g * p.stride_g + k * p.stride_k;
              sc += p.score_bias[(size_t)n * p.L + l];
            }
            s_w[k * p.L + l] = sc;
          }
        }
      }
    };
    if constexpr (VEC) {
      // two register sets of LB locations each: the next set's loads are in flight while the
      // current one is scored (same locations per wave and same order as the plain loop)
      // (More rows in flight do not pay beside the projection: at beam 5, 4 or 8 instead of 2
      // locations per set take the kernel from 161 to 182 / 256 registers, it no longer fits
      // beside the projection's waves and the decode drops from 190k to 156k / 150k captions/s.)
      constexpr int LB = (RB * UCH <= 6 && NT <= 256) ? 4 : 2;  // (512-thread form: inside a projection launch, 128 registers)
      constexpr int STEP = WAVES * LB;
      int off[UCH];
#pragma unroll
      for (int ch = 0; ch < UCH; ++ch) off[ch] = (ch * 256 + lane * 4 + 3 < p.U) ? ch * 256 + lane * 4 : 0;
      auto walk = [&](auto fast) {
        constexpr bool FAST16 = X16 && decltype(fast)::value;  // (the plain form reads pf itself)
        auto load = [&](float4 (&x)[LB][UCH], int l0) {
          if constexpr (FAST16) att_load_batch16<LB, UCH, WAVES>(x, x16_g, p.U, off, l0, p.L);
          else att_load_batch<LB, UCH, WAVES>(x, x_g, p.U, off, l0, p.L);
        };
        float4 xa[LB][UCH], xb[LB][UCH];
        load(xa, wave);
        for (int l0 = wave; l0 < p.L; l0 += 2 * STEP) {
          load(xb, l0 + STEP);
#pragma unroll
          for (int b = 0; b < LB; ++b)
            if (l0 + b * WAVES < p.L) score_location(xa[b], l0 + b * WAVES, fast);
          load(xa, l0 + 2 * STEP);
#pragma unroll
          for (int b = 0; b < LB; ++b)
            if (l0 + STEP + b * WAVES < p.L) score_location(xb[b], l0 + STEP + b * WAVES, fast);
        }
      };
      if constexpr (ACT == ATT_ACT_TANH) {
        if (factored) walk(ATT_TRUE{});
        else walk(ATT_FALSE{});
      } else {
        walk(ATT_FALSE{});
      }
    } else {
      for (int l = wave; l < p.L; l += WAVES) {
        float4 x[UCH];
#pragma unroll
        for (int ch = 0; ch < UCH; ++ch)
          x[ch] = att_ld4<VEC>(pf_g + (size_t)l * p.U, ch * 256 + lane * 4, p.U);
        score_location(x, l, ATT_FALSE{});
      }
    }
  }
  __syncthreads();

  // ---- phase 2: softmax over l, one wave per row ----
  for (int k = kb + wave; k < ke; k += WAVES) {
    float m = -INFINITY;
    for (int l = lane; l < p.L; l += 64) m = fmaxf(m, s_w[k * p.L + l]);
    m = wave_max(m);
    float sum = 0.f;
    for (int l = lane; l < p.L; l += 64) {
      const float e = fast_exp(s_w[k * p.L + l] - m);
      s_w[k * p.L + l] = e;
      sum += e;
    }
    sum = wave_sum(sum);
    const int n = g * p.stride_g + k * p.stride_k;
    for (int l = lane; l < p.L; l += 64) {
      const float w = s_w[k * p.L + l] / sum;
      s_w[k * p.L + l] = w;
      if (p.weights) p.weights[(size_t)n * p.L + l] = w;
    }
  }
  __syncthreads();

  // ---- phase 3: context ----
  {
    const int k0 = kb;
    for (int d0 = 0; d0 < p.D; d0 += 256) {
      const int d = d0 + lane * 4;
      float4 acc[RB];
#pragma unroll
      for (int r = 0; r < RB; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
      auto add_location = [&](const float4& f, int l) {
#pragma unroll
        for (int r = 0; r < RB; ++r) {
          const float w = s_w[min(k0 + r, ke - 1) * p.L + l];
          acc[r].x += w * f.x;
          acc[r].y += w * f.y;
          acc[r].z += w * f.z;
          acc[r].w += w * f.w;
        }
      };
      if constexpr (VEC) {
        constexpr int LB = 4, STEP = WAVES * LB;
        const int off[1] = {(d + 3 < p.D) ? d : 0};
        float4 fa[LB][1], fb[LB][1];
        const uint16_t* f16_g = X16 ? p.features16 + (size_t)g * p.L * p.D : nullptr;
        auto loadf = [&](float4 (&x)[LB][1], int l0) {
          if constexpr (X16) att_load_batch16<LB, 1, WAVES>(x, f16_g, p.D, off, l0, p.L);
          else att_load_batch<LB, 1, WAVES>(x, f_g, p.D, off, l0, p.L);
        };
        loadf(fa, wave);
        for (int l0 = wave; l0 < p.L; l0 += 2 * STEP) {
          loadf(fb, l0 + STEP);
#pragma unroll
          for (int b = 0; b < LB; ++b)
            if (l0 + b * WAVES < p.L) add_location(fa[b][0], l0 + b * WAVES);
          loadf(fa, l0 + 2 * STEP);
#pragma unroll
          for (int b = 0; b < LB; ++b)
            if (l0 + STEP + b * WAVES < p.L) add_location(fb[b][0], l0 + STEP + b * WAVES);
        }
      } else {
        for (int l = wave; l < p.L; l += WAVES) add_location(att_ld4<VEC>(f_g + (size_t)l * p.D, d, p.D), l);
      }
      if (d < dpad) {
#pragma unroll
        for (int r = 0; r < RB; ++r)
          *reinterpret_cast<float4*>(&s_part[(wave * RB + r) * dpad + d]) = acc[r];
      }
      __syncthreads();
      // reduce the wave partials; thread t owns (row r, element d0 + e)
      const int span = min(256, p.D - d0);
      for (int idx = tid; idx < RB * span; idx += NT) {
        const int r = idx / span;
        const int e = d0 + idx % span;
        const int k = k0 + r;
        if (k < ke) {
          float sum = 0.f;
#pragma unroll
          for (int w = 0; w < WAVES; ++w) sum += s_part[(w * RB + r) * dpad + e];
          const int n = g * p.stride_g + k * p.stride_k;
          if constexpr (SCOH)
            __hip_atomic_store(p.ctx + (size_t)n * p.D + e, sum * s_beta[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else
            p.ctx[(size_t)n * p.D + e] = sum * s_beta[k];
        }
      }
      __syncthreads();
    }
  }
  }  // work items
}
#endif

// ---------------------------------------------------------------------------------------
// The attention step with the query projection folded in, for a workgroup of NT threads that
// shares a compute unit with a matrix kernel: few resident waves, so every phase issues its
// loads in batches ahead of the arithmetic.
//   R = RB rows per group (one work item per group), U <= 512, U / 4 <= NT,
//   U % 4 == D % 4 == H % 4 == 0.
//   A pass takes IP groups: phase 0 computes q = state . W2 + b2 for their IP*RB rows in one
//   sweep over W2 (a thread owns four consecutive u and one of KG = NT/(U/4) slices of H;
//   slices meet in LDS), then phases 1-3 run per group as in attention_body.
// LDS floats: score/weight [R][L] | beta [16] | max(partial ctx [NT/64][RB][Dpad], query
// partials [KG][IP*RB][U] with the state rows [IP*RB][H] staged in their tail) | q [IP*RB][U].
// ---------------------------------------------------------------------------------------
inline size_t attention_fq_lds_floats(int nt, int R, int ip, int L, int U, int D, int H) {
  const size_t sw = (size_t)((R * L + 3) / 4) * 4;
  const size_t dpad = (size_t)((D + 3) / 4) * 4;
  const int quads = U / 4;
  const int kg = quads > 0 && quads <= nt ? nt / quads : 1;
  const size_t rows = (size_t)ip * R;
  size_t part = (size_t)(nt / 64) * R * dpad;
  const size_t qp = (size_t)kg * rows * U, st = rows * H;
  if (qp > part) part = qp;
  if (st > part) part = st;
  return sw + ATT_MAX_ROWS + part + rows * U;
}

#ifndef ICAP_ATT_STAMP
#define ICAP_ATT_STAMP(i)  // (experiment builds define it to record the device clock)
#endif
#ifndef ICAP_FQ_LB512
#define ICAP_FQ_LB512 4
#endif
#ifndef ICAP_FQ_QU
#define ICAP_FQ_QU 4
#endif
#ifndef ICAP_FQ_LB
#define ICAP_FQ_LB 2
#endif
#if defined(__HIPCC__)
template <int NT, int RB, int IP>
__device__ __forceinline__ void attention_fq_body(const AttArgs& p, char* smem, int first, int stride) {
  constexpr int WAVES = NT / 64, ROWS = RB * IP, UCH = 2, LB = NT >= 512 ? ICAP_FQ_LB512 : ICAP_FQ_LB;
  float* s_w = reinterpret_cast<float*>(smem);
  const int sw_elems = ((p.R * p.L + 3) / 4) * 4;
  float* s_beta = s_w + sw_elems;
  float* s_part = s_beta + ATT_MAX_ROWS;
  const int dpad = ((p.D + 3) / 4) * 4;
  const int quads = p.U >> 2;
  const int KG = NT / quads;
  const int part_elems = max(max(WAVES * RB * dpad, KG * ROWS * p.U), ROWS * p.H);
  float* s_q = s_part + part_elems;
  float* s_state = s_part + part_elems - ROWS * p.H;  // tail of the partials' region

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  float4 v4[UCH];
#pragma unroll
  for (int ch = 0; ch < UCH; ++ch) v4[ch] = att_ld4<true>(p.v, ch * 256 + lane * 4, p.U);

  for (int g0 = first; g0 < p.G; g0 += IP * stride) {
    // ---- phase 0: the queries of the pass's rows ----
    for (int idx = tid; idx < ROWS * (p.H >> 2); idx += NT) {
      const int rr = idx / (p.H >> 2), i4 = idx - rr * (p.H >> 2);
      const int g = min(g0 + (rr / RB) * stride, p.G - 1);  // (a missing group repeats the last one)
      const int n = g * p.stride_g + (rr % RB) * p.stride_k;
      reinterpret_cast<float4*>(s_state)[idx] =
          reinterpret_cast<const float4*>(p.q_state + (size_t)n * p.H)[i4];
    }
    __syncthreads();
    {
      const int kgi = tid / quads, quad = tid - kgi * quads;
      const bool active = kgi < KG;
      const int hk = (((p.H + KG - 1) / KG) + 7) & ~7;  // k values per slice
      const int k_lo = active ? min(kgi * hk, p.H) : 0, k_hi = active ? min(k_lo + hk, p.H) : 0;
      float4 acc[ROWS];
#pragma unroll
      for (int r = 0; r < ROWS; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
      const float* wcol = p.q_w + 4 * quad;
      constexpr int QU = ROWS <= 6 ? ICAP_FQ_QU : 4;  // weight vectors in flight per thread
      for (int k = k_lo; k < k_hi; k += QU) {  // (k_lo, k_hi, H are multiples of 4)
        float4 wv[QU];
#pragma unroll
        for (int j = 0; j < QU; ++j) {
          const bool ok = k + j < k_hi;
          const float4 t = *reinterpret_cast<const float4*>(wcol + (size_t)(ok ? k + j : k) * p.U);
          wv[j] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j4 = 0; j4 < QU; j4 += 4) {
          const int kk = min(k + j4, k_hi - 4);
#pragma unroll
          for (int r = 0; r < ROWS; ++r) {
            // (beyond k_hi the weight vectors are zero, so the repeated state values add nothing)
            const float4 sv = *reinterpret_cast<const float4*>(&s_state[r * p.H + kk]);
            const float se[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              acc[r].x += se[j] * wv[j4 + j].x;
              acc[r].y += se[j] * wv[j4 + j].y;
              acc[r].z += se[j] * wv[j4 + j].z;
              acc[r].w += se[j] * wv[j4 + j].w;
            }
            // (keeps the rows' LDS reads from being hoisted together, which would not fit the
            // register budget of two workgroups per compute unit)
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      __syncthreads();  // the state rows are dead: the partials may overwrite them
      if (active) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
          *reinterpret_cast<float4*>(&s_part[(kgi * ROWS + r) * p.U + 4 * quad]) = acc[r];
      }
      __syncthreads();
      for (int idx = tid; idx < ROWS * p.U; idx += NT) {
        const int r = idx / p.U, u = idx - r * p.U;
        float sum = p.q_b ? p.q_b[u] : 0.f;
        for (int w = 0; w < KG; ++w) sum += s_part[(w * ROWS + r) * p.U + u];
        s_q[idx] = sum;
      }
      __syncthreads();
    }

    ICAP_ATT_STAMP(2)
    for (int it = 0; it < IP; ++it) {
      ICAP_ATT_STAMP(3 + it)
      const int g = g0 + it * stride;
      if (g >= p.G) break;  // uniform
      const float* pf_g = p.pf + (size_t)g * p.L * p.U;
      const float* f_g = p.features + (size_t)g * p.L * p.D;

      // beta gate per row (one wave per row)
      for (int k = wave; k < RB; k += WAVES) {
        const int n = g * p.stride_g + k * p.stride_k;
        const float* cs = p.gate_state + (size_t)n * p.H;
        float part = 0.f;
        for (int i = lane; i < p.H; i += 64) part += cs[i] * p.gate_w[i];
        part = wave_sum(part);
        if (lane == 0) s_beta[k] = fast_sigmoid(part + p.gate_b);
      }

      // ---- phase 1: scores, LB locations' loads in flight per wave ----
      {
        float4 q4[RB][UCH];
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
          for (int ch = 0; ch < UCH; ++ch)
            q4[r][ch] = att_ld4<true>(s_q + (it * RB + r) * p.U, ch * 256 + lane * 4, p.U);
        for (int l0 = wave; l0 < p.L; l0 += WAVES * LB) {
          float4 x[LB][UCH];
#pragma unroll
          for (int b = 0; b < LB; ++b) {
            const int l = min(l0 + b * WAVES, p.L - 1);
#pragma unroll
            for (int ch = 0; ch < UCH; ++ch)
              x[b][ch] = att_ld4<true>(pf_g + (size_t)l * p.U, ch * 256 + lane * 4, p.U);
          }
#pragma unroll
          for (int b = 0; b < LB; ++b) {
            const int l = l0 + b * WAVES;
            float part[RB];
#pragma unroll
            for (int r = 0; r < RB; ++r) {
              float a = 0.f;
#pragma unroll
              for (int ch = 0; ch < UCH; ++ch) {
                a += v4[ch].x * fast_tanh(x[b][ch].x + q4[r][ch].x);
                a += v4[ch].y * fast_tanh(x[b][ch].y + q4[r][ch].y);
                a += v4[ch].z * fast_tanh(x[b][ch].z + q4[r][ch].z);
                a += v4[ch].w * fast_tanh(x[b][ch].w + q4[r][ch].w);
              }
              part[r] = wave_sum(a);
            }
            if (lane == 0 && l < p.L) {
#pragma unroll
              for (int r = 0; r < RB; ++r) s_w[r * p.L + l] = part[r] + p.bv;
            }
          }
        }
      }
      __syncthreads();
      if (it == 0) { ICAP_ATT_STAMP(5) }

      // ---- phase 2: softmax over l, one wave per row ----
      for (int k = wave; k < RB; k += WAVES) {
        float m = -INFINITY;
        for (int l = lane; l < p.L; l += 64) m = fmaxf(m, s_w[k * p.L + l]);
        m = wave_max(m);
        float sum = 0.f;
        for (int l = lane; l < p.L; l += 64) {
          const float e = fast_exp(s_w[k * p.L + l] - m);
          s_w[k * p.L + l] = e;
          sum += e;
        }
        sum = wave_sum(sum);
        for (int l = lane; l < p.L; l += 64) s_w[k * p.L + l] = s_w[k * p.L + l] / sum;
      }
      __syncthreads();

      // ---- phase 3: context, LB locations' loads in flight per wave ----
      for (int d0 = 0; d0 < p.D; d0 += 256) {
        const int d = d0 + lane * 4;
        float4 acc[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int l0 = wave; l0 < p.L; l0 += WAVES * LB) {
          float4 f[LB];
#pragma unroll
          for (int b = 0; b < LB; ++b)
            f[b] = att_ld4<true>(f_g + (size_t)min(l0 + b * WAVES, p.L - 1) * p.D, d, p.D);
#pragma unroll
          for (int b = 0; b < LB; ++b) {
            const int l = l0 + b * WAVES;
            if (l < p.L) {
#pragma unroll
              for (int r = 0; r < RB; ++r) {
                const float w = s_w[r * p.L + l];
                acc[r].x += w * f[b].x;
                acc[r].y += w * f[b].y;
                acc[r].z += w * f[b].z;
                acc[r].w += w * f[b].w;
              }
            }
          }
        }
        if (d < dpad) {
#pragma unroll
          for (int r = 0; r < RB; ++r)
            *reinterpret_cast<float4*>(&s_part[(wave * RB + r) * dpad + d]) = acc[r];
        }
        __syncthreads();
        const int span = min(256, p.D - d0);
        for (int idx = tid; idx < RB * span; idx += NT) {
          const int r = idx / span;
          const int e = d0 + idx % span;
          float sum = 0.f;
#pragma unroll
          for (int w = 0; w < WAVES; ++w) sum += s_part[(w * RB + r) * dpad + e];
          const int n = g * p.stride_g + r * p.stride_k;
          p.ctx[(size_t)n * p.D + e] = sum * s_beta[r];
        }
        __syncthreads();
      }
    }  // groups of the pass
  }    // passes
}
#endif

}  // namespace icap
