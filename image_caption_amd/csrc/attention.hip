// Additive (Bahdanau) attention over L feature locations plus the small row-wise
// helpers of the decoder step (feature pooling, row expansion, embedding gather).
//
// attention_f32: one workgroup per feature block ("group": one image).  The rows of
// the group (its beams) are processed together so that PF = W1(features)+b1 and the
// features themselves are streamed once per group and step.
//   phase 1  score[k][l] = sum_u v[u]*tanh(PF[l][u] + q_k[u])  : each wave takes
//            locations l = wave, wave+4, ...; a lane owns 4 consecutive u per 256-wide
//            chunk, q_k and v stay in registers; wavefront shuffles reduce over u and
//            the scores are staged in LDS.
//   phase 2  softmax over l per row (one wave per row, shuffles), weights kept in LDS.
//   phase 3  ctx[k][d] = sum_l w[k][l]*F[l][d] : waves split l, lanes own 4 d each;
//            partial sums meet in LDS; optional sigmoid gate (beta) applied on store.
#include "icap_common.h"

namespace icap {
namespace {

constexpr int ATT_THREADS = 256;
constexpr int ATT_WAVES = ATT_THREADS / 64;
constexpr int ATT_MAX_ROWS = ICAP_MAX_BEAM_WIDTH;

template <bool VEC>
__device__ __forceinline__ float4 ld4(const float* __restrict__ p, int i, int n) {
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

struct AttArgs {
  const float* pf;        // [G,L,U]
  const float* features;  // [G,L,D]
  const float* q;         // [N,U]
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
};

// RB rows are held in registers at a time; UCH = number of 256-wide chunks of U.
template <int RB, int UCH, bool VEC>
__global__ __launch_bounds__(ATT_THREADS) void attention_f32_kernel(AttArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // carve: score/weight [R][L] | beta [R] (padded to 16) | partial ctx [WAVES][RB][Dpad]
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
  for (int ch = 0; ch < UCH; ++ch) v4[ch] = ld4<VEC>(p.v, ch * 256 + lane * 4, p.U);

  // a workgroup walks groups g, g + gridDim.x, ... (the grid may be capped, see attention_f32)
  // and every block of RB rows of a group is its own work item: rows only share reads
  const int n_blk = (p.R + RB - 1) / RB;
  for (int wi = blockIdx.x; wi < p.G * n_blk; wi += gridDim.x) {
  const int g = wi / n_blk;
  const int kb = (wi - g * n_blk) * RB;             // first row of this work item
  const int ke = min(kb + RB, p.R);                 // one past its last row
  const float* pf_g = p.pf + (size_t)g * p.L * p.U;
  const float* f_g = p.features + (size_t)g * p.L * p.D;

  // beta gate per row (one wave per row)
  for (int k = kb + wave; k < ke; k += ATT_WAVES) {
    float beta = 1.f;
    if (p.gate_state) {
      const int n = g * p.stride_g + k * p.stride_k;
      const float* cs = p.gate_state + (size_t)n * p.H;
      float part = 0.f;
      for (int i = lane; i < p.H; i += 64) part += cs[i] * p.gate_w[i];
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
#pragma unroll
      for (int ch = 0; ch < UCH; ++ch)
        q4[r][ch] = ld4<VEC>(p.q + (size_t)n * p.U, ch * 256 + lane * 4, p.U);
    }
    for (int l = wave; l < p.L; l += ATT_WAVES) {
      float4 x[UCH];
#pragma unroll
      for (int ch = 0; ch < UCH; ++ch)
        x[ch] = ld4<VEC>(pf_g + (size_t)l * p.U, ch * 256 + lane * 4, p.U);
      float part[RB];
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        part[r] = 0.f;
        if (k0 + r >= ke) continue;  // uniform: the last block may hold fewer rows
        float a = 0.f;
#pragma unroll
        for (int ch = 0; ch < UCH; ++ch) {
          a += v4[ch].x * fast_tanh(x[ch].x + q4[r][ch].x);
          a += v4[ch].y * fast_tanh(x[ch].y + q4[r][ch].y);
          a += v4[ch].z * fast_tanh(x[ch].z + q4[r][ch].z);
          a += v4[ch].w * fast_tanh(x[ch].w + q4[r][ch].w);
        }
        part[r] = wave_sum(a);
      }
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
    }
  }
  __syncthreads();

  // ---- phase 2: softmax over l, one wave per row ----
  for (int k = kb + wave; k < ke; k += ATT_WAVES) {
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
      for (int l = wave; l < p.L; l += ATT_WAVES) {
        const float4 f = ld4<VEC>(f_g + (size_t)l * p.D, d, p.D);
#pragma unroll
        for (int r = 0; r < RB; ++r) {
          const float w = s_w[min(k0 + r, ke - 1) * p.L + l];
          acc[r].x += w * f.x;
          acc[r].y += w * f.y;
          acc[r].z += w * f.z;
          acc[r].w += w * f.w;
        }
      }
      if (d < dpad) {
#pragma unroll
        for (int r = 0; r < RB; ++r)
          *reinterpret_cast<float4*>(&s_part[(wave * RB + r) * dpad + d]) = acc[r];
      }
      __syncthreads();
      // reduce the four wave partials; thread t owns (row r, element d0 + e)
      const int span = min(256, p.D - d0);
      for (int idx = tid; idx < RB * span; idx += ATT_THREADS) {
        const int r = idx / span;
        const int e = d0 + idx % span;
        const int k = k0 + r;
        if (k < ke) {
          float sum = 0.f;
#pragma unroll
          for (int w = 0; w < ATT_WAVES; ++w) sum += s_part[(w * RB + r) * dpad + e];
          const int n = g * p.stride_g + k * p.stride_k;
          p.ctx[(size_t)n * p.D + e] = sum * s_beta[k];
        }
      }
      __syncthreads();
    }
  }
  }  // groups
}

template <int RB, int UCH>
int launch_attention(const AttArgs& a, bool vec, size_t smem, hipStream_t s) {
  if (vec)
    hipLaunchKernelGGL((attention_f32_kernel<RB, UCH, true>), dim3(a.grid), dim3(ATT_THREADS), smem,
                       s, a);
  else
    hipLaunchKernelGGL((attention_f32_kernel<RB, UCH, false>), dim3(a.grid), dim3(ATT_THREADS),
                       smem, s, a);
  ICAP_LAUNCH_CHECK("attention_f32_kernel");
  return ICAP_OK;
}

template <int RB>
int dispatch_uch(const AttArgs& a, bool vec, size_t smem, hipStream_t s) {
  const int uch = ceil_div(a.U, 256);
  switch (uch) {
    case 1: return launch_attention<RB, 1>(a, vec, smem, s);
    case 2: return launch_attention<RB, 2>(a, vec, smem, s);
    case 3:
    case 4: return launch_attention<RB, 4>(a, vec, smem, s);
    default: return fail(ICAP_ERR_UNSUPPORTED, "attention_f32: U = %d exceeds 1024", a.U);
  }
}

// ---- pooled[G,D] = mean_L features (or masked sum) ----
__global__ __launch_bounds__(256) void pool_features_kernel(const float* __restrict__ features,
                                                            const float* __restrict__ mask,
                                                            float* __restrict__ pooled, int L,
                                                            int D) {
  const int g = blockIdx.x;
  const float* f = features + (size_t)g * L * D;
  float msum = 0.f;
  if (mask) {
    for (int l = 0; l < L; ++l) msum += mask[(size_t)g * L + l];
  }
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    float acc = 0.f;
    if (mask) {
      for (int l = 0; l < L; ++l) acc += f[(size_t)l * D + d] * mask[(size_t)g * L + l];
      pooled[(size_t)g * D + d] = (msum == 0.f) ? 0.f : acc;
    } else {
      for (int l = 0; l < L; ++l) acc += f[(size_t)l * D + d];
      pooled[(size_t)g * D + d] = acc / (float)L;
    }
  }
}

__global__ __launch_bounds__(256) void expand_rows_kernel(const float* __restrict__ src,
                                                          float* __restrict__ dst, int B, int W,
                                                          int H, float scale, int tile_mode) {
  const int r = blockIdx.x;
  const int b = (tile_mode == ICAP_TILE_REPEAT) ? r / W : r % B;
  for (int i = threadIdx.x; i < H; i += blockDim.x)
    dst[(size_t)r * H + i] = scale * src[(size_t)b * H + i];
}

__global__ __launch_bounds__(256) void embedding_kernel(const float* __restrict__ table,
                                                        const int32_t* __restrict__ ids,
                                                        float* __restrict__ out, int V, int E) {
  const int n = blockIdx.x;
  const int id = ids[n];
  const bool ok = id >= 0 && id < V;
  for (int i = threadIdx.x; i < E; i += blockDim.x)
    out[(size_t)n * E + i] = ok ? table[(size_t)id * E + i] : 0.f;
}

}  // namespace

// max_workgroups > 0 caps the grid: the workgroups then walk the groups in turn.  A caller
// that runs this kernel beside a matrix kernel on another stream caps it at one workgroup
// per compute unit, which leaves the matrix kernel its full occupancy.
int attention_f32(const float* pf, const float* features, const float* q, const float* v,
                  float bv, const float* score_bias, const float* gate_state,
                  const float* gate_w, float gate_b, float* ctx, float* weights, int G, int R,
                  int stride_g, int stride_k, int L, int U, int D, int H, int max_workgroups,
                  hipStream_t s) {
  ICAP_REQUIRE(pf && features && q && v && ctx, "attention_f32: null pointer");
  ICAP_REQUIRE(!gate_state || gate_w, "attention_f32: gate_state needs gate_w");
  ICAP_REQUIRE(G > 0 && L > 0 && U > 0 && D > 0, "attention_f32: bad shape");
  ICAP_REQUIRE(R >= 1 && R <= ATT_MAX_ROWS, "attention_f32: rows_per_group %d outside [1,%d]", R,
               ATT_MAX_ROWS);
  ICAP_REQUIRE(!gate_state || H > 0, "attention_f32: H must be positive with a gate");
  AttArgs a{pf, features, q, v, bv, score_bias, gate_state, gate_w, gate_b, ctx, weights,
            G, R, stride_g, stride_k, L, U, D, H,
            0};
  const bool vec = (U % 4 == 0) && (D % 4 == 0) && aligned16(pf) && aligned16(features) &&
                   aligned16(q) && aligned16(v);
  // rows held in registers per pass over PF: the fewest passes, then the fewest idle slots
  int rb = R;
  if (R > 5) {
    int best_passes = 1 << 30, best_waste = 1 << 30;
    for (int cand = 5; cand >= 3; --cand) {
      const int passes = ceil_div(R, cand), waste = passes * cand - R;
      if (passes < best_passes || (passes == best_passes && waste < best_waste)) {
        best_passes = passes; best_waste = waste; rb = cand;
      }
    }
  }
  a.grid = G * ceil_div(R, rb);  // one work item per (group, block of rb rows)
  if (max_workgroups > 0 && max_workgroups < a.grid) a.grid = max_workgroups;
  const int dpad = ((D + 3) / 4) * 4;
  const size_t smem =
      sizeof(float) * ((size_t)((R * L + 3) / 4) * 4 + ATT_MAX_ROWS + (size_t)ATT_WAVES * rb * dpad);
  ICAP_REQUIRE(smem <= 160 * 1024, "attention_f32: L*R or D too large for LDS (%zu bytes)", smem);
  switch (rb) {
    case 1: return dispatch_uch<1>(a, vec, smem, s);
    case 2: return dispatch_uch<2>(a, vec, smem, s);
    case 3: return dispatch_uch<3>(a, vec, smem, s);
    case 4: return dispatch_uch<4>(a, vec, smem, s);
    default: return dispatch_uch<5>(a, vec, smem, s);
  }
}

int pool_features_f32(const float* features, const float* mask, float* pooled, int G, int L,
                      int D, hipStream_t s) {
  ICAP_REQUIRE(features && pooled, "pool_features_f32: null pointer");
  ICAP_REQUIRE(G > 0 && L > 0 && D > 0, "pool_features_f32: bad shape");
  hipLaunchKernelGGL(pool_features_kernel, dim3(G), dim3(256), 0, s, features, mask, pooled, L, D);
  ICAP_LAUNCH_CHECK("pool_features_kernel");
  return ICAP_OK;
}

int expand_rows_f32(const float* src, float* dst, int B, int W, int H, float scale,
                    int tile_mode, hipStream_t s) {
  ICAP_REQUIRE(src && dst, "expand_rows_f32: null pointer");
  ICAP_REQUIRE(B > 0 && W > 0 && H > 0, "expand_rows_f32: bad shape");
  ICAP_REQUIRE(tile_mode == ICAP_TILE_REFERENCE || tile_mode == ICAP_TILE_REPEAT,
               "expand_rows_f32: unknown tile_mode %d", tile_mode);
  hipLaunchKernelGGL(expand_rows_kernel, dim3(B * W), dim3(256), 0, s, src, dst, B, W, H, scale,
                     tile_mode);
  ICAP_LAUNCH_CHECK("expand_rows_kernel");
  return ICAP_OK;
}

int embedding_f32(const float* table, const int32_t* ids, float* out, int N, int V, int E,
                  hipStream_t s) {
  ICAP_REQUIRE(table && ids && out, "embedding_f32: null pointer");
  ICAP_REQUIRE(N > 0 && V > 0 && E > 0, "embedding_f32: bad shape");
  hipLaunchKernelGGL(embedding_kernel, dim3(N), dim3(256), 0, s, table, ids, out, V, E);
  ICAP_LAUNCH_CHECK("embedding_kernel");
  return ICAP_OK;
}

}  // namespace icap
