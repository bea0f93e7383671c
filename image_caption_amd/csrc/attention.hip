// Additive (Bahdanau) attention over L feature locations plus the small row-wise
// helpers of the decoder step (feature pooling, row expansion, embedding gather).
//
// attention_f32: one workgroup per feature block ("group": one image).  The rows of
// the group (its beams) are processed together so that PF = W1(features)+b1 and the
// features themselves are streamed once per group and step.
// The three phases (scores, softmax, context) are in attention_body.h.
#include <stdlib.h>

#include "attention_body.h"
#include "icap_common.h"

namespace icap {
namespace {

constexpr int ATT_THREADS = 256;

// A workgroup walks work items blockIdx.x, blockIdx.x + gridDim.x, ... (the grid may be
// capped, see attention_f32).
template <int RB, int UCH, bool VEC, int ACT = ATT_ACT_TANH, bool X16 = false>
__global__ __launch_bounds__(ATT_THREADS) void attention_f32_kernel(AttArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (p.high_prio) __builtin_amdgcn_s_setprio(3);
  attention_body<ATT_THREADS, RB, UCH, VEC, ACT, false, X16>(p, smem, blockIdx.x, gridDim.x);
}

// Query projection + attention in one launch (attention_fq_body), shaped to sit beside two
// workgroups of the exact-f32 projection on every compute unit: 4 waves, at most 96 VGPRs
// (five waves per SIMD; beam widths up to 3) and about 19 KB of LDS at the headline shape.
template <int RB>
__global__ __launch_bounds__(ATT_THREADS, RB <= 3 ? 5 : 4) void attention_fq_kernel(AttArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  attention_fq_body<ATT_THREADS, RB, 1>(p, smem, blockIdx.x, gridDim.x);
}

template <int RB, int UCH>
int launch_attention(const AttArgs& a, bool vec, size_t smem, hipStream_t s) {
  if (vec && a.epf16 != nullptr) {
    hipLaunchKernelGGL((attention_f32_kernel<RB, UCH, true, ATT_ACT_TANH, true>), dim3(a.grid),
                       dim3(ATT_THREADS), smem, s, a);
    ICAP_LAUNCH_CHECK("attention_f32_kernel<bf16 operands>");
    return ICAP_OK;
  }
  if (vec)
    hipLaunchKernelGGL((attention_f32_kernel<RB, UCH, true>), dim3(a.grid), dim3(ATT_THREADS), smem,
                       s, a);
  else
    hipLaunchKernelGGL((attention_f32_kernel<RB, UCH, false>), dim3(a.grid), dim3(ATT_THREADS),
                       smem, s, a);
  ICAP_LAUNCH_CHECK("attention_f32_kernel");
  return ICAP_OK;
}

// Other activations than tanh (module API only): one row per group.
template <int UCH, int ACT>
int launch_attention_act(const AttArgs& a, bool vec, size_t smem, hipStream_t s) {
  if (vec)
    hipLaunchKernelGGL((attention_f32_kernel<1, UCH, true, ACT>), dim3(a.grid), dim3(ATT_THREADS), smem,
                       s, a);
  else
    hipLaunchKernelGGL((attention_f32_kernel<1, UCH, false, ACT>), dim3(a.grid), dim3(ATT_THREADS),
                       smem, s, a);
  ICAP_LAUNCH_CHECK("attention_f32_kernel<act>");
  return ICAP_OK;
}

template <int ACT>
int dispatch_uch_act(const AttArgs& a, bool vec, size_t smem, hipStream_t s) {
  const int uch = ceil_div(a.U, 256);
  switch (uch) {
    case 1: return launch_attention_act<1, ACT>(a, vec, smem, s);
    case 2: return launch_attention_act<2, ACT>(a, vec, smem, s);
    case 3:
    case 4: return launch_attention_act<4, ACT>(a, vec, smem, s);
    default: return fail(ICAP_ERR_UNSUPPORTED, "attention_f32: U = %d exceeds 1024", a.U);
  }
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

// ---- epf = exp(2 * pf), the step-invariant factor of the tanh scores; *extreme is raised when
// an entry is too large for the factored form (or not a number) ----
__global__ __launch_bounds__(256) void att_exp_table_kernel(const float4* __restrict__ pf,
                                                            float4* __restrict__ epf, int* extreme,
                                                            size_t n4) {
  constexpr float C2 = 2.88539008177792681f;  // 2 / ln 2
  bool bad = false;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 x = pf[i];
    const float m = fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), fmaxf(fabsf(x.z), fabsf(x.w)));
    bad |= !(m < ATT_EXP_LIMIT);
    epf[i] = make_float4(__builtin_amdgcn_exp2f(x.x * C2), __builtin_amdgcn_exp2f(x.y * C2),
                         __builtin_amdgcn_exp2f(x.z * C2), __builtin_amdgcn_exp2f(x.w * C2));
  }
  if (__builtin_amdgcn_ballot_w64(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(extreme, 1);
}

// The same table rounded to bf16 (round to nearest even), and a plain float32 -> bf16 copy
// (the features): the operands of the X16 attention bodies.
__device__ __forceinline__ unsigned bf16_pair(float lo, float hi) {
  return (unsigned)__builtin_bit_cast(uint16_t, (__bf16)lo) |
         ((unsigned)__builtin_bit_cast(uint16_t, (__bf16)hi) << 16);
}
__global__ __launch_bounds__(256) void att_exp_table16_kernel(const float4* __restrict__ pf,
                                                              uint2* __restrict__ epf16, int* extreme,
                                                              size_t n4) {
  constexpr float C2 = 2.88539008177792681f;  // 2 / ln 2
  bool bad = false;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 x = pf[i];
    const float m = fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), fmaxf(fabsf(x.z), fabsf(x.w)));
    bad |= !(m < ATT_EXP_LIMIT);
    epf16[i] = make_uint2(bf16_pair(__builtin_amdgcn_exp2f(x.x * C2), __builtin_amdgcn_exp2f(x.y * C2)),
                          bf16_pair(__builtin_amdgcn_exp2f(x.z * C2), __builtin_amdgcn_exp2f(x.w * C2)));
  }
  if (__builtin_amdgcn_ballot_w64(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(extreme, 1);
}
__global__ __launch_bounds__(256) void to_bf16_kernel(const float4* __restrict__ x,
                                                      uint2* __restrict__ y, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = x[i];
    y[i] = make_uint2(bf16_pair(v.x, v.y), bf16_pair(v.z, v.w));
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
int attention_act_f32(const float* pf, const float* features, const float* q, const float* v,
                      float bv, const float* score_bias, const float* gate_state,
                      const float* gate_w, float gate_b, float* ctx, float* weights, int G, int R,
                      int stride_g, int stride_k, int L, int U, int D, int H, int max_workgroups,
                      int activation, hipStream_t s, const float* epf = nullptr,
                      const int* pf_extreme = nullptr, const void* epf16 = nullptr,
                      const void* features16 = nullptr);

// epf / pf_extreme: the table exp(2 * pf) and its flag from attention_exp_table_f32 (optional)
int attention_f32(const float* pf, const float* features, const float* q, const float* v,
                  float bv, const float* score_bias, const float* gate_state,
                  const float* gate_w, float gate_b, float* ctx, float* weights, int G, int R,
                  int stride_g, int stride_k, int L, int U, int D, int H, int max_workgroups,
                  hipStream_t s, const float* epf, const int* pf_extreme, const void* epf16,
                  const void* features16) {
  return attention_act_f32(pf, features, q, v, bv, score_bias, gate_state, gate_w, gate_b, ctx,
                           weights, G, R, stride_g, stride_k, L, U, D, H, max_workgroups,
                           ICAP_ATT_ACT_TANH, s, epf, pf_extreme, epf16, features16);
}

// activation: ICAP_ATT_ACT_*; anything but tanh serves one row per group (the layer called on
// its own, attention/attention.py:40-65 with a non-default activation, :31).
int attention_act_f32(const float* pf, const float* features, const float* q, const float* v,
                      float bv, const float* score_bias, const float* gate_state,
                      const float* gate_w, float gate_b, float* ctx, float* weights, int G, int R,
                      int stride_g, int stride_k, int L, int U, int D, int H, int max_workgroups,
                      int activation, hipStream_t s, const float* epf, const int* pf_extreme,
                      const void* epf16, const void* features16) {
  ICAP_REQUIRE(pf && features && q && v && ctx, "attention_f32: null pointer");
  ICAP_REQUIRE(((epf == nullptr && epf16 == nullptr) == (pf_extreme == nullptr)) && (!epf || aligned16(epf)),
               "attention_f32: the exp table comes with its flag, 16-byte aligned");
  ICAP_REQUIRE((epf16 == nullptr) == (features16 == nullptr) && (!epf16 || activation == ICAP_ATT_ACT_TANH) &&
                   (!epf16 || (aligned16(epf16) && aligned16(features16) && U % 4 == 0 && D % 4 == 0)),
               "attention_f32: the bf16 table and the bf16 features come together, aligned, with tanh");
  ICAP_REQUIRE(activation >= ICAP_ATT_ACT_TANH && activation <= ICAP_ATT_ACT_SIGMOID,
               "attention_f32: unknown activation %d", activation);
  ICAP_REQUIRE(activation == ICAP_ATT_ACT_TANH || R == 1,
               "attention_f32: activations other than tanh serve one row per group (got %d)", R);
  ICAP_REQUIRE(!gate_state || gate_w, "attention_f32: gate_state needs gate_w");
  ICAP_REQUIRE(G > 0 && L > 0 && U > 0 && D > 0, "attention_f32: bad shape");
  ICAP_REQUIRE(R >= 1 && R <= ATT_MAX_ROWS, "attention_f32: rows_per_group %d outside [1,%d]", R,
               ATT_MAX_ROWS);
  ICAP_REQUIRE(!gate_state || H > 0, "attention_f32: H must be positive with a gate");
  AttArgs a{pf, features, q, v, bv, score_bias, gate_state, gate_w, gate_b, ctx, weights,
            G, R, stride_g, stride_k, L, U, D, H,
            0, nullptr, nullptr, nullptr};
  a.epf = epf;
  a.pf_extreme = pf_extreme;
  a.epf16 = static_cast<const uint16_t*>(epf16);
  a.features16 = static_cast<const uint16_t*>(features16);
  {
    // The kernel's waves raise their instruction-issue priority: beside a matrix kernel on
    // another stream they get out of its way sooner (headline shape 99.2k against 98.7k
    // captions/s, B = 128: 54.9k against 52.9k, bf16 beam 3: 209k against 204k; raising the
    // matrix kernel's waves instead costs 6 %).  Tuning knob ICAP_ATT_PRIO=0|1.
    static const char* prio_env = getenv("ICAP_ATT_PRIO");
    a.high_prio = prio_env ? (prio_env[0] == '1') : 1;
  }
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
  const size_t smem = sizeof(float) * attention_lds_floats(ATT_THREADS, R, rb, L, D);
  ICAP_REQUIRE(smem <= 160 * 1024, "attention_f32: L*R or D too large for LDS (%zu bytes)", smem);
  if (activation == ICAP_ATT_ACT_LINEAR) return dispatch_uch_act<ATT_ACT_LINEAR>(a, vec, smem, s);
  if (activation == ICAP_ATT_ACT_RELU) return dispatch_uch_act<ATT_ACT_RELU>(a, vec, smem, s);
  if (activation == ICAP_ATT_ACT_SIGMOID) return dispatch_uch_act<ATT_ACT_SIGMOID>(a, vec, smem, s);
  switch (rb) {
    case 1: return dispatch_uch<1>(a, vec, smem, s);
    case 2: return dispatch_uch<2>(a, vec, smem, s);
    case 3: return dispatch_uch<3>(a, vec, smem, s);
    case 4: return dispatch_uch<4>(a, vec, smem, s);
    default: return dispatch_uch<5>(a, vec, smem, s);
  }
}

// ctx = attention(pf, features, q = state . q_w + q_b) with the beta gate on the same state:
// the query never reaches memory.  R <= 5 rows per group, U <= 512, U, D, H multiples of 4.
bool attention_fq_supported(int R, int L, int U, int D, int H) {
  return R >= 1 && R <= 5 && L >= 1 && U >= 4 && U <= 512 && U % 4 == 0 && D >= 4 && D % 4 == 0 &&
         H >= 4 && H % 4 == 0 &&
         sizeof(float) * attention_fq_lds_floats(ATT_THREADS, R, 1, L, U, D, H) <= 64 * 1024;
}

int attention_fq_f32(const float* pf, const float* features, const float* v, float bv,
                     const float* state, const float* gate_w, float gate_b, const float* q_w,
                     const float* q_b, float* ctx, int G, int R, int stride_g, int stride_k, int L,
                     int U, int D, int H, int max_workgroups, hipStream_t s) {
  ICAP_REQUIRE(pf && features && v && state && gate_w && q_w && ctx, "attention_fq_f32: null pointer");
  ICAP_REQUIRE(G > 0 && attention_fq_supported(R, L, U, D, H),
               "attention_fq_f32: shape not served (R=%d L=%d U=%d D=%d H=%d)", R, L, U, D, H);
  ICAP_REQUIRE(aligned16(pf) && aligned16(features) && aligned16(v) && aligned16(state) &&
                   aligned16(q_w) && aligned16(ctx),
               "attention_fq_f32: operands must be 16-byte aligned");
  AttArgs a{pf, features, nullptr, v, bv, nullptr, state, gate_w, gate_b, ctx, nullptr,
            G, R, stride_g, stride_k, L, U, D, H, G, state, q_w, q_b};
  if (max_workgroups > 0 && max_workgroups < a.grid) a.grid = max_workgroups;
  const size_t smem = sizeof(float) * attention_fq_lds_floats(ATT_THREADS, R, 1, L, U, D, H);
  const dim3 grid(a.grid), block(ATT_THREADS);
  switch (R) {
    case 1: hipLaunchKernelGGL(attention_fq_kernel<1>, grid, block, smem, s, a); break;
    case 2: hipLaunchKernelGGL(attention_fq_kernel<2>, grid, block, smem, s, a); break;
    case 3: hipLaunchKernelGGL(attention_fq_kernel<3>, grid, block, smem, s, a); break;
    case 4: hipLaunchKernelGGL(attention_fq_kernel<4>, grid, block, smem, s, a); break;
    default: hipLaunchKernelGGL(attention_fq_kernel<5>, grid, block, smem, s, a); break;
  }
  ICAP_LAUNCH_CHECK("attention_fq_kernel");
  return ICAP_OK;
}

// n = G*L*U floats (a multiple of 4), both buffers 16-byte aligned; *extreme must be 0 before.
int attention_exp_table_f32(const float* pf, float* epf, int* extreme, size_t n, hipStream_t s) {
  ICAP_REQUIRE(pf && epf && extreme, "attention_exp_table_f32: null pointer");
  ICAP_REQUIRE(n > 0 && n % 4 == 0 && aligned16(pf) && aligned16(epf),
               "attention_exp_table_f32: needs a multiple of 4 floats in 16-byte aligned buffers");
  const size_t n4 = n / 4;
  const unsigned grid = (unsigned)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  hipLaunchKernelGGL(att_exp_table_kernel, dim3(grid), dim3(256), 0, s,
                     reinterpret_cast<const float4*>(pf), reinterpret_cast<float4*>(epf), extreme, n4);
  ICAP_LAUNCH_CHECK("att_exp_table_kernel");
  return ICAP_OK;
}

int attention_exp_table_bf16(const float* pf, void* epf16, int* extreme, size_t n, hipStream_t s) {
  ICAP_REQUIRE(pf && epf16 && extreme, "attention_exp_table_bf16: null pointer");
  ICAP_REQUIRE(n > 0 && n % 4 == 0 && aligned16(pf) && aligned16(epf16),
               "attention_exp_table_bf16: needs a multiple of 4 elements in 16-byte aligned buffers");
  const size_t n4 = n / 4;
  const unsigned grid = (unsigned)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  hipLaunchKernelGGL(att_exp_table16_kernel, dim3(grid), dim3(256), 0, s,
                     reinterpret_cast<const float4*>(pf), reinterpret_cast<uint2*>(epf16), extreme, n4);
  ICAP_LAUNCH_CHECK("att_exp_table16_kernel");
  return ICAP_OK;
}

int to_bf16(const float* x, void* y16, size_t n, hipStream_t s) {
  ICAP_REQUIRE(x && y16 && n > 0 && n % 4 == 0 && aligned16(x) && aligned16(y16),
               "to_bf16: needs a multiple of 4 elements in 16-byte aligned buffers");
  const size_t n4 = n / 4;
  const unsigned grid = (unsigned)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  hipLaunchKernelGGL(to_bf16_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<const float4*>(x),
                     reinterpret_cast<uint2*>(y16), n4);
  ICAP_LAUNCH_CHECK("to_bf16_kernel");
  return ICAP_OK;
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
