// Kernels of the transformer decoder path (reference decoder/transform_decoder.py,
// attention/attention.py:183-562, embedding/position_embedding.py) in incremental form: one
// launch set per new target position, with the self-attention keys / values of earlier
// positions kept per row in HBM.  The reference re-runs the whole prefix every step
// (transform_decoder.py:177); the per-position results are the same because every sub-layer
// is position-wise or causal and a row's prefix only ever grows (the decode loop does not
// reorder rows by parent beam).
//
//   embed_pos_f32            x[n] = embedding[ids[n]] + position_row         (`:206-210`)
//   residual_layernorm_f32   x += o (optional); y = LayerNorm(x) * gamma + beta (`:35,42`)
//   mha_self_step_f32        attention of the new position over cached positions 0..pos
//   mha_cross_step_f32       attention over the L encoder outputs of the row's image
//
// All four are bandwidth / latency bound; the dense projections around them run on the MFMA
// kernels of gemm_f32.hip.  Attention threads own CH consecutive floats of the heads*key_dim
// axis (a head spans key_dim/CH adjacent lanes, combined by lane shuffles), so consecutive
// lanes read consecutive addresses of every K / V row and the softmax state (running max,
// sum, weighted value) stays in registers.
#include <math.h>

#include "icap_common.h"

namespace icap {
namespace {

constexpr int NT = 256;

__global__ __launch_bounds__(NT) void embed_pos_kernel(const float* __restrict__ table,
                                                       const int32_t* __restrict__ ids,
                                                       const float* __restrict__ pos_row,
                                                       float* __restrict__ x, int N, int V, int D4) {
  const long long i = (long long)blockIdx.x * NT + threadIdx.x;
  if (i >= (long long)N * D4) return;
  const int n = (int)(i / D4), d = (int)(i - (long long)n * D4);
  const int id = ids ? ids[n] : 0;
  const bool ok = id >= 0 && id < V;  // (an id outside the table reads as a zero row)
  const float4 e = ok ? reinterpret_cast<const float4*>(table + (size_t)id * D4 * 4)[d]
                      : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 p = reinterpret_cast<const float4*>(pos_row)[d];
  reinterpret_cast<float4*>(x)[i] = make_float4(e.x + p.x, e.y + p.y, e.z + p.z, e.w + p.w);
}

// One wave per row; a lane keeps its MAXV float4 of the row in registers across the three
// passes (sum, squared deviations, output).  D % 4 == 0, D <= 64 * 4 * MAXV.
template <int MAXV>
__global__ __launch_bounds__(NT) void residual_layernorm_kernel(float* __restrict__ x,
                                                                const float* __restrict__ o,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta,
                                                                float* __restrict__ y, int N, int D,
                                                                float eps) {
  const int row = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= N) return;
  const int D4 = D >> 2;
  float4* xr = reinterpret_cast<float4*>(x + (size_t)row * D);
  const float4* orow = o ? reinterpret_cast<const float4*>(o + (size_t)row * D) : nullptr;
  float4 v[MAXV];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int d = lane + 64 * i;
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d < D4) {
      v[i] = xr[d];
      if (orow) {
        const float4 a = orow[d];
        v[i] = make_float4(v[i].x + a.x, v[i].y + a.y, v[i].z + a.z, v[i].w + a.w);
        xr[d] = v[i];
      }
      sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  const float mean = wave_sum(sum) / (float)D;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int d = lane + 64 * i;
    if (d < D4) {
      const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, e = v[i].w - mean;
      sq += (a * a + b * b) + (c * c + e * e);
    }
  }
  const float inv = 1.0f / sqrtf(wave_sum(sq) / (float)D + eps);
  float4* yr = reinterpret_cast<float4*>(y + (size_t)row * D);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int d = lane + 64 * i;
    if (d < D4) {
      const float4 g = reinterpret_cast<const float4*>(gamma)[d];
      const float4 b = reinterpret_cast<const float4*>(beta)[d];
      yr[d] = make_float4((v[i].x - mean) * inv * g.x + b.x, (v[i].y - mean) * inv * g.y + b.y,
                          (v[i].z - mean) * inv * g.z + b.z, (v[i].w - mean) * inv * g.w + b.w);
    }
  }
}

template <int CH>
struct Chunk {
  float e[CH];
};
template <int CH>
__device__ __forceinline__ Chunk<CH> load_chunk(const float* p) {
  Chunk<CH> c;
#pragma unroll
  for (int q = 0; q < CH / 4; ++q) {
    const float4 v = reinterpret_cast<const float4*>(p)[q];
    c.e[4 * q] = v.x; c.e[4 * q + 1] = v.y; c.e[4 * q + 2] = v.z; c.e[4 * q + 3] = v.w;
  }
  return c;
}
template <int CH>
__device__ __forceinline__ void store_chunk(float* p, const float (&e)[CH], float scale) {
#pragma unroll
  for (int q = 0; q < CH / 4; ++q)
    reinterpret_cast<float4*>(p)[q] = make_float4(e[4 * q] * scale, e[4 * q + 1] * scale,
                                                  e[4 * q + 2] * scale, e[4 * q + 3] * scale);
}
// Dot product of two chunks, summed over the lph adjacent lanes that share a head.
template <int CH>
__device__ __forceinline__ float head_dot(const float (&a)[CH], const Chunk<CH>& b, int lph) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < CH; ++i) s += a[i] * b.e[i];
  for (int m = 1; m < lph; m <<= 1) s += __shfl_xor(s, m, 64);
  return s;
}

// Softmax state of one (row, head chunk): running maximum, sum of exp and weighted value.
template <int CH>
struct Online {
  float m, l, acc[CH];
  __device__ __forceinline__ void init() {
    m = -INFINITY;
    l = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) acc[i] = 0.f;
  }
  __device__ __forceinline__ void add(float s, const Chunk<CH>& v) {
    const float mn = fmaxf(m, s);
    const float c = (m > -INFINITY) ? fast_exp(m - mn) : 0.f;
    const float w = fast_exp(s - mn);
    l = l * c + w;
#pragma unroll
    for (int i = 0; i < CH; ++i) acc[i] = acc[i] * c + w * v.e[i];
    m = mn;
  }
};

struct SelfArgs {
  const float* q;   // [N,HD]
  const float* kv;  // [Tcap][N][2*HD]: keys then values of every cached position
  float* out;       // [N,HD]
  int N, HD, pos, lph, mask_mode;
  float scale;
};

// Thread = (row, chunk).  mask_mode 0 reproduces the reference's use of the additive causal
// bias as a 0/1 mask (attention.py:471-473): every visible position gets -1e9 added in fp32
// (which absorbs ordinary scores: the weights become uniform), hidden ones about -1e18 and
// contribute exactly zero, so only positions 0..pos are visited either way.
template <int CH>
__global__ __launch_bounds__(NT) void mha_self_step_kernel(SelfArgs p) {
  const int chunks = p.HD / CH;
  const long long i = (long long)blockIdx.x * NT + threadIdx.x;
  // (whole heads never straddle the end of the grid: chunks * N is a multiple of lph)
  if (i >= (long long)p.N * chunks) return;
  const int n = (int)(i / chunks), c = (int)(i - (long long)n * chunks);
  const Chunk<CH> qc = load_chunk<CH>(p.q + (size_t)n * p.HD + c * CH);
  float q[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) q[e] = qc.e[e] * p.scale;   // query scaled first (`:501`)
  Online<CH> st;
  st.init();
  const size_t pos_stride = (size_t)p.N * 2 * p.HD;
  const float* base = p.kv + (size_t)n * 2 * p.HD + c * CH;
  for (int j = 0; j <= p.pos; ++j) {
    const Chunk<CH> k = load_chunk<CH>(base + j * pos_stride);
    const Chunk<CH> v = load_chunk<CH>(base + j * pos_stride + p.HD);
    float s = head_dot<CH>(q, k, p.lph);
    if (p.mask_mode == 0) s = s + (-1e9f);
    st.add(s, v);
  }
  store_chunk<CH>(p.out + (size_t)n * p.HD + c * CH, st.acc, 1.0f / st.l);
}

struct CrossArgs {
  const float* q;    // [N,HD]
  const float* kv;   // [G][L][2*HD]: keys then values of the encoder outputs of image g
  float* out;        // [N,HD]
  int G, R, stride_g, stride_k;  // row of (image g, beam k) = g*stride_g + k*stride_k
  int L, HD, lph;
  float scale;
  const float* mask;  // optional [G or 1][R][L] attention_mask, applied as (1 - mask) * -1e9
  int mask_stride_g;  // R*L, or 0 when one mask serves every group
};

// Workgroup = (image, 256 chunks): the K / V slices of the image are read once and applied to
// all R rows of the image, RB rows at a time in registers.
template <int CH, int RB>
__global__ __launch_bounds__(NT) void mha_cross_step_kernel(CrossArgs p) {
  const int chunks = p.HD / CH;
  const int cblocks = (chunks + NT - 1) / NT;
  const int g = blockIdx.x / cblocks;
  const int c = (blockIdx.x % cblocks) * NT + threadIdx.x;
  if (c >= chunks) return;   // (chunks is a multiple of lph, so heads stay whole)
  const float* kbase = p.kv + (size_t)g * p.L * 2 * p.HD + c * CH;
  for (int k0 = 0; k0 < p.R; k0 += RB) {
    float q[RB][CH];
    Online<CH> st[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int k = min(k0 + r, p.R - 1);   // surplus slots repeat the last row
      const int n = g * p.stride_g + k * p.stride_k;
      const Chunk<CH> qc = load_chunk<CH>(p.q + (size_t)n * p.HD + c * CH);
#pragma unroll
      for (int e = 0; e < CH; ++e) q[r][e] = qc.e[e] * p.scale;
      st[r].init();
    }
    const float* mrow = p.mask ? p.mask + (size_t)g * p.mask_stride_g : nullptr;
    for (int j = 0; j < p.L; ++j) {
      const Chunk<CH> kc = load_chunk<CH>(kbase + (size_t)j * 2 * p.HD);
      const Chunk<CH> vc = load_chunk<CH>(kbase + (size_t)j * 2 * p.HD + p.HD);
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        float sc = head_dot<CH>(q[r], kc, p.lph);
        // MultiHeadAttention._masked_softmax, attention.py:471-473, in fp32
        if (mrow) sc = sc + (1.0f - mrow[(size_t)min(k0 + r, p.R - 1) * p.L + j]) * -1e9f;
        st[r].add(sc, vc);
      }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int k = k0 + r;
      if (k < p.R) {
        const int n = g * p.stride_g + k * p.stride_k;
        store_chunk<CH>(p.out + (size_t)n * p.HD + c * CH, st[r].acc, 1.0f / st[r].l);
      }
    }
  }
}

// The same for few, wide heads (heads * key_dim / CH <= NT / 2 chunks, a multiple of 64, e.g.
// 8 heads of 64): one thread per chunk would leave most of the workgroup idle and one wave
// walking all L positions, so NT / chunks groups of waves take every (NT / chunks)-th position
// each and their soft-max states are joined through LDS (max of the maxima, sums and weighted
// values rescaled to it).
template <int CH, int RB>
__global__ __launch_bounds__(NT) void mha_cross_step_split_kernel(CrossArgs p) {
  __shared__ float s_m[NT * RB], s_l[NT * RB], s_acc[NT * RB * CH];
  const int chunks = p.HD / CH;
  const int nsplit = NT / chunks;
  const int g = blockIdx.x;
  const int c = threadIdx.x % chunks, split = threadIdx.x / chunks;
  const float* kbase = p.kv + (size_t)g * p.L * 2 * p.HD + c * CH;
  for (int k0 = 0; k0 < p.R; k0 += RB) {
    float q[RB][CH];
    Online<CH> st[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int k = min(k0 + r, p.R - 1);
      const int n = g * p.stride_g + k * p.stride_k;
      const Chunk<CH> qc = load_chunk<CH>(p.q + (size_t)n * p.HD + c * CH);
#pragma unroll
      for (int e = 0; e < CH; ++e) q[r][e] = qc.e[e] * p.scale;
      st[r].init();
    }
    const float* mrow = p.mask ? p.mask + (size_t)g * p.mask_stride_g : nullptr;
    for (int j = split; j < p.L; j += nsplit) {
      const Chunk<CH> kc = load_chunk<CH>(kbase + (size_t)j * 2 * p.HD);
      const Chunk<CH> vc = load_chunk<CH>(kbase + (size_t)j * 2 * p.HD + p.HD);
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        float sc = head_dot<CH>(q[r], kc, p.lph);
        if (mrow) sc = sc + (1.0f - mrow[(size_t)min(k0 + r, p.R - 1) * p.L + j]) * -1e9f;
        st[r].add(sc, vc);
      }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int o = r * NT + threadIdx.x;
      s_m[o] = st[r].m;
      s_l[o] = st[r].l;
#pragma unroll
      for (int e = 0; e < CH; ++e) s_acc[(size_t)e * NT * RB + o] = st[r].acc[e];
    }
    __syncthreads();
    if (split == 0) {
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const int k = k0 + r;
        if (k >= p.R) continue;
        float m = -INFINITY;
        for (int sp = 0; sp < nsplit; ++sp) m = fmaxf(m, s_m[r * NT + sp * chunks + c]);
        float l = 0.f, acc[CH];
#pragma unroll
        for (int e = 0; e < CH; ++e) acc[e] = 0.f;
        for (int sp = 0; sp < nsplit; ++sp) {
          const int o = r * NT + sp * chunks + c;
          const float w = (s_m[o] > -INFINITY) ? fast_exp(s_m[o] - m) : 0.f;  // (a group without positions)
          l += s_l[o] * w;
#pragma unroll
          for (int e = 0; e < CH; ++e) acc[e] += s_acc[(size_t)e * NT * RB + o] * w;
        }
        const int n = g * p.stride_g + k * p.stride_k;
        store_chunk<CH>(p.out + (size_t)n * p.HD + c * CH, acc, 1.0f / l);
      }
    }
    __syncthreads();
  }
}

// x[i] += o[i % period]: the residual add of the pre-norm wrapper (period = n) and the
// position-encoding add broadcast over the batch (period = length * hidden).
__global__ __launch_bounds__(NT) void add_kernel(float* __restrict__ x, const float* __restrict__ o,
                                                 long long n, long long period) {
  const long long i = (long long)blockIdx.x * NT + threadIdx.x;
  if (i < n) x[i] += o[i % period];
}

bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// ---- the general form of MultiHeadAttention._compute_attention (attention.py:476-524) ----
// for what the step kernels above do not serve: keys and values from different tensors, a
// value width of its own, several attention axes (flattened by the caller; the soft-max then
// runs over the last key axis only, as the reference's Softmax(axis=-1) does, `:457-458`),
// and the attention probabilities as an output.  Two plain kernels, one wave per output row.
//
// probs[b][n][t][:] = softmax per `seg` consecutive s of ((q[b][t][n][:] * scale) . k[b][s][n][:]
//                                                        + (1 - mask[b][n][t][s]) * -1e9)
constexpr int MHA_S_MAX = 4096;  // scores of one row staged in LDS (4 waves x 16 KB)

__global__ __launch_bounds__(NT) void mha_scores_kernel(const float* __restrict__ q,
                                                        const float* __restrict__ k,
                                                        const float* __restrict__ mask,
                                                        float* __restrict__ probs, int Bf, int T,
                                                        int S, int N, int Dk, int seg, float scale) {
  extern __shared__ float s_sc[];  // [NT / 64][S]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long row = (long long)blockIdx.x * (NT / 64) + wave;  // (b, n, t)
  if (row >= (long long)Bf * N * T) return;
  const int t = (int)(row % T);
  const int n = (int)((row / T) % N);
  const int b = (int)(row / ((long long)T * N));
  const float* qr = q + (((size_t)b * T + t) * N + n) * Dk;
  float* sc = s_sc + (size_t)wave * S;
  for (int s = lane; s < S; s += 64) {
    const float* kr = k + (((size_t)b * S + s) * N + n) * Dk;
    float a = 0.f;
    for (int d = 0; d < Dk; ++d) a += (qr[d] * scale) * kr[d];
    if (mask) a += (1.0f - mask[(size_t)row * S + s]) * -1e9f;
    sc[s] = a;
  }
  // (a wave reads back only what its own lanes wrote: no barrier, the LDS counter orders it)
  float* pr = probs + (size_t)row * S;
  for (int s0 = 0; s0 < S; s0 += seg) {
    float m = -INFINITY;
    for (int s = s0 + lane; s < s0 + seg; s += 64) m = fmaxf(m, sc[s]);
    m = wave_max(m);
    float sum = 0.f;
    for (int s = s0 + lane; s < s0 + seg; s += 64) {
      const float e = expf(sc[s] - m);
      sc[s] = e;
      sum += e;
    }
    sum = wave_sum(sum);
    for (int s = s0 + lane; s < s0 + seg; s += 64) pr[s] = sc[s] / sum;
  }
}

// out[b][t][n][:] = sum_s probs[b][n][t][s] * v[b][s][n][:]
__global__ __launch_bounds__(NT) void mha_combine_kernel(const float* __restrict__ probs,
                                                         const float* __restrict__ v,
                                                         float* __restrict__ out, int Bf, int T, int S,
                                                         int N, int Dv) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long row = (long long)blockIdx.x * (NT / 64) + wave;  // (b, t, n)
  if (row >= (long long)Bf * T * N) return;
  const int n = (int)(row % N);
  const int t = (int)((row / N) % T);
  const int b = (int)(row / ((long long)N * T));
  const float* pr = probs + (((size_t)b * N + n) * T + t) * S;
  for (int d = lane; d < Dv; d += 64) {
    float a = 0.f;
    for (int s = 0; s < S; ++s) a += pr[s] * v[(((size_t)b * S + s) * N + n) * Dv + d];
    out[(size_t)row * Dv + d] = a;
  }
}

}  // namespace

int mha_scores_f32(const float* q, const float* k, const float* mask, float* probs, int Bf, int T,
                   int S, int N, int Dk, int seg, hipStream_t s) {
  ICAP_REQUIRE(q && k && probs, "mha_scores_f32: null pointer");
  ICAP_REQUIRE(Bf > 0 && T > 0 && S > 0 && N > 0 && Dk > 0, "mha_scores_f32: bad shape");
  ICAP_REQUIRE(seg > 0 && S % seg == 0, "mha_scores_f32: the soft-max segment %d does not divide S = %d",
               seg, S);
  ICAP_REQUIRE(S <= MHA_S_MAX, "mha_scores_f32: S = %d exceeds %d key positions", S, MHA_S_MAX);
  const long long rows = (long long)Bf * N * T;
  ICAP_REQUIRE(rows / (NT / 64) < (1LL << 31), "mha_scores_f32: too many rows");
  const dim3 grid((unsigned)((rows + NT / 64 - 1) / (NT / 64))), block(NT);
  hipLaunchKernelGGL(mha_scores_kernel, grid, block, sizeof(float) * (NT / 64) * S, s, q, k, mask, probs,
                     Bf, T, S, N, Dk, seg, 1.0f / sqrtf((float)Dk));
  ICAP_LAUNCH_CHECK("mha_scores_kernel");
  return ICAP_OK;
}

int mha_combine_f32(const float* probs, const float* v, float* out, int Bf, int T, int S, int N, int Dv,
                    hipStream_t s) {
  ICAP_REQUIRE(probs && v && out, "mha_combine_f32: null pointer");
  ICAP_REQUIRE(Bf > 0 && T > 0 && S > 0 && N > 0 && Dv > 0, "mha_combine_f32: bad shape");
  const long long rows = (long long)Bf * T * N;
  ICAP_REQUIRE(rows / (NT / 64) < (1LL << 31), "mha_combine_f32: too many rows");
  const dim3 grid((unsigned)((rows + NT / 64 - 1) / (NT / 64))), block(NT);
  hipLaunchKernelGGL(mha_combine_kernel, grid, block, 0, s, probs, v, out, Bf, T, S, N, Dv);
  ICAP_LAUNCH_CHECK("mha_combine_kernel");
  return ICAP_OK;
}

int embed_pos_f32(const float* table, const int32_t* ids, const float* pos_row, float* x, int N,
                  int V, int D, hipStream_t s) {
  ICAP_REQUIRE(table && pos_row && x, "embed_pos_f32: null pointer");
  ICAP_REQUIRE(N > 0 && V > 0 && D > 0 && D % 4 == 0, "embed_pos_f32: bad shape (D %% 4 == 0)");
  ICAP_REQUIRE(aligned16(table) && aligned16(pos_row) && aligned16(x), "embed_pos_f32: unaligned buffer");
  const long long total = (long long)N * (D / 4);
  hipLaunchKernelGGL(embed_pos_kernel, dim3((unsigned)((total + NT - 1) / NT)), dim3(NT), 0, s, table,
                     ids, pos_row, x, N, V, D / 4);
  ICAP_LAUNCH_CHECK("embed_pos_kernel");
  return ICAP_OK;
}

int add_f32(float* x, const float* o, long long n, long long period, hipStream_t s) {
  ICAP_REQUIRE(x && o, "add_f32: null pointer");
  ICAP_REQUIRE(n > 0 && period > 0 && period <= n, "add_f32: bad sizes");
  ICAP_REQUIRE(n / NT < (1LL << 31), "add_f32: too many elements");
  hipLaunchKernelGGL(add_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, s, x, o, n, period);
  ICAP_LAUNCH_CHECK("add_kernel");
  return ICAP_OK;
}

int residual_layernorm_f32(float* x, const float* o, const float* gamma, const float* beta, float* y,
                           int N, int D, float eps, hipStream_t s) {
  ICAP_REQUIRE(x && gamma && beta && y, "residual_layernorm_f32: null pointer");
  ICAP_REQUIRE(N > 0 && D > 0 && D % 4 == 0 && D <= 4096,
               "residual_layernorm_f32: D must be a multiple of 4, at most 4096 (got %d)", D);
  ICAP_REQUIRE(aligned16(x) && (!o || aligned16(o)) && aligned16(gamma) && aligned16(beta) && aligned16(y),
               "residual_layernorm_f32: unaligned buffer");
  const dim3 grid(ceil_div(N, NT / 64)), block(NT);
  if (D <= 1024)
    hipLaunchKernelGGL(residual_layernorm_kernel<4>, grid, block, 0, s, x, o, gamma, beta, y, N, D, eps);
  else
    hipLaunchKernelGGL(residual_layernorm_kernel<16>, grid, block, 0, s, x, o, gamma, beta, y, N, D, eps);
  ICAP_LAUNCH_CHECK("residual_layernorm_kernel");
  return ICAP_OK;
}

int mha_self_step_f32(const float* q, const float* kv_cache, float* out, int N, int heads, int key_dim,
                      int pos, int mask_mode, hipStream_t s) {
  ICAP_REQUIRE(q && kv_cache && out, "mha_self_step_f32: null pointer");
  ICAP_REQUIRE(N > 0 && heads > 0 && key_dim > 0 && pos >= 0, "mha_self_step_f32: bad shape");
  ICAP_REQUIRE(key_dim % 4 == 0, "mha_self_step_f32: key_dim must be a multiple of 4 (got %d)", key_dim);
  ICAP_REQUIRE(mask_mode == 0 || mask_mode == 1, "mha_self_step_f32: unknown mask_mode %d", mask_mode);
  ICAP_REQUIRE(aligned16(q) && aligned16(kv_cache) && aligned16(out), "mha_self_step_f32: unaligned buffer");
  // (16-byte chunks with two lanes per head read at the same rate: 5.3 TB/s either way)
  const int ch = key_dim % 8 == 0 ? 8 : 4;
  const int lph = key_dim / ch;
  ICAP_REQUIRE(pow2(lph) && lph <= 64,
               "mha_self_step_f32: key_dim / %d must be a power of two up to 64 (key_dim %d)", ch, key_dim);
  SelfArgs a{q, kv_cache, out, N, heads * key_dim, pos, lph, mask_mode, 1.0f / sqrtf((float)key_dim)};
  const long long total = (long long)N * (a.HD / ch);
  ICAP_REQUIRE(total / NT < (1LL << 31), "mha_self_step_f32: too many elements");
  const dim3 grid((unsigned)((total + NT - 1) / NT)), block(NT);
  if (ch == 8) hipLaunchKernelGGL(mha_self_step_kernel<8>, grid, block, 0, s, a);
  else hipLaunchKernelGGL(mha_self_step_kernel<4>, grid, block, 0, s, a);
  ICAP_LAUNCH_CHECK("mha_self_step_kernel");
  return ICAP_OK;
}

int mha_cross_step_f32(const float* q, const float* kv, float* out, int G, int rows_per_group,
                       int row_stride_g, int row_stride_k, int L, int heads, int key_dim,
                       const float* mask, int mask_per_group, hipStream_t s) {
  ICAP_REQUIRE(q && kv && out, "mha_cross_step_f32: null pointer");
  ICAP_REQUIRE(G > 0 && rows_per_group > 0 && L > 0 && heads > 0 && key_dim > 0,
               "mha_cross_step_f32: bad shape");
  ICAP_REQUIRE(key_dim % 4 == 0, "mha_cross_step_f32: key_dim must be a multiple of 4 (got %d)", key_dim);
  ICAP_REQUIRE(aligned16(q) && aligned16(kv) && aligned16(out), "mha_cross_step_f32: unaligned buffer");
  // (16-byte chunks with two lanes per head read at the same rate: 5.3 TB/s either way)
  const int ch = key_dim % 8 == 0 ? 8 : 4;
  const int lph = key_dim / ch;
  ICAP_REQUIRE(pow2(lph) && lph <= 64,
               "mha_cross_step_f32: key_dim / %d must be a power of two up to 64 (key_dim %d)", ch, key_dim);
  CrossArgs a{q, kv, out, G, rows_per_group, row_stride_g, row_stride_k, L, heads * key_dim, lph,
              1.0f / sqrtf((float)key_dim), mask, mask_per_group ? rows_per_group * L : 0};
  const int cblocks = ceil_div(a.HD / ch, NT);
  ICAP_REQUIRE((long long)G * cblocks < (1LL << 31), "mha_cross_step_f32: too many workgroups");
  const dim3 grid((unsigned)(G * cblocks)), block(NT);
  const int rb = rows_per_group >= 3 ? 3 : rows_per_group;   // rows per register pass
  const int chunks = a.HD / ch;
  if (ch == 8 && chunks <= NT / 2 && chunks % 64 == 0 && NT % chunks == 0 && L >= 2 * (NT / chunks)) {
    // few wide heads: the positions are split over the workgroup's waves
    if (rb == 1) hipLaunchKernelGGL((mha_cross_step_split_kernel<8, 1>), dim3(G), block, 0, s, a);
    else if (rb == 2) hipLaunchKernelGGL((mha_cross_step_split_kernel<8, 2>), dim3(G), block, 0, s, a);
    else hipLaunchKernelGGL((mha_cross_step_split_kernel<8, 3>), dim3(G), block, 0, s, a);
    ICAP_LAUNCH_CHECK("mha_cross_step_split_kernel");
    return ICAP_OK;
  }
  if (ch == 8) {
    if (rb == 1) hipLaunchKernelGGL((mha_cross_step_kernel<8, 1>), grid, block, 0, s, a);
    else if (rb == 2) hipLaunchKernelGGL((mha_cross_step_kernel<8, 2>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((mha_cross_step_kernel<8, 3>), grid, block, 0, s, a);
  } else {
    if (rb == 1) hipLaunchKernelGGL((mha_cross_step_kernel<4, 1>), grid, block, 0, s, a);
    else if (rb == 2) hipLaunchKernelGGL((mha_cross_step_kernel<4, 2>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((mha_cross_step_kernel<4, 3>), grid, block, 0, s, a);
  }
  ICAP_LAUNCH_CHECK("mha_cross_step_kernel");
  return ICAP_OK;
}

}  // namespace icap
