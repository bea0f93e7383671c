// Exact-f32 MFMA GEMM kernels (v_mfma_f32_32x32x2_f32) for gfx950:
//   dense_f32 : y = act(x . w + bias)            (all Dense layers on the path)
//   lstm_f32  : Keras LSTMCell step with the [ctx | emb | h] concatenation, the
//               embedding gather and the i,f,g,o epilogue fused into the GEMM.
//
// Tiling.  A workgroup owns a BM x BN output tile and walks K in steps of BK = 32.
// Global -> registers -> LDS staging is double buffered (one barrier per K step);
// the loads of step t+1 are in flight while step t is multiplied.
//   A tile in LDS : float4 As[BK/4][BM+1]   As[kq][m] = A[m][4kq .. 4kq+3]
//   B tile in LDS : float  Bs[BK][BN]
// The MFMA takes one A and one B value per lane: lane l holds A[i = l&31][k = l>>5]
// and B[k = l>>5][j = l&31].  Inside a group of 8 k values, lane half h = l>>5 takes
// k = 4h + s at sub-step s, so one ds_read_b128 of As feeds four MFMAs.  The order in
// which k is summed inside a group is therefore permuted, which is irrelevant for
// the sum being complete.  Result layout of a 32x32 tile: column = l&31,
// row = (r&3) + 8*(r>>2) + 4*(l>>5) for accumulator register r.
#include <math.h>
#include <stdlib.h>

#ifdef ICAP_TIMELINE
#define ICAP_ATT_STAMP(i) if (threadIdx.x == 0 && blockIdx.x < 4096) icap_tl_f32[blockIdx.x][i] = wall_clock64();
extern __device__ long long icap_tl_f32[4096][8];
#endif
#include "attention_body.h"
#include "beam_records_body.h"
#include "icap_common.h"
#include "topk_epilogue.h"

namespace icap {
namespace {

constexpr int BK = 32;

struct DenseArgs {
  const float* A;
  const float* B;
  const float* bias;
  float* C;
  int M, N, K;
  int lda, ldb, ldc;
  int act;
  int tiles_m, tiles_n;
  int band_order;  // tile order: 0 = down a column of tiles, 1 = a band of panels per XCD,
                   // 2 = a band of tile rows per XCD (small kernels)
  // top-k epilogue (dense_topk_f32): per (row, 64-column half tile) records instead of C
  float* rec;    // [M][2*tiles_n][2 + 2*WL]: max, sum exp(x - max), WL best x, WL columns
  float* eos_x;  // [M]: the logit of column `eos`
  int eos;
  LogitMasks masks;  // additive masks applied before the records are built (default: none)
  // Device-side switch (engine only, null otherwise): the launch does nothing unless
  // *run_flag != 0.
  const int* run_flag = nullptr;
  int band_width = 0;  // band_order 1: panels per band (0: an eighth of the panels)
};

// Panels per band of the band tile order (tile_of_block_band).  An XCD sweeps a band with the
// panel index fastest, so the band's slice of the weights is what has to stay in its 4 MB L2
// while the activations stream through once per band.  An eighth of a 10 000-word vocabulary
// (10 panels of 256 KB at K = 512) nearly does; an eighth of 50 000 words (49 panels, 12.8 MB)
// does not, and every M tile then fetches the whole band again: 2.4 GB of L2 fills per launch at
// 2560 rows x 50 000 words against 0.90 GB with bands of 6 panels (8: 1.19 GB, 4: 0.99 GB,
// 3: 1.14 GB, 2: 1.41 GB; the launch is bound by the matrix pipe, so the rate moves by half a
// percent).  Hence: bands of at most 1.5 MB.  Tuning knob ICAP_BAND_WIDTH (0 = eighths).
inline int band_width_for(int K, int tiles_n) {
  static const char* bw_env = getenv("ICAP_BAND_WIDTH");
  if (bw_env) return atoi(bw_env);
  const size_t panel_bytes = (size_t)K * 128 * sizeof(float);
  const int eighth = (tiles_n + 7) / 8;
  const int fit = (int)(((size_t)3 << 19) / panel_bytes);
  return eighth <= fit ? 0 : (fit < 1 ? 1 : fit);
}

// Guarded 4-float load without control flow: the address is clamped to `safe` (any
// readable 16-byte aligned location) when the vector is out of range and the result
// is zeroed by a select, so the compiler keeps every load of a tile in flight at once
// (a branch around each load would serialise them behind s_waitcnt vmcnt(0)).
// VEC: col and ncols are multiples of 4, so a vector is entirely inside or outside.
template <bool VEC>
__device__ __forceinline__ float4 load4(const float* __restrict__ row_ptr,
                                        const float* __restrict__ safe, bool row_ok, int col,
                                        int ncols) {
  if (VEC) {
    const bool ok = row_ok && (col + 3 < ncols);
    const float4 v = *reinterpret_cast<const float4*>(ok ? row_ptr + col : safe);
    return ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float e[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bool ok = row_ok && (col + i < ncols);
    const float v = *(ok ? row_ptr + col + i : safe);
    e[i] = ok ? v : 0.f;
  }
  return make_float4(e[0], e[1], e[2], e[3]);
}

template <int BM, int BN, int WM, int WN, bool VEC>
__global__ __launch_bounds__(WM* WN * 64) void dense_f32_kernel(DenseArgs p) {
  constexpr int NT = WM * WN * 64;
  constexpr int TM = BM / WM / 32;
  constexpr int TN = BN / WN / 32;
  constexpr int A_LD = (BM * BK / 4) / NT;  // float4 loads per thread per K step
  constexpr int B_LD = (BK * BN / 4) / NT;
  static_assert(A_LD >= 1 && B_LD >= 1, "tile too small for the block");

  __shared__ float4 As[2][BK / 4][BM + 1];
  __shared__ float Bs[2][BK][BN];

  int tile_m, tile_n;
  tile_of_block(blockIdx.x, p.tiles_m, p.tiles_n, tile_m, tile_n);
  const int m0 = tile_m * BM;
  const int n0 = tile_n * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN;
  const int wn = wave % WN;
  const int l31 = lane & 31;
  const int half = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[A_LD], rb[B_LD];

  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      const int f = tid + i * NT;
      const int kq = f % (BK / 4);
      const int row = f / (BK / 4);
      const int gm = m0 + row;
      ra[i] = load4<VEC>(p.A + (size_t)gm * p.lda, p.A, gm < p.M, k0 + 4 * kq, p.K);
    }
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
      const int f = tid + i * NT;
      const int cg = f % (BN / 4);
      const int k = f / (BN / 4);
      const int gk = k0 + k;
      rb[i] = load4<VEC>(p.B + (size_t)gk * p.ldb, p.B, gk < p.K, n0 + 4 * cg, p.N);
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      const int f = tid + i * NT;
      As[buf][f % (BK / 4)][f / (BK / 4)] = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
      const int f = tid + i * NT;
      *reinterpret_cast<float4*>(&Bs[buf][f / (BN / 4)][4 * (f % (BN / 4))]) = rb[i];
    }
  };

  const int nk = (p.K + BK - 1) / BK;
  load_tiles(0);
  store_tiles(0);
  __syncthreads();

  for (int t = 0; t < nk; ++t) {
    const int buf = t & 1;
    if (t + 1 < nk) load_tiles((t + 1) * BK);
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      float4 a[TM];
      float b[TN][4];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[buf][2 * g + half][(wm * TM + i) * 32 + l31];
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int s = 0; s < 4; ++s)
          b[j][s] = Bs[buf][8 * g + 4 * half + s][(wn * TN + j) * 32 + l31];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j][2], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j][3], acc[i][j], 0, 0, 0);
        }
    }
    if (t + 1 < nk) store_tiles(buf ^ 1);
    __syncthreads();
  }

  // epilogue: bias, activation, store (two 128-byte row segments per instruction)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + (wn * TN + j) * 32 + l31;
    if (col >= p.N) continue;
    const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (row < p.M) {
          float v = acc[i][j][r] + bv;
          if (p.act == ICAP_ACT_RELU) v = fmaxf(v, 0.f);
          p.C[(size_t)row * p.ldc + col] = v;
        }
      }
    }
  }
}


// Reductions over each 32-lane half of a wavefront (the two halves hold different output
// rows of a 32x32 MFMA tile); every lane gets the result of its own half.
__device__ __forceinline__ float half_max(float v, int half) {
  const float ninf = -INFINITY;
  v = fmaxf(v, dpp_f<ICAP_DPP_ROW_SHR(1), 0xf>(ninf, v));
  v = fmaxf(v, dpp_f<ICAP_DPP_ROW_SHR(2), 0xf>(ninf, v));
  v = fmaxf(v, dpp_f<ICAP_DPP_ROW_SHR(4), 0xf>(ninf, v));
  v = fmaxf(v, dpp_f<ICAP_DPP_ROW_SHR(8), 0xf>(ninf, v));
  v = fmaxf(v, dpp_f<ICAP_DPP_BCAST15, 0xa>(ninf, v));
  const float lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
  const float hi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
  return half ? hi : lo;
}
__device__ __forceinline__ float half_sum(float v, int half) {
  v += dpp_f<ICAP_DPP_ROW_SHR(1), 0xf>(0.f, v);
  v += dpp_f<ICAP_DPP_ROW_SHR(2), 0xf>(0.f, v);
  v += dpp_f<ICAP_DPP_ROW_SHR(4), 0xf>(0.f, v);
  v += dpp_f<ICAP_DPP_ROW_SHR(8), 0xf>(0.f, v);
  v += dpp_f<ICAP_DPP_BCAST15, 0xa>(0.f, v);
  const float lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
  const float hi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
  return half ? hi : lo;
}

// Fast path of dense_f32: K % 32 == 0, 16-byte aligned operands, element counts below
// 2^31.  Out-of-range rows / columns of edge tiles are CLAMPED to valid ones instead of
// masked (their products land in accumulator rows / columns that are never stored), so
// the K loop has no masks and no per-iteration address arithmetic: every thread keeps
// 32-bit element offsets and the tile base pointers advance as scalars.
//
// TOPK > 0 replaces the store of C by the fused first half of the beam-search step: every
// wave reduces its 32 rows x 64 columns to one record per row - the maximum, the sum of
// exp(x - max) and the TOPK largest values with their columns (ties: lower column) - and
// the logit of column `eos` is kept aside.  A row's records over all half tiles hold what
// the log-softmax and the top-k over the whole row need, in 2 + 2*TOPK floats per 64
// logits, so the [M,N] logits are never written to memory.
// One LDS object per workgroup: the A/B staging tiles, aliased (TOPK) by the waves'
// transposition patches.
template <int BM, int BN, int WM, int WN, int TOPK>
struct FastLds {
  static constexpr int T_LD = 68;  // floats per patch row: 16-byte aligned, rows 4 banks apart
  struct Stage {
    float4 As[2][BK / 4][BM + 1];
    float Bs[2][BK][BN];
  };
  union U {
    Stage st;
    float patch[TOPK > 0 ? WM * WN * 32 * T_LD : 1];
  };
};

// The work of workgroup `block` of the grid (a kernel may give its first workgroups other
// work and pass the rest on with their index shifted down).
// Returns the tile row it worked on.  COH: the records are stored device-coherently.
// BMG / sub: the grid of tiles is BMG rows high and this workgroup takes rows
// [sub * BM, (sub + 1) * BM) of its tile (BMG == BM, sub == 0: the whole tile).
template <int BM, int BN, int WM, int WN, int TOPK, bool COH = false, int BMG = BM>
__device__ __forceinline__ int dense_fast_body(const DenseArgs& p, int block,
                                               typename FastLds<BM, BN, WM, WN, TOPK>::U& lds,
                                               int sub = 0) {
  constexpr int NT = WM * WN * 64;
  constexpr int TM = BM / WM / 32;
  constexpr int TN = BN / WN / 32;
  constexpr int A_F4 = BM * BK / 4, B_F4 = BK * BN / 4;  // float4 per staged tile
  // A tile does not have to divide evenly over the threads: the surplus threads of the last
  // pass repeat the tile's last vector (same address, same LDS slot, same value)
  constexpr int A_LD = (A_F4 + NT - 1) / NT;
  constexpr int B_LD = (B_F4 + NT - 1) / NT;
  constexpr int NG = BK / 8;
  constexpr int T_LD = FastLds<BM, BN, WM, WN, TOPK>::T_LD;
  auto& As = lds.st.As;
  auto& Bs = lds.st.Bs;

  int tile_m, tile_n;
  if (p.band_order) tile_of_block_band(block, p.tiles_m, p.tiles_n, tile_m, tile_n, p.band_width);
  else tile_of_block(block, p.tiles_m, p.tiles_n, tile_m, tile_n);
  const int m0 = tile_m * BMG + sub * BM;
  const int n0 = tile_n * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN;
  const int wn = wave % WN;
  const int l31 = lane & 31;
  const int half = lane >> 5;

  int off_a[A_LD], off_b[B_LD];
#pragma unroll
  for (int i = 0; i < A_LD; ++i) {
    const int f = min(tid + i * NT, A_F4 - 1);
    const int gm = min(m0 + f / (BK / 4), p.M - 1);
    off_a[i] = gm * p.lda + 4 * (f % (BK / 4));
  }
#pragma unroll
  for (int i = 0; i < B_LD; ++i) {
    const int f = min(tid + i * NT, B_F4 - 1);
    const int col = min(n0 + 4 * (f % (BN / 4)), p.N - 4);
    off_b[i] = (f / (BN / 4)) * p.ldb + col;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[A_LD], rb[B_LD];
#define ICAP_LOAD_TILES(a_tile_, b_tile_)                                             \
  _Pragma("unroll") for (int i = 0; i < A_LD; ++i) ra[i] =                            \
      *reinterpret_cast<const float4*>((a_tile_) + off_a[i]);                         \
  _Pragma("unroll") for (int i = 0; i < B_LD; ++i) rb[i] =                            \
      *reinterpret_cast<const float4*>((b_tile_) + off_b[i]);
#define ICAP_STORE_TILES(buf_)                                                        \
  _Pragma("unroll") for (int i = 0; i < A_LD; ++i) {                                  \
    const int f = min(tid + i * NT, A_F4 - 1);                                        \
    As[buf_][f % (BK / 4)][f / (BK / 4)] = ra[i];                                     \
  }                                                                                   \
  _Pragma("unroll") for (int i = 0; i < B_LD; ++i) {                                  \
    const int f = min(tid + i * NT, B_F4 - 1);                                        \
    *reinterpret_cast<float4*>(&Bs[buf_][f / (BN / 4)][4 * (f % (BN / 4))]) = rb[i];  \
  }
  auto read_frags = [&](int buf, int g, float4 (&a)[TM], float (&b)[TN][4]) {
#pragma unroll
    for (int i = 0; i < TM; ++i) a[i] = As[buf][2 * g + half][(wm * TM + i) * 32 + l31];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int s = 0; s < 4; ++s) b[j][s] = Bs[buf][8 * g + 4 * half + s][(wn * TN + j) * 32 + l31];
  };
  auto mfma_group = [&](const float4 (&a)[TM], const float (&b)[TN][4]) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j][1], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j][2], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j][3], acc[i][j], 0, 0, 0);
      }
  };

  // One barrier per K step, placed BELOW the next step's loads: memory operations are not
  // moved across a barrier, so the prefetch stays at the top of the step and is in flight
  // during the multiplies (issued below the barrier, the scheduler sinks the loads to just
  // above the LDS stores and every step waits out a memory latency).  Before the stores the
  // staging registers pass through an empty asm, which keeps their first use - and the wait
  // that goes with it - below the multiplies.
  const int nk = p.K / BK;
  const float* a_tile = p.A;
  const float* b_tile = p.B;
  ICAP_LOAD_TILES(a_tile, b_tile)
  ICAP_STORE_TILES(0)

  for (int t = 0; t < nk; ++t) {
    const int buf = t & 1;
    // No control flow around the prefetch: the last iteration simply re-loads its own
    // tile into the idle buffer, which keeps the staging registers out of scratch.
    const bool more = t + 1 < nk;
    a_tile += more ? BK : 0;
    b_tile += more ? (size_t)BK * p.ldb : 0;
    ICAP_LOAD_TILES(a_tile, b_tile)
    __syncthreads();  // step t is in LDS; nobody reads buffer buf^1 any more
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      float4 a[TM];
      float b[TN][4];
      read_frags(buf, g, a, b);
      mfma_group(a, b);
    }
#pragma unroll
    for (int i = 0; i < A_LD; ++i) asm volatile("" : "+v"(ra[i].x), "+v"(ra[i].y), "+v"(ra[i].z), "+v"(ra[i].w));
#pragma unroll
    for (int i = 0; i < B_LD; ++i) asm volatile("" : "+v"(rb[i].x), "+v"(rb[i].y), "+v"(rb[i].z), "+v"(rb[i].w));
    ICAP_STORE_TILES(buf ^ 1)
  }
  __syncthreads();  // (the top-k epilogue reuses the LDS)

  if constexpr (TOPK == 0) {
    // The bias values are fetched and added before any store, and interior tiles store
    // without per-row predicates: a load whose wait sits inside a predicated store block
    // makes every block wait for ALL outstanding memory operations, the earlier stores
    // included, which serialises the 32 stores of a wave.
    const bool relu = p.act == ICAP_ACT_RELU;
    const bool interior = (m0 + BM <= p.M) && (n0 + BN <= p.N);  // uniform
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + (wn * TN + j) * 32 + l31;
      const bool col_ok = col < p.N;
      float bv = (p.bias && col_ok) ? p.bias[col] : 0.f;
      asm volatile("" : "+v"(bv));  // the wait for the bias happens here, once
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[i][j][r] + bv;
          acc[i][j][r] = relu ? fmaxf(v, 0.f) : v;
        }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + (wn * TN + j) * 32 + l31;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row0 = m0 + (wm * TM + i) * 32 + 4 * half;
        float* c0 = p.C + (size_t)row0 * p.ldc + col;
        // (COH: written through to where every XCD sees it, for a reader inside the same launch)
        auto put = [](float* dst, float v) {
          if constexpr (COH) __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else *dst = v;
        };
        if (interior) {
#pragma unroll
          for (int r = 0; r < 16; ++r) put(c0 + (size_t)((r & 3) + 8 * (r >> 2)) * p.ldc, acc[i][j][r]);
        } else if (col < p.N) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            if (row0 + dr < p.M) put(c0 + (size_t)dr * p.ldc, acc[i][j][r]);
          }
        }
      }
    }
  } else {
    static_assert(TOPK == 0 || (TM == 1 && TN == 2), "the top-k epilogue assumes 32x64 wave tiles");
    constexpr int RL = 2 + 2 * TOPK;
    const int ht = 2 * tile_n + wn;       // half-tile index of this wave within the row
    const int n_ht = 2 * p.tiles_n;
    const int colbase = n0 + wn * 64;     // first column of the wave's patch
    // 1. the wave's 32 x 64 patch (+ bias, -inf beyond N) goes to LDS, row-major, so that
    //    the reductions over a row become lane-local (the staging tiles are idle: the K
    //    loop ended on a barrier)
    float* patch = lds.patch + wave * 32 * T_LD;
    {
      const int col0 = colbase + l31, col1 = col0 + 32;
      const bool ok0 = col0 < p.N, ok1 = col1 < p.N;
      const float bv0 = (p.bias && ok0) ? p.bias[col0] : 0.f;
      const float bv1 = (p.bias && ok1) ? p.bias[col1] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lr = (r & 3) + 8 * (r >> 2) + 4 * half;
        patch[lr * T_LD + l31] = ok0 ? acc[0][0][r] + bv0 : -INFINITY;
        patch[lr * T_LD + 32 + l31] = ok1 ? acc[0][1][r] + bv1 : -INFINITY;
      }
    }
    __syncthreads();
    // 2. lane (row = lane/2, part = lane%2) scans 32 consecutive columns of its row
    const int lr = lane >> 1, part = lane & 1;
    const int row = m0 + wm * 32 + lr;
    const int cbase = colbase + part * 32;
    float x[32];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(patch + lr * T_LD + part * 32 + 4 * q);
      x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
    }
    const bool masked = p.masks.any();
    if (masked) apply_logit_masks<32>(x, cbase, row, p.M, p.masks);
    TopkRow<TOPK> rec;
    if constexpr (TOPK > 1) rec = topk_scan_keyed<TOPK, 32>(x, patch + lr * T_LD + part * 32, cbase, p.eos, masked);
    else rec = topk_scan<TOPK, 32>(x, cbase, p.eos);
    if (rec.has_eos && row < p.M) {
      if constexpr (COH) __hip_atomic_store(p.eos_x + row, rec.eos_v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else p.eos_x[row] = rec.eos_v;
    }
    // 3. the two lanes of a row meet (neighbour swap inside each quad)
    rec = topk_merge_neighbour<TOPK>(rec);
    if (part == 0 && row < p.M) topk_store<TOPK, COH>(p.rec + ((size_t)row * n_ht + ht) * RL, rec);
  }
  return tile_m;
}

template <int BM, int BN, int WM, int WN, int MINW, int TOPK = 0>
__global__ __launch_bounds__(WM* WN * 64, MINW)
    __attribute__((amdgpu_waves_per_eu(WM * WN * MINW / 4, WM * WN * MINW / 4)))
    void dense_f32_fast_kernel(DenseArgs p) {
  __shared__ typename FastLds<BM, BN, WM, WN, TOPK>::U lds;
  if (p.run_flag != nullptr && *p.run_flag == 0) return;
  dense_fast_body<BM, BN, WM, WN, TOPK>(p, blockIdx.x, lds);
}

// The vocab projection with its top-k epilogue, carrying the NEXT decode step's attention
// (with the query projection folded in) in its first `att_blocks` workgroups: that attention
// depends only on the recurrent state the cell kernel has already written, not on this
// projection, and its loads and transcendentals run beside the matrix pipe of the tiles on
// the same compute units.  One launch on one stream replaces a second stream, its two
// event dependencies per step and the separate query kernel.  att_blocks is a multiple of 8
// so that the tile workgroups keep their XCD.
#ifdef ICAP_TIMELINE
// Experiment builds only: start and end of every workgroup of the merged launch on the
// 100 MHz clock, read back through icap_debug_timeline_f32.
#define ICAP_STAMP_F32(i) if (threadIdx.x == 0 && blockIdx.x < 4096) icap_tl_f32[blockIdx.x][i] = wall_clock64();
#else
#define ICAP_STAMP_F32(i)
#endif

template <int TOPK, int RB>
__global__ __launch_bounds__(512, 4)  // (min 4 waves per SIMD: two workgroups per compute unit)
    void dense_topk_attention_kernel(DenseArgs p, AttArgs a, int att_blocks) {
  __shared__ typename FastLds<128, 128, 4, 2, TOPK>::U lds;
  ICAP_STAMP_F32(0)
  if ((int)blockIdx.x < att_blocks) {
    __builtin_amdgcn_s_setprio(3);  // (get out of the tiles' way sooner)
    attention_fq_body<512, RB, (RB <= 3 ? 2 : 1)>(a, reinterpret_cast<char*>(&lds), blockIdx.x, att_blocks);
    ICAP_STAMP_F32(1)
    return;
  }
  dense_fast_body<128, 128, 4, 2, TOPK>(p, blockIdx.x - att_blocks, lds);
  ICAP_STAMP_F32(1)
}

// The same launch with the query projection as matrix tiles of its own instead of a product
// inside every attention workgroup (where each of them streams the whole [H,U] kernel for its
// six rows: 65 of an attention workgroup's 122 us at the headline shape).  Grid: nq tiles of
// 64 x 128 of q = state . W2 + b2 first, then the attention workgroups, then the vocab tiles.
// The query tiles store q with device scope and count themselves on *q_done; an attention
// workgroup waits (one lane polling, with a sleep) until the count reaches q_target and
// reads q with device scope.  Workgroups start in grid order, so the query tiles never wait
// for a slot held by a waiting attention workgroup; should a wait run out all the same
// (2^16 polls, a fraction of a second) the workgroup raises *err and leaves its rows
// untouched - the caller turns that into an invalid result rather than a wrong one.
//
// The tail: the launch ends when its last workgroups do, and a 128 x 128 tile that starts late
// holds its compute unit for 40 to 55 us.  The vocab tiles from number split_from on (the last
// ones in grid order, a multiple of 8 of them) are therefore cut into two workgroups of
// 64 x 128 each (four of the eight waves work, the others leave at once; the two halves sit 8
// apart in the grid, on the same XCD, where they share the tile's weight panel in L2), which
// halves the granularity of what is still running at the end.
template <int TOPK, int RB>
__global__ __launch_bounds__(512, 4)
    void dense_topk_attq_kernel(DenseArgs p, DenseArgs pq, AttArgs a, int nq, int att_blocks,
                                int tiles_before, int split_from, unsigned* q_done, unsigned q_target,
                                int* err) {
  __shared__ typename FastLds<128, 128, 4, 2, TOPK>::U lds;
  static_assert(sizeof(typename FastLds<64, 128, 2, 2, TOPK>::U) <= sizeof(lds), "the half tile's LDS lives in the tile's");
  static_assert(sizeof(typename FastLds<64, 128, 2, 4, 0>::U) <= sizeof(lds), "the query tile's LDS lives in the vocab tile's");
  // (both leading sections are padded to multiples of 8 workgroups so that the vocab tiles
  // keep their XCD; the padding workgroups return at once)
  const int nq_pad = (nq + 7) & ~7, att_pad = (att_blocks + 7) & ~7;
  // grid order: query tiles, the first tiles_before vocab tiles (a multiple of 8), the
  // attention workgroups, the other vocab tiles
  int b = blockIdx.x;
  int tile_id = -1;
  if (b >= nq_pad) {
    if (b < nq_pad + tiles_before) { tile_id = b - nq_pad; }
    else if (b >= nq_pad + tiles_before + att_pad) { tile_id = b - nq_pad - att_pad; }
    else { b -= tiles_before; }
  }
  ICAP_STAMP_F32(0)
  if (tile_id >= 0) {
    if (tile_id < split_from) {
      dense_fast_body<128, 128, 4, 2, TOPK>(p, tile_id, lds);
    } else {
      if (threadIdx.x >= 256) return;  // (a half tile is the work of four waves)
      const int idx = tile_id - split_from;
      dense_fast_body<64, 128, 2, 2, TOPK, false, 128>(
          p, split_from + ((idx >> 4) << 3) + (idx & 7),
          *reinterpret_cast<typename FastLds<64, 128, 2, 2, TOPK>::U*>(&lds), (idx >> 3) & 1);
    }
    ICAP_STAMP_F32(1)
    return;
  }
  if (b < nq_pad) {
    if (b >= nq) return;
    dense_fast_body<64, 128, 2, 4, 0, true>(pq, b, *reinterpret_cast<typename FastLds<64, 128, 2, 4, 0>::U*>(&lds));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every store of the tile acknowledged
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(q_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ICAP_STAMP_F32(1)
    return;
  }
  if (b < nq_pad + att_pad) {
    if (b - nq_pad >= att_blocks) return;
    int* s_ok = reinterpret_cast<int*>(&lds);
    if (threadIdx.x == 0) {
      unsigned polls = 0;
      bool ok = true;
      while (__hip_atomic_load(q_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < q_target) {
        __builtin_amdgcn_s_sleep(32);
        if (++polls > (1u << 16)) { ok = false; break; }
      }
      if (!ok) *err = 1;
      *s_ok = ok ? 1 : 0;
    }
    __syncthreads();
    const bool ok = *s_ok != 0;
    __syncthreads();  // (the attention body reuses the word's LDS)
    ICAP_STAMP_F32(2)
    if (!ok) return;
    __builtin_amdgcn_s_setprio(3);
    attention_body<512, RB, 2, true, ATT_ACT_TANH, true>(a, reinterpret_cast<char*>(&lds), b - nq_pad, att_blocks);
    ICAP_STAMP_F32(1)
    return;
  }
}

// The vocab projection with its top-k epilogue, carrying the beam-search step that consumes
// its records in the same launch: besides the tiles the grid holds one workgroup per image,
// which waits until the row tiles holding the image's records are complete (a counter per
// tile row, raised once by every tile after its records are written, with device-scope
// stores that the step reads back with device-scope loads) and then runs the step of
// beam_records_kernel.  What it saves is the launch of that
// kernel behind the projection's last tile, and the images whose rows finish early are
// stepped in the compute-unit slots that the last, partly filled round of tiles leaves idle.
// Measured at the headline shape it loses all the same (92k against 100k captions/s): the
// tiles of the last round end over a window of 45 us, the image workgroups that become
// resident beside them slow them by about 12 us, and the step behind the last tile takes as
// long as the separate kernel - so the engine uses it only on request.
//
// Grid layout: workgroup pid lives on XCD pid % 8 at position pid / 8.  An XCD's positions
// are its tiles in the usual order with `chunks` runs of cb image workgroups put in front of
// its ins[c]-th tile (image = 8 * slot + xcd); the host places a run behind the tiles it waits
// for by a whole round of resident workgroups, so that in the normal course it never waits,
// and the runs of the last round behind every tile.  Workgroups start in grid order on each
// XCD, so everything a waiting workgroup needs has been started before it; should a wait
// run out all the same (2^18 polls, about a second), the workgroup raises *err and leaves its image
// untouched - the caller turns that into an invalid result rather than a wrong one.
struct BeamMergeArgs {
  unsigned* done;   // [tiles_m], counted up over the launches of a decode
  unsigned target;  // done[tm] of a complete tile row in this launch
  int* err;
  int q, r;         // tiles per XCD: q, one more on the first r
  int chunks, cb;
  int ins[8];
};

template <int TOPK, int ITEMS>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_waves_per_eu(4, 4)))
    void dense_topk_beam_kernel(DenseArgs p, RecArgs r, BeamMergeArgs m) {
  __shared__ typename FastLds<128, 128, 4, 2, TOPK>::U lds;
  __shared__ int s_ok;
  static_assert(sizeof(RecLds<TOPK, 512>) <= sizeof(lds), "the step's LDS lives in the tile buffers");
  const int xcd = blockIdx.x & 7;
  int pos = blockIdx.x >> 3;
  int slot = -1;  // image slot, or -1: the XCD's pos-th tile
  {
    int c = 0;
    for (; c < m.chunks; ++c) {
      const int start = m.ins[c] + c * m.cb;
      if (pos < start) break;
      if (pos < start + m.cb) { slot = c * m.cb + (pos - start); break; }
    }
    if (slot < 0) pos -= c * m.cb;
  }
  if (slot < 0) {
    if (pos >= m.q + (xcd < m.r ? 1 : 0)) return;
    const int tile_m = dense_fast_body<128, 128, 4, 2, TOPK, true>(p, pos * 8 + xcd, lds);
    // The records were written through with device scope; once every wave has seen its
    // stores acknowledged the count may go up.  No cache-wide write-back or invalidate is
    // involved on either side (the tiles next door live on what those would throw out).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
      __hip_atomic_fetch_add(m.done + tile_m, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  const int b = slot * 8 + xcd;
  if (b >= r.B) return;
  if (threadIdx.x == 0) {
    const int row_lo = b * r.rpi;
    const int row_hi = row_lo + (r.time <= 0 ? 0 : r.W - 1);
    bool ok = true;
    for (int tm = row_lo / 128; tm <= row_hi / 128 && ok; ++tm) {
      unsigned polls = 0;
      while (__hip_atomic_load(m.done + tm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < m.target) {
        __builtin_amdgcn_s_sleep(127);
        if (++polls > (1u << 18)) { ok = false; break; }
      }
    }
    if (!ok) *m.err = 1;
    s_ok = ok ? 1 : 0;
  }
  __syncthreads();
  if (!s_ok) return;
  __builtin_amdgcn_s_setprio(3);
  beam_records_body<TOPK, ITEMS, 512, true>(r, b, *reinterpret_cast<RecLds<TOPK, 512>*>(&lds));
}

// ---------------------------------------------------------------------------------
// Small projections (a few hundred 64x64 tiles at most, e.g. [1536,512] x [512,512]): with
// one tile per compute unit the K loop is a chain of short steps whose latency (barrier,
// LDS round trip, global loads) is not hidden by other workgroups.  This form puts KS
// groups of four waves on every tile; group ks multiplies every KS-th slice of 8 k values
// of each 64-deep K step, so a step costs each wave 1/KS of the matrix work, and the KS
// partial tiles are summed through LDS at the end.  K % 64 == 0, aligned operands.
// ---------------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(256 * KS, 1) void dense_f32_small_kernel(DenseArgs p) {
  constexpr int BM = 64, BN = 64, KB = 64, NT = 256 * KS;
  constexpr int A_LD = (BM * KB / 4) / NT;
  constexpr int B_LD = (KB * BN / 4) / NT;
  constexpr int NG = KB / 8;  // slices of 8 k per step
  static_assert(NG % KS == 0 && A_LD >= 1 && B_LD >= 1, "bad split");
  struct Stage {
    float4 As[2][KB / 4][BM + 1];
    float Bs[2][KB][BN];
  };
  union Lds {
    Stage st;
    float red[KS > 1 ? (KS - 1) * 4 * 16 * 64 : 1];  // partial tiles of groups 1..KS-1
  };
  __shared__ Lds lds;
  auto& As = lds.st.As;
  auto& Bs = lds.st.Bs;

  int tile_m, tile_n;
  if (p.band_order == 2) tile_of_block_rowband(blockIdx.x, p.tiles_m, p.tiles_n, tile_m, tile_n);
  else tile_of_block(blockIdx.x, p.tiles_m, p.tiles_n, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ks = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;

  int off_a[A_LD], off_b[B_LD];
#pragma unroll
  for (int i = 0; i < A_LD; ++i) {
    const int f = tid + i * NT;
    const int gm = min(m0 + f / (KB / 4), p.M - 1);
    off_a[i] = gm * p.lda + 4 * (f % (KB / 4));
  }
#pragma unroll
  for (int i = 0; i < B_LD; ++i) {
    const int f = tid + i * NT;
    const int col = min(n0 + 4 * (f % (BN / 4)), p.N - 4);
    off_b[i] = (f / (BN / 4)) * p.ldb + col;
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  float4 ra[A_LD], rb[B_LD];
#define ICAP_SM_LOAD(a_tile_, b_tile_)                                                \
  _Pragma("unroll") for (int i = 0; i < A_LD; ++i) ra[i] =                            \
      *reinterpret_cast<const float4*>((a_tile_) + off_a[i]);                         \
  _Pragma("unroll") for (int i = 0; i < B_LD; ++i) rb[i] =                            \
      *reinterpret_cast<const float4*>((b_tile_) + off_b[i]);
#define ICAP_SM_STORE(buf_)                                                           \
  _Pragma("unroll") for (int i = 0; i < A_LD; ++i) {                                  \
    const int f = tid + i * NT;                                                       \
    As[buf_][f % (KB / 4)][f / (KB / 4)] = ra[i];                                     \
  }                                                                                   \
  _Pragma("unroll") for (int i = 0; i < B_LD; ++i) {                                  \
    const int f = tid + i * NT;                                                       \
    *reinterpret_cast<float4*>(&Bs[buf_][f / (BN / 4)][4 * (f % (BN / 4))]) = rb[i];  \
  }

  const int nk = p.K / KB;
  const float* a_tile = p.A;
  const float* b_tile = p.B;
  ICAP_SM_LOAD(a_tile, b_tile)
  ICAP_SM_STORE(0)
  __syncthreads();
  for (int t = 0; t < nk; ++t) {
    const int buf = t & 1;
    const bool more = t + 1 < nk;
    a_tile += more ? KB : 0;
    b_tile += more ? (size_t)KB * p.ldb : 0;
    ICAP_SM_LOAD(a_tile, b_tile)
#pragma unroll
    for (int gg = 0; gg < NG / KS; ++gg) {
      const int g = gg * KS + ks;
      const float4 a = As[buf][2 * g + half][wm * 32 + l31];
      float b[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) b[q] = Bs[buf][8 * g + 4 * half + q][wn * 32 + l31];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[3], acc, 0, 0, 0);
    }
    ICAP_SM_STORE(buf ^ 1)
    __syncthreads();
  }
#undef ICAP_SM_LOAD
#undef ICAP_SM_STORE

  // sum the partial tiles: groups 1.. park theirs in LDS (staging is idle), group 0 adds
  if (KS > 1) {
    if (ks > 0) {
      float* dst = lds.red + ((ks - 1) * 4 + (wave & 3)) * 16 * 64;
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[r * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (ks > 0) return;
#pragma unroll
    for (int q = 1; q < KS; ++q) {
      const float* src = lds.red + ((q - 1) * 4 + (wave & 3)) * 16 * 64;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += src[r * 64 + lane];
    }
  }
  const int col = n0 + wn * 32 + l31;
  if (col >= p.N) return;
  const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (row < p.M) {
      float v = acc[r] + bv;
      if (p.act == ICAP_ACT_RELU) v = fmaxf(v, 0.f);
      p.C[(size_t)row * p.ldc + col] = v;
    }
  }
}

// Short-K form of the small projections (K = 64*NK <= 512): 16 waves per 64x64 tile and
// EVERY global load of the tile's K loop issued before the first multiply (one float4 of
// A and one of B per thread and K step, NK of each held in registers), so the loop pays the
// memory latency once instead of once per step.  Group ks of four waves multiplies slices
// ks and ks+4 of each 64-deep step; the four partial tiles are summed through LDS.
template <int NK>
__global__ __launch_bounds__(1024, 1) void dense_f32_shortk_kernel(DenseArgs p) {
  constexpr int BM = 64, BN = 64, KB = 64, KS = 4;
  struct Stage {
    float4 As[2][KB / 4][BM + 1];
    float Bs[2][KB][BN];
  };
  union Lds {
    Stage st;
    float red[(KS - 1) * 4 * 16 * 64];
  };
  __shared__ Lds lds;
  auto& As = lds.st.As;
  auto& Bs = lds.st.Bs;

  int tile_m, tile_n;
  if (p.band_order == 2) tile_of_block_rowband(blockIdx.x, p.tiles_m, p.tiles_n, tile_m, tile_n);
  else tile_of_block(blockIdx.x, p.tiles_m, p.tiles_n, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ks = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;

  // thread -> one float4 of the A tile (row a_m, k quad a_q) and one of the B tile
  const int a_q = tid % (KB / 4), a_m = tid / (KB / 4);
  const int b_c = 4 * (tid % (BN / 4)), b_k = tid / (BN / 4);
  const float* a_ptr = p.A + (size_t)min(m0 + a_m, p.M - 1) * p.lda + 4 * a_q;
  const float* b_ptr = p.B + (size_t)b_k * p.ldb + min(n0 + b_c, p.N - 4);
  static_assert(NK == 4 || NK == 8, "short-K form is built for K = 256 and K = 512");
  // (named registers, not arrays: an array of 8 float4 is not split into registers here)
  float4 ra0, ra1, ra2, ra3, ra4, ra5, ra6, ra7, rb0, rb1, rb2, rb3, rb4, rb5, rb6, rb7;
#define ICAP_SK_LOAD(T)                                                 \
  ra##T = *reinterpret_cast<const float4*>(a_ptr + (T) * KB);           \
  rb##T = *reinterpret_cast<const float4*>(b_ptr + (size_t)(T) * KB * p.ldb);
  ICAP_SK_LOAD(0) ICAP_SK_LOAD(1) ICAP_SK_LOAD(2) ICAP_SK_LOAD(3)
  if constexpr (NK == 8) { ICAP_SK_LOAD(4) ICAP_SK_LOAD(5) ICAP_SK_LOAD(6) ICAP_SK_LOAD(7) }
#undef ICAP_SK_LOAD
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  // one barrier per step: buffer buf^1 is rewritten only after the next barrier
#define ICAP_SK_STEP(T)                                                                  \
  {                                                                                      \
    constexpr int buf = (T) & 1;                                                         \
    As[buf][a_q][a_m] = ra##T;                                                           \
    *reinterpret_cast<float4*>(&Bs[buf][b_k][b_c]) = rb##T;                              \
    __syncthreads();                                                                     \
    _Pragma("unroll") for (int gg = 0; gg < 2; ++gg) {                                   \
      const int g = gg * KS + ks;                                                        \
      const float4 a = As[buf][2 * g + half][wm * 32 + l31];                             \
      float b[4];                                                                        \
      _Pragma("unroll") for (int q = 0; q < 4; ++q)                                      \
          b[q] = Bs[buf][8 * g + 4 * half + q][wn * 32 + l31];                           \
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[0], acc, 0, 0, 0);               \
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[1], acc, 0, 0, 0);               \
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[2], acc, 0, 0, 0);               \
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[3], acc, 0, 0, 0);               \
    }                                                                                    \
  }
  ICAP_SK_STEP(0) ICAP_SK_STEP(1) ICAP_SK_STEP(2) ICAP_SK_STEP(3)
  if constexpr (NK == 8) { ICAP_SK_STEP(4) ICAP_SK_STEP(5) ICAP_SK_STEP(6) ICAP_SK_STEP(7) }
#undef ICAP_SK_STEP
  __syncthreads();  // staging is idle from here on
  if (ks > 0) {
    float* dst = lds.red + ((ks - 1) * 4 + (wave & 3)) * 16 * 64;
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[r * 64 + lane] = acc[r];
  }
  __syncthreads();
  if (ks > 0) return;
#pragma unroll
  for (int q = 1; q < KS; ++q) {
    const float* src = lds.red + ((q - 1) * 4 + (wave & 3)) * 16 * 64;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += src[r * 64 + lane];
  }
  const int col = n0 + wn * 32 + l31;
  if (col >= p.N) return;
  const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (row < p.M) {
      float v = acc[r] + bv;
      if (p.act == ICAP_ACT_RELU) v = fmaxf(v, 0.f);
      p.C[(size_t)row * p.ldc + col] = v;
    }
  }
}

// ---------------------------------------------------------------------------------
// LSTM step.  N = B*W rows is small (1536 at the headline shape), so the output is cut
// into many small units to fill 1024 SIMDs evenly: a workgroup owns 32 rows x 32 hidden
// units, and its four waves each own ONE gate (i, f, g, o) of that patch - one 32x32
// accumulator per wave, 768 workgroups = 3 per CU, all resident, so three waves share
// each SIMD's matrix pipe and cover each other's barriers and LDS waits.  The four gate
// tiles meet in LDS for the cell update.
// ---------------------------------------------------------------------------------
struct LstmArgs {
  const float* ctx;    // [N,C]
  const float* emb;    // [N,E] or null
  const float* table;  // [V,E] or null
  const int32_t* ids;  // [N] or null
  const float* h;      // [N,H]
  const float* c;      // [N,H]
  const float* kernel;     // [C+E,4H]
  const float* recurrent;  // [H,4H]
  const float* bias;       // [4H]
  const float* emb_proj;   // [V,4H] = table . kernel[C:C+E] or null (then E > 0)
  float* h_out;
  float* c_out;
  int N, C, E, H, V;
  int tiles_m, tiles_j;
};

constexpr int LSTM_BM = 32;
constexpr int LSTM_BJ = 32;

// CELL selects the cell update applied to the four pre-activation tiles of a unit:
//   CELL_LSTM  i, f, g, o  (Keras LSTMCell)
//   CELL_GRU   z, r, x_h, rec_h with the operands packed by gru_pack_f32 (Keras GRUCell,
//              reset_after=True): h' = z*h + (1-z)*tanh(x_h + r*rec_h)
constexpr int CELL_LSTM = 0;
constexpr int CELL_GRU = 1;

template <int CELL>
__device__ __forceinline__ void cell_update(const LstmArgs& p, size_t o, float z0, float z1, float z2,
                                            float z3) {
  if (CELL == CELL_LSTM) {
    const float ig = fast_sigmoid(z0);
    const float fg = fast_sigmoid(z1);
    const float gg = fast_tanh(z2);
    const float og = fast_sigmoid(z3);
    const float c_new = fg * p.c[o] + ig * gg;
    p.c_out[o] = c_new;
    p.h_out[o] = og * fast_tanh(c_new);
  } else {
    const float zg = fast_sigmoid(z0);
    const float rg = fast_sigmoid(z1);
    const float hh = fast_tanh(z2 + rg * z3);
    p.h_out[o] = zg * p.h[o] + (1.f - zg) * hh;
  }
}

// The same update with the previous state (c for the LSTM, h for the GRU) already in a register.
template <int CELL>
__device__ __forceinline__ void cell_update_pre(const LstmArgs& p, size_t o, float z0, float z1,
                                                float z2, float z3, float prev) {
  if (CELL == CELL_LSTM) {
    const float ig = fast_sigmoid(z0);
    const float fg = fast_sigmoid(z1);
    const float gg = fast_tanh(z2);
    const float og = fast_sigmoid(z3);
    const float c_new = fg * prev + ig * gg;
    p.c_out[o] = c_new;
    p.h_out[o] = og * fast_tanh(c_new);
  } else {
    const float zg = fast_sigmoid(z0);
    const float rg = fast_sigmoid(z1);
    const float hh = fast_tanh(z2 + rg * z3);
    p.h_out[o] = zg * prev + (1.f - zg) * hh;
  }
}

template <bool VEC, int CELL>
__global__ __launch_bounds__(256) void lstm_f32_kernel(LstmArgs p) {
  constexpr int NT = 256;
  constexpr int BM = LSTM_BM, BN = 4 * LSTM_BJ;
  constexpr int B_LD = (BK * BN / 4) / NT;  // 4 float4 of B per thread per K step (A: 1)
  constexpr int A_BYTES = 2 * (BK / 4) * (BM + 1) * 16;
  constexpr int B_BYTES = 2 * BK * BN * 4;
  constexpr int Z_BYTES = 4 * BM * (LSTM_BJ + 1) * 4;
  static_assert(Z_BYTES <= A_BYTES + B_BYTES, "gate exchange reuses the staging buffers");
  __shared__ __attribute__((aligned(16))) char smem[A_BYTES + B_BYTES];
  __shared__ int row_src[BM];  // embedding row of each tile row (-1: zero row)
  typedef float4 (*AsT)[BK / 4][BM + 1];
  typedef float (*BsT)[BK][BN];
  typedef float (*ZsT)[BM][LSTM_BJ + 1];
  AsT As = reinterpret_cast<AsT>(smem);
  BsT Bs = reinterpret_cast<BsT>(smem + A_BYTES);
  ZsT Zs = reinterpret_cast<ZsT>(smem);

  int tile_m, tile_j;
  tile_of_block(blockIdx.x, p.tiles_m, p.tiles_j, tile_m, tile_j);
  const int m0 = tile_m * BM;
  const int j0 = tile_j * LSTM_BJ;
  const int K = p.C + p.E + p.H;
  const int CE = p.C + p.E;
  const int H4 = 4 * p.H;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int gate = tid >> 6;  // wave index = gate
  const int l31 = lane & 31;
  const int half = lane >> 5;

  if (tid < BM) {
    const int gm = m0 + tid;
    int src = -1;
    if (gm < p.N) {
      if (p.ids) {
        const int id = p.ids[gm];
        src = (id >= 0 && id < p.V) ? id : -1;
      } else {
        src = gm;
      }
    }
    row_src[tid] = src;
  }
  __syncthreads();

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  float4 ra, rb[B_LD];
  const float* emb_base = p.ids ? p.table : p.emb;  // unused when E == 0 (emb_proj mode)

  // A[m][k]: k < C from ctx, k < C+E from the embedding row, else from h.  On the VEC
  // path a float4 never straddles two sources because C and E are multiples of 4.
  // Branch-free: pick the source pointer with selects, clamp to a safe address when
  // the element is out of range, zero by select.
  auto a_elem_ptr = [&](int gm, int src, int k, bool& ok) -> const float* {
    const bool in_ctx = k < p.C, in_emb = k < CE;
    ok = (gm < p.N) && (k < K) && (in_ctx || !in_emb || src >= 0);
    const float* q = in_ctx ? p.ctx + (size_t)gm * p.C + k
                            : (in_emb ? emb_base + (size_t)(src < 0 ? 0 : src) * p.E + (k - p.C)
                                      : p.h + (size_t)gm * p.H + (k - CE));
    return ok ? q : p.ctx;
  };
  auto load_a4 = [&](int row, int k) -> float4 {
    const int gm = m0 + row;
    const int src = row_src[row];
    if (VEC) {
      bool ok;
      const float* q = a_elem_ptr(gm, src, k, ok);
      const float4 v = *reinterpret_cast<const float4*>(q);
      return ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float e[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bool ok;
      const float* q = a_elem_ptr(gm, src, k + s, ok);
      const float v = *q;
      e[s] = ok ? v : 0.f;
    }
    return make_float4(e[0], e[1], e[2], e[3]);
  };
  // B[k][c]: tile column c = gate*32 + jj maps to weight column gate*H + j0 + jj.
  auto load_b4 = [&](int k, int c) -> float4 {
    const int g = c / LSTM_BJ;
    const int jj = c % LSTM_BJ;
    const int kk = k < K ? k : 0;
    const float* rowp = (kk < CE) ? p.kernel + (size_t)kk * H4
                                  : p.recurrent + (size_t)(kk - CE) * H4;
    rowp += g * p.H;
    return load4<VEC>(rowp, p.kernel, k < K, j0 + jj, p.H);
  };
  auto load_tiles = [&](int k0) {
    ra = load_a4(tid / (BK / 4), k0 + 4 * (tid % (BK / 4)));
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
      const int f = tid + i * NT;
      rb[i] = load_b4(k0 + f / (BN / 4), 4 * (f % (BN / 4)));
    }
  };
  auto store_tiles = [&](int buf) {
    As[buf][tid % (BK / 4)][tid / (BK / 4)] = ra;
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
      const int f = tid + i * NT;
      *reinterpret_cast<float4*>(&Bs[buf][f / (BN / 4)][4 * (f % (BN / 4))]) = rb[i];
    }
  };

  const int nk = (K + BK - 1) / BK;
  load_tiles(0);
  store_tiles(0);
  __syncthreads();
  for (int t = 0; t < nk; ++t) {
    const int buf = t & 1;
    if (t + 1 < nk) load_tiles((t + 1) * BK);
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      const float4 a = As[buf][2 * g + half][l31];
      float b[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) b[s] = Bs[buf][8 * g + 4 * half + s][gate * 32 + l31];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[3], acc, 0, 0, 0);
    }
    if (t + 1 < nk) store_tiles(buf ^ 1);
    __syncthreads();
  }

  // gate pre-activations (+ bias, + the pre-multiplied embedding row) to LDS:
  // Zs[gate][row][unit]
  {
    const int j = j0 + l31;
    const float bz = (j < p.H) ? p.bias[gate * p.H + j] : 0.f;
    float ep[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      const int src = row_src[row];
      const bool ok = p.emb_proj != nullptr && src >= 0 && j < p.H;
      const float v = *(ok ? p.emb_proj + (size_t)src * H4 + gate * p.H + j : p.bias);
      ep[r] = ok ? v : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      Zs[gate][row][l31] = (acc[r] + ep[r]) + bz;
    }
  }
  __syncthreads();
  // cell update: thread owns unit j of rows rb8, rb8+8, rb8+16, rb8+24
  const int j = j0 + (tid & 31);
  if (j >= p.H) return;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = (tid >> 5) + 8 * q;
    const int gm = m0 + row;
    if (gm < p.N)
      cell_update<CELL>(p, (size_t)gm * p.H + j, Zs[0][row][tid & 31], Zs[1][row][tid & 31],
                        Zs[2][row][tid & 31], Zs[3][row][tid & 31]);
  }
}

// ---------------------------------------------------------------------------------
// LSTM step, wide form: the gate product over [ctx | h] (the token part comes in as
// emb_proj rows) for shapes whose 96-row x 32-unit patches tile the chip evenly: one
// workgroup of 12 waves per compute unit (wave = 32 rows x one gate), three waves per
// SIMD.  Against the 32-row kernel this moves 2.1x fewer operand bytes from L2 per
// flop (the B tile is shared by three row groups) and halves the barriers (BK = 64,
// LDS 114 KB, affordable with one workgroup per CU).  Staging is split by role: waves
// 0-7 fetch the B tile, waves 8-11 the A tile, so every wave issues one uniform run of
// loads per K step.  Requires C % 64 == 0, H % 64 == 0, 16-byte aligned operands.
// ---------------------------------------------------------------------------------
constexpr int LW_BM = 96;
constexpr int LW_BK = 64;
constexpr int LW_NT = 768;

template <int CELL>
__global__ __launch_bounds__(LW_NT, 1) __attribute__((amdgpu_waves_per_eu(3, 3))) void lstm_f32_wide_kernel(LstmArgs p) {
  constexpr int BM = LW_BM, BN = 128, LBK = LW_BK, KQ = LBK / 4;
  constexpr int A_BYTES = 2 * KQ * (BM + 1) * 16;
  constexpr int B_BYTES = 2 * LBK * BN * 4;
  constexpr int Z_BYTES = 4 * BM * 33 * 4;
  static_assert(Z_BYTES <= A_BYTES + B_BYTES, "gate exchange reuses the staging buffers");
  constexpr int B_F4 = LBK * BN / 4;                  // float4 per B tile
  constexpr int B_LD = (B_F4 + LW_NT - 1) / LW_NT;    // per thread per K step (tail clamped)
  constexpr int A_LD = (BM * KQ) / LW_NT;
  static_assert(A_LD * LW_NT == BM * KQ, "A tile must divide evenly over the threads");
  __shared__ __attribute__((aligned(16))) char smem[A_BYTES + B_BYTES + BM * 4];
  typedef float4 (*AsT)[KQ][BM + 1];
  typedef float (*BsT)[LBK][BN];
  typedef float (*ZsT)[BM][33];
  AsT As = reinterpret_cast<AsT>(smem);
  BsT Bs = reinterpret_cast<BsT>(smem + A_BYTES);
  ZsT Zs = reinterpret_cast<ZsT>(smem);
  int* row_src = reinterpret_cast<int*>(smem + A_BYTES + B_BYTES);

  ICAP_STAMP_F32(0)
  int tile_m, tile_j;
  tile_of_block(blockIdx.x, p.tiles_m, p.tiles_j, tile_m, tile_j);
  const int m0 = tile_m * BM;
  const int j0 = tile_j * 32;
  const int K = p.C + p.H;
  const int H4 = 4 * p.H;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int rg = wave >> 2;   // row group: rows 32*rg .. 32*rg+31 of the patch
  const int gate = wave & 3;
  const int l31 = lane & 31;
  const int half = lane >> 5;

  if (tid < BM) {
    const int gm = m0 + tid;
    int src = -1;
    if (gm < p.N && p.ids) {
      const int id = p.ids[gm];
      src = (id >= 0 && id < p.V) ? id : -1;
    }
    row_src[tid] = src;
  }
  __syncthreads();

  // the token-only part of the pre-activations: issued now, consumed in the epilogue
  const int j = j0 + l31;
  const float bz = p.bias[gate * p.H + j];
  float ep[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = rg * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    const int src = row_src[row];
    const bool ok = p.emb_proj != nullptr && src >= 0;
    const float v = *(ok ? p.emb_proj + (size_t)src * H4 + gate * p.H + j : p.bias);
    ep[r] = ok ? v : 0.f;
  }

  // the previous state of this thread's (row, unit) pairs for the cell update, fetched ahead
  // of the K loop too: loaded inside the update loop, every iteration's load would have to
  // stay behind the previous iteration's stores (the buffers may alias) and wait out its own
  // memory latency
  constexpr int NQ = BM * 32 / LW_NT;
  float st_pre[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int gm = min(m0 + (tid >> 5) + (LW_NT / 32) * q, p.N - 1);
    const size_t o = (size_t)gm * p.H + j0 + (tid & 31);
    st_pre[q] = (CELL == CELL_LSTM) ? p.c[o] : p.h[o];
  }

  // Staging offsets (elements).  The B tile does not divide evenly over 768 threads: the
  // surplus threads of the last pass repeat the tile's last vector (same address, same
  // LDS slot, same value), which keeps the loads free of control flow.
  int off_b[B_LD], lds_b[B_LD], off_a_ctx[A_LD], off_a_h[A_LD];
#pragma unroll
  for (int i = 0; i < B_LD; ++i) {
    const int f = min(tid + i * LW_NT, B_F4 - 1);
    const int c = 4 * (f % (BN / 4));
    off_b[i] = (f / (BN / 4)) * H4 + (c / 32) * p.H + j0 + (c % 32);
    lds_b[i] = (f / (BN / 4)) * BN + c;
  }
#pragma unroll
  for (int i = 0; i < A_LD; ++i) {
    const int f = tid + i * LW_NT;
    const int gm = min(m0 + f / KQ, p.N - 1);  // clamped rows are computed but never stored
    off_a_ctx[i] = gm * p.C + 4 * (f % KQ);
    off_a_h[i] = gm * p.H + 4 * (f % KQ);
  }

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  float4 ra[A_LD], rb[B_LD];
#define ICAP_LW_LOAD(K0)                                                                      \
  {                                                                                           \
    const int k0_ = (K0);                                                                     \
    const bool in_ctx_ = k0_ < p.C; /* uniform: C is a multiple of the K step */              \
    const float* b_ = in_ctx_ ? p.kernel + (size_t)k0_ * H4                                   \
                              : p.recurrent + (size_t)(k0_ - p.C) * H4;                       \
    const float* a_ = in_ctx_ ? p.ctx + k0_ : p.h + (k0_ - p.C);                              \
    _Pragma("unroll") for (int i = 0; i < B_LD; ++i) rb[i] =                                  \
        *reinterpret_cast<const float4*>(b_ + off_b[i]);                                      \
    _Pragma("unroll") for (int i = 0; i < A_LD; ++i) ra[i] =                                  \
        *reinterpret_cast<const float4*>(a_ + (in_ctx_ ? off_a_ctx[i] : off_a_h[i]));         \
  }
#define ICAP_LW_STORE(BUF)                                                                    \
  {                                                                                           \
    _Pragma("unroll") for (int i = 0; i < B_LD; ++i)                                          \
        *reinterpret_cast<float4*>(&Bs[BUF][0][0] + lds_b[i]) = rb[i];                        \
    _Pragma("unroll") for (int i = 0; i < A_LD; ++i) {                                        \
      const int f = tid + i * LW_NT;                                                          \
      As[BUF][f % KQ][f / KQ] = ra[i];                                                        \
    }                                                                                         \
  }

  // One barrier per K step, below the next step's loads (see dense_f32_fast_kernel): the
  // prefetch stays in flight during the multiplies, which matters here because there is no
  // second workgroup on the compute unit to cover a memory latency.
  // (Device-clock stamps at the headline shape, kernel alone: prologue 3.1 us, K loop 42.3 us,
  // cell update 1.0 us.  Timing experiments on the loop: the multiplies alone 36.2 us (a bare
  // loop of the same multiplies reaches 0.98 of the matrix peak in a long launch), with the
  // LDS reads 37.0, with the barrier 36.2; the global loads or the LDS stores, either or
  // both, add 5 us whether the stores sit behind the multiplies, among them (with the
  // registers one tile ahead) or in front; two accumulator chains per wave change nothing.)
  const int nk = K / LBK;
  ICAP_LW_LOAD(0)
  ICAP_LW_STORE(0)
  ICAP_STAMP_F32(2)
  for (int t = 0; t < nk; ++t) {
    const int buf = t & 1;
    ICAP_LW_LOAD(min(t + 1, nk - 1) * LBK)  // the last step re-loads its own tile: no branch
    __syncthreads();
#pragma unroll
    for (int g = 0; g < LBK / 8; ++g) {
      const float4 a = As[buf][2 * g + half][rg * 32 + l31];
      float b[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) b[s] = Bs[buf][8 * g + 4 * half + s][gate * 32 + l31];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[3], acc, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < A_LD; ++i) asm volatile("" : "+v"(ra[i].x), "+v"(ra[i].y), "+v"(ra[i].z), "+v"(ra[i].w));  // uses stay below the multiplies
#pragma unroll
    for (int i = 0; i < B_LD; ++i) asm volatile("" : "+v"(rb[i].x), "+v"(rb[i].y), "+v"(rb[i].z), "+v"(rb[i].w));
    ICAP_LW_STORE(buf ^ 1)
  }
  __syncthreads();  // the gate exchange reuses the staging buffers
  ICAP_STAMP_F32(3)
#undef ICAP_LW_LOAD
#undef ICAP_LW_STORE

  // gate pre-activations to LDS (the staging buffers are idle after the last barrier)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = rg * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    Zs[gate][row][l31] = (acc[r] + ep[r]) + bz;
  }
  __syncthreads();
  // cell update: 96 x 32 (row, unit) pairs over 768 threads
  const int u = tid & 31;
#pragma unroll
  for (int q = 0; q < BM * 32 / LW_NT; ++q) {
    const int row = (tid >> 5) + (LW_NT / 32) * q;
    const int gm = m0 + row;
    if (gm < p.N)
      cell_update_pre<CELL>(p, (size_t)gm * p.H + j0 + u, Zs[0][row][u], Zs[1][row][u], Zs[2][row][u],
                            Zs[3][row][u], st_pre[q]);
  }
  ICAP_STAMP_F32(1)
}

// ---------------------------------------------------------------------------------
// LSTM / GRU step for small batches (a few dozen rows): the gate product is a weight
// stream, and with so few 32-row patches the K loop of the tiled kernels is a chain of
// exposed memory latencies on a handful of compute units.  Here a workgroup of 16 waves
// owns 32 rows x 8 units (one 32-column MFMA block = 4 gates x 8 units), so even 32 rows
// spread over H/8 compute units; wave q multiplies the q-th sixteenth of K, operands go
// from global memory straight into the MFMA registers (lane = row / column, its half-wave
// picks 4 of the 8 k values of a group) with all of a wave's loads in flight before its
// multiplies; the sixteen partial blocks meet in LDS for the cell update.  Token part from
// emb_proj rows.  Requires C % 4 == 0, (C + H) % 128 == 0, H % 8 == 0, aligned operands.
// ---------------------------------------------------------------------------------
template <int CELL, int NG>
__global__ __launch_bounds__(1024, 1) void lstm_f32_skinny_kernel(LstmArgs p) {
  __shared__ float Zs[16][32][33];  // [K slice][row][gate*8 + unit]
  __shared__ float Z2[32][33];
  const int tile_m = blockIdx.x % p.tiles_m, tile_j = blockIdx.x / p.tiles_m;
  const int m0 = tile_m * 32, j0 = tile_j * 8;
  const int K = p.C + p.H, H4 = 4 * p.H;
  const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int gm = min(m0 + l31, p.N - 1);      // clamped rows are computed, never stored
  const int col = (l31 >> 3) * p.H + j0 + (l31 & 7);
  const int k_begin = q * (K / 16);

  float4 a[NG];
  float b[NG][4];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int k = k_begin + 8 * g + 4 * half;
    const bool in_ctx = k < p.C;  // C % 4 == 0: a 4-k slice never straddles the two sources
    const float* ap = in_ctx ? p.ctx + (size_t)gm * p.C + k : p.h + (size_t)gm * p.H + (k - p.C);
    a[g] = *reinterpret_cast<const float4*>(ap);
    const float* bp = (in_ctx ? p.kernel + (size_t)k * H4 : p.recurrent + (size_t)(k - p.C) * H4) + col;
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
  {  // thread = (row, column of the block): sum of the sixteen K slices
    const int row = tid >> 5, c = tid & 31;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += Zs[i][row][c];
    Z2[row][c] = sum;
  }
  __syncthreads();
  if (tid >= 256) return;
  const int row = tid >> 3, u = tid & 7;
  const int m = m0 + row, j = j0 + u;
  if (m >= p.N) return;
  int id = p.ids[m];
  const bool tok = id >= 0 && id < p.V;
  id = tok ? id : 0;
  float z[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float e = p.emb_proj[(size_t)id * H4 + g * p.H + j];
    z[g] = (Z2[row][g * 8 + u] + (tok ? e : 0.f)) + p.bias[g * p.H + j];
  }
  cell_update<CELL>(p, (size_t)m * p.H + j, z[0], z[1], z[2], z[3]);
}

// Dense layer for small batches (M up to a few dozen rows), same scheme as the small-batch
// LSTM step: workgroup = 32 rows x 32 columns, wave q multiplies the q-th sixteenth of K with
// all its loads in flight, the sixteen partial blocks are summed through LDS.  The weights
// are streamed once by N/32 workgroups.  K % 128 == 0, N % 4 == 0, aligned operands.
template <int NG>
__global__ __launch_bounds__(1024, 1) void dense_f32_skinny_kernel(DenseArgs p) {
  __shared__ float Zs[16][32][33];
  const int tile_m = blockIdx.x % p.tiles_m, tile_n = blockIdx.x / p.tiles_m;
  const int m0 = tile_m * 32, n0 = tile_n * 32;
  const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int gm = min(m0 + l31, p.M - 1);
  const int col = min(n0 + l31, p.N - 1);   // clamped rows / columns are computed, never stored
  const int k_begin = q * (p.K / 16);
  float4 a[NG];
  float b[NG][4];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int k = k_begin + 8 * g + 4 * half;
    a[g] = *reinterpret_cast<const float4*>(p.A + (size_t)gm * p.lda + k);
    const float* bp = p.B + (size_t)k * p.ldb + col;
#pragma unroll
    for (int s = 0; s < 4; ++s) b[g][s] = bp[(size_t)s * p.ldb];
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
  const int row = tid >> 5, c = tid & 31;
  const int m = m0 + row, n = n0 + c;
  if (m >= p.M || n >= p.N) return;
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) sum += Zs[i][row][c];
  float v = sum + (p.bias ? p.bias[n] : 0.f);
  if (p.act == ICAP_ACT_RELU) v = fmaxf(v, 0.f);
  p.C[(size_t)m * p.ldc + n] = v;
}

template <int BM, int BN, int WM, int WN, int MINW>
int launch_dense(const DenseArgs& a, bool vec, hipStream_t s) {
  DenseArgs p = a;
  p.tiles_m = ceil_div(p.M, BM);
  p.tiles_n = ceil_div(p.N, BN);
  const int grid = p.tiles_m * p.tiles_n;
  const bool fast = vec && (p.K % BK == 0) && ((long long)p.M * p.lda < (1LL << 31)) &&
                    ((long long)p.K * p.ldb < (1LL << 31));
  if (fast)
    hipLaunchKernelGGL((dense_f32_fast_kernel<BM, BN, WM, WN, MINW>), dim3(grid),
                       dim3(WM * WN * 64), 0, s, p);
  else if (vec)
    hipLaunchKernelGGL((dense_f32_kernel<BM, BN, WM, WN, true>), dim3(grid), dim3(WM * WN * 64),
                       0, s, p);
  else
    hipLaunchKernelGGL((dense_f32_kernel<BM, BN, WM, WN, false>), dim3(grid),
                       dim3(WM * WN * 64), 0, s, p);
  ICAP_LAUNCH_CHECK("dense_f32_kernel");
  return ICAP_OK;
}

}  // namespace

int dense_f32_strided(const float* x, int lda, const float* w, const float* bias, float* y, int M,
                      int K, int N, int act, hipStream_t s);
static int dense_f32_impl(const float* x, int lda, const float* w, const float* bias, float* y, int M,
                          int K, int N, int act, bool throughput, hipStream_t s);

int dense_f32(const float* x, const float* w, const float* bias, float* y, int M, int K, int N,
              int act, hipStream_t s) {
  return dense_f32_strided(x, K, w, bias, y, M, K, N, act, s);
}

// dense_f32 for callers whose products come in long runs of equal launches (the transformer
// stack: six layers of them per position) rather than on a decode step's critical path: where
// dense_f32 picks a split-K form for its short latency, this one picks the plain tile kernel
// when that is the faster launch.  The two forms sum in different orders, so a caller sticks to
// one of the entry points.
int dense_f32_throughput(const float* x, const float* w, const float* bias, float* y, int M, int K,
                         int N, int act, hipStream_t s) {
  return dense_f32_impl(x, K, w, bias, y, M, K, N, act, true, s);
}

// x rows lda floats apart (lda >= K): e.g. every W-th row of a [B*W,K] matrix with lda = W*K.
int dense_f32_strided(const float* x, int lda, const float* w, const float* bias, float* y, int M,
                      int K, int N, int act, hipStream_t s) {
  return dense_f32_impl(x, lda, w, bias, y, M, K, N, act, false, s);
}

static int dense_f32_impl(const float* x, int lda, const float* w, const float* bias, float* y, int M,
                          int K, int N, int act, bool throughput, hipStream_t s) {
  ICAP_REQUIRE(x && w && y, "dense_f32: null pointer");
  ICAP_REQUIRE(lda >= K && lda % 4 == K % 4, "dense_f32: bad row stride %d for K = %d", lda, K);
  ICAP_REQUIRE(M > 0 && K > 0 && N > 0, "dense_f32: M, K, N must be positive (got %d, %d, %d)", M,
               K, N);
  ICAP_REQUIRE(act == ICAP_ACT_NONE || act == ICAP_ACT_RELU, "dense_f32: unknown activation %d",
               act);
  DenseArgs a{x, w, bias, y, M, N, K, lda, N, N, act, 0, 0, 0, nullptr, nullptr, -1};
  {
    // Tile order (tuning knob ICAP_TILE_ORDER=col|band overrides): a band of panels per XCD
    // when the A operand is too large to stay in one 4 MB L2 next to a B panel.
    static const char* order_env = getenv("ICAP_TILE_ORDER");
    const size_t a_bytes = (size_t)M * K * sizeof(float);
    a.band_order = a_bytes > (size_t)(5 << 19) ? 1 : 0;  // > 2.5 MB
    if (order_env) a.band_order = order_env[0] == 'b' ? 1 : 0;
  }
  const bool vec = (K % 4 == 0) && (N % 4 == 0) && aligned16(x) && aligned16(w);
  {
    // a few dozen rows: the small-batch form (tuning knob ICAP_SKINNY=0 disables)
    static const char* sk_env = getenv("ICAP_SKINNY");
    const int ng = K / 128;
    if (vec && M <= 64 && K % 128 == 0 && lda % 4 == 0 && (ng == 2 || ng == 4 || ng == 6 || ng == 8) &&
        (long long)K * N < (1LL << 31) && !(sk_env && sk_env[0] == '0')) {
      a.tiles_m = ceil_div(M, 32);
      a.tiles_n = ceil_div(N, 32);
      const dim3 g(a.tiles_m * a.tiles_n), b(1024);
      if (ng == 2) hipLaunchKernelGGL(dense_f32_skinny_kernel<2>, g, b, 0, s, a);
      else if (ng == 4) hipLaunchKernelGGL(dense_f32_skinny_kernel<4>, g, b, 0, s, a);
      else if (ng == 6) hipLaunchKernelGGL(dense_f32_skinny_kernel<6>, g, b, 0, s, a);
      else hipLaunchKernelGGL(dense_f32_skinny_kernel<8>, g, b, 0, s, a);
      ICAP_LAUNCH_CHECK("dense_f32_skinny_kernel");
      return ICAP_OK;
    }
  }
  // 128x128 tiles (8 waves, 32x64 per wave, 2 workgroups = 4 waves per SIMD) when they
  // fill the chip at least 1.5 times, 64x64 tiles (4 waves) for the small projections.
  const long big_tiles = (long)ceil_div(M, 128) * ceil_div(N, 128);
  {
    // just under one round of 128x128 tiles (two per compute unit): half of the units would run
    // one tile, the other half two - 64x128 tiles fill them evenly ([3072,512] x [512,2048]:
    // the feed-forward network 16.5 against 18.2 ms per decode of the
    // 6-layer stack).  Tuning knob ICAP_MID_FORM=0 (128x128) | 1 (64x128) | 2 (64x64).
    static const char* mid_env = getenv("ICAP_MID_FORM");
    const int mid = mid_env ? atoi(mid_env) : 1;
    // (768 tiles, a round and a half: no difference, 90.2 against 90.0 ms of projections per decode)
    if (mid && vec && K % BK == 0 && big_tiles >= 384 && big_tiles < 512) {
      if (mid == 1) return launch_dense<64, 128, 2, 4, 2>(a, vec, s);
      return launch_dense<64, 64, 2, 2, 2>(a, vec, s);
    }
  }
  if (big_tiles >= 384) return launch_dense<128, 128, 4, 2, 2>(a, vec, s);
  {
    // few tiles: the split-K form of the 64x64 tile (tuning knob ICAP_SMALL_KS = 1|2|4)
    static const char* ks_env = getenv("ICAP_SMALL_KS");
    const int ks = ks_env ? atoi(ks_env) : 2;
    const long tiles64 = (long)ceil_div(M, 64) * ceil_div(N, 64);
    // long-K products with few tiles (e.g. [3072,4096] x [4096,512]): 64x128 tiles of the
    // main kernel beat the split-K form (93 against 86 TFLOP/s)
    if (vec && K >= 1024 && K % BK == 0 && tiles64 > 128 && tiles64 <= 512 && !ks_env)
      return launch_dense<64, 128, 2, 4, 2>(a, vec, s);
    // (dense_f32_throughput; more 64x64 tiles than compute units and K = 512 or 256: the plain
    // 64x64 kernel below is ahead of the 1024-thread split-K form - [3072,512] x [512,512],
    // 384 tiles: 22.8 against 26.4 us per launch)
    const bool plain64 = throughput && tiles64 > 256 && (K == 512 || K == 256) && !ks_env;
    if (ks > 1 && !plain64 && vec && K % 64 == 0 && N >= 4 && tiles64 <= 512 &&
        (long long)M * lda < (1LL << 31) && (long long)K * N < (1LL << 31)) {
      a.tiles_m = ceil_div(M, 64);
      a.tiles_n = ceil_div(N, 64);
      // row bands per XCD when A is the larger operand (tuning knob ICAP_SMALL_ORDER=col|row)
      static const char* so_env = getenv("ICAP_SMALL_ORDER");
      a.band_order = (M >= N) ? 2 : 0;
      if (so_env) a.band_order = so_env[0] == 'r' ? 2 : 0;
      static const char* sk_env = getenv("ICAP_SHORTK");  // tuning knob: "0" disables
      const bool shortk = !(sk_env && sk_env[0] == '0');
      if (shortk && K == 512)
        hipLaunchKernelGGL(dense_f32_shortk_kernel<8>, dim3((unsigned)tiles64), dim3(1024), 0, s, a);
      else if (shortk && K == 256)
        hipLaunchKernelGGL(dense_f32_shortk_kernel<4>, dim3((unsigned)tiles64), dim3(1024), 0, s, a);
      else if (ks >= 4)
        hipLaunchKernelGGL(dense_f32_small_kernel<4>, dim3((unsigned)tiles64), dim3(1024), 0, s, a);
      else
        hipLaunchKernelGGL(dense_f32_small_kernel<2>, dim3((unsigned)tiles64), dim3(512), 0, s, a);
      ICAP_LAUNCH_CHECK("dense_f32_small_kernel");
      return ICAP_OK;
    }
  }
  return launch_dense<64, 64, 2, 2, 2>(a, vec, s);
}

// dense_f32_strided on the 128x128 tile kernel whatever the shape (the kernel and the sums of
// dense_topk_f32), switched on the device: nothing runs unless *run_flag != 0.  The engine
// enqueues it behind the fused projection + wide beam step, which raises the flag when its
// records cannot prove the step's result.
bool dense_f32_flagged_supported(const float* x, const float* w, int lda, int M, int K, int N) {
  return M > 0 && K > 0 && N > 0 && K % BK == 0 && N % 4 == 0 && lda % 4 == 0 && aligned16(x) &&
         aligned16(w) && (long long)M * lda < (1LL << 31) && (long long)K * N < (1LL << 31);
}

int dense_f32_flagged(const float* x, int lda, const float* w, const float* bias, float* y, int M,
                      int K, int N, const int* run_flag, hipStream_t s) {
  ICAP_REQUIRE(x && w && y && run_flag, "dense_f32_flagged: null pointer");
  ICAP_REQUIRE(lda >= K && dense_f32_flagged_supported(x, w, lda, M, K, N),
               "dense_f32_flagged: shape not served (M=%d K=%d N=%d lda=%d)", M, K, N, lda);
  DenseArgs a{x, w, bias, y, M, N, K, lda, N, N, ICAP_ACT_NONE, 0, 0, 0, nullptr, nullptr, -1};
  a.band_order = (size_t)M * K * sizeof(float) > (size_t)(5 << 19) ? 1 : 0;
  a.band_width = band_width_for(K, a.tiles_n);
  a.run_flag = run_flag;
  return launch_dense<128, 128, 4, 2, 2>(a, true, s);
}

// Records per row and floats per record of dense_topk_f32 (0 when the shape is not served).
int dense_topk_half_tiles(int N) { return N > 0 ? 2 * ceil_div(N, 128) : 0; }
int dense_topk_record_floats(int top_k) { return (top_k == 1 || top_k == 3 || top_k == 5) ? 2 + 2 * top_k : 0; }

bool dense_topk_supported(const float* x, const float* w, int M, int K, int N, int top_k) {
  return dense_topk_record_floats(top_k) != 0 && M > 0 && K > 0 && N > 0 && (K % BK == 0) &&
         (N % 4 == 0) && N >= 4 && aligned16(x) && aligned16(w) &&
         (long long)M * K < (1LL << 31) && (long long)K * N < (1LL << 31);
}

int dense_topk_f32_strided(const float* x, int lda, const float* w, const float* bias, int M, int K,
                           int N, int eos, int top_k, float* records, float* eos_logit,
                           const LogitMasks* masks, hipStream_t s);

int dense_topk_f32(const float* x, const float* w, const float* bias, int M, int K, int N,
                   int eos, int top_k, float* records, float* eos_logit, const LogitMasks* masks,
                   hipStream_t s) {
  return dense_topk_f32_strided(x, K, w, bias, M, K, N, eos, top_k, records, eos_logit, masks, s);
}

int dense_topk_f32_strided(const float* x, int lda, const float* w, const float* bias, int M, int K,
                           int N, int eos, int top_k, float* records, float* eos_logit,
                           const LogitMasks* masks, hipStream_t s) {
  ICAP_REQUIRE(x && w && records && eos_logit, "dense_topk_f32: null pointer");
  ICAP_REQUIRE(lda >= K && lda % 4 == 0 && (long long)M * lda < (1LL << 31),
               "dense_topk_f32: bad row stride %d", lda);
  ICAP_REQUIRE(dense_topk_supported(x, w, M, K, N, top_k),
               "dense_topk_f32: needs top_k in {1,3,5}, K %% 32 == 0, N %% 4 == 0 and 16-byte "
               "aligned operands (M=%d K=%d N=%d top_k=%d)", M, K, N, top_k);
  ICAP_REQUIRE(aligned16(records), "dense_topk_f32: records must be 16-byte aligned");
  DenseArgs a{x, w, bias, nullptr, M, N, K, lda, N, N, ICAP_ACT_NONE, 0, 0, 0, records, eos_logit, eos};
  if (masks) a.masks = *masks;
  a.tiles_m = ceil_div(M, 128);
  a.tiles_n = ceil_div(N, 128);
  a.band_order = (size_t)M * K * sizeof(float) > (size_t)(5 << 19) ? 1 : 0;
  a.band_width = band_width_for(K, a.tiles_n);
  // (96-row tiles would leave fewer compute units idle at the end - 4.94 tiles per CU
  // instead of 3.70 at 1536 x 10000 - but their six waves spread unevenly over the four
  // SIMDs: measured 197 us against 146 us)
  const dim3 grid(a.tiles_m * a.tiles_n), block(512);
  if (top_k == 1)
    hipLaunchKernelGGL((dense_f32_fast_kernel<128, 128, 4, 2, 2, 1>), grid, block, 0, s, a);
  else if (top_k == 3)
    hipLaunchKernelGGL((dense_f32_fast_kernel<128, 128, 4, 2, 2, 3>), grid, block, 0, s, a);
  else
    hipLaunchKernelGGL((dense_f32_fast_kernel<128, 128, 4, 2, 2, 5>), grid, block, 0, s, a);
  ICAP_LAUNCH_CHECK("dense_f32_fast_kernel<topk>");
  return ICAP_OK;
}

// dense_topk_f32_strided + beam_step_from_records_f32 in one launch (dense_topk_beam_kernel).
// done: [ceil(M/128)] device counters, zero before the first launch of a sequence; launch_no
// counts the launches that have used them since (1, 2, ...).  err: device word, raised if the
// step of some image could not be taken.
bool dense_topk_beam_supported(int M, int W, int half_tiles, int top_k, int time, int compact_first) {
  const int n_rec = (time <= 0 ? 1 : W) * half_tiles;
  return W >= 1 && W <= top_k && n_rec <= 2 * 512 && 512 >= W * (W + 2) &&
         (compact_first ? time <= 0 : true) && M > 0;
}

int dense_topk_beam_f32(const float* x, int lda, const float* w, const float* bias, int M, int K, int N,
                        int eos, int top_k, float* records, float* eos_logit,
                        const float* log_probs, const uint8_t* finished, const int32_t* lengths,
                        int time, double length_penalty_weight, float* scores, int32_t* predicted_ids,
                        int32_t* parent_ids, float* next_log_probs, uint8_t* next_finished,
                        int32_t* next_lengths, int B, int W, int compact_first, unsigned* done,
                        unsigned launch_no, int* err, hipStream_t s) {
  ICAP_REQUIRE(x && w && records && eos_logit && done && err, "dense_topk_beam_f32: null pointer");
  ICAP_REQUIRE(log_probs && finished && lengths && scores && predicted_ids && parent_ids &&
                   next_log_probs && next_finished && next_lengths,
               "dense_topk_beam_f32: null beam-state pointer");
  ICAP_REQUIRE(lda >= K && lda % 4 == 0 && (long long)M * lda < (1LL << 31),
               "dense_topk_beam_f32: bad row stride %d", lda);
  ICAP_REQUIRE(dense_topk_supported(x, w, M, K, N, top_k) && aligned16(records),
               "dense_topk_beam_f32: projection shape not served (M=%d K=%d N=%d top_k=%d)", M, K, N,
               top_k);
  ICAP_REQUIRE(B > 0 && M == B * (compact_first ? 1 : W) && launch_no >= 1 &&
                   (long long)W * N < (1LL << 31),
               "dense_topk_beam_f32: %d rows do not belong to %d images of beam width %d", M, B, W);
  DenseArgs a{x, w, bias, nullptr, M, N, K, lda, N, N, ICAP_ACT_NONE, 0, 0, 0, records, eos_logit, eos};
  a.tiles_m = ceil_div(M, 128);
  a.tiles_n = ceil_div(N, 128);
  a.band_order = 1;  // (tile rows complete one after the other on every XCD)
  const int half_tiles = 2 * a.tiles_n;
  ICAP_REQUIRE(dense_topk_beam_supported(M, W, half_tiles, top_k, time, compact_first),
               "dense_topk_beam_f32: %d rows x %d records exceed one workgroup", W, half_tiles);
  RecArgs ra;
  ra.rec = records; ra.eos_x = eos_logit; ra.HT = half_tiles;
  ra.rpi = compact_first ? 1 : W;
  ra.log_probs = log_probs; ra.finished = finished; ra.lengths = lengths;
  ra.time = time; ra.eos = eos;
  ra.scores = scores; ra.ids = predicted_ids; ra.parents = parent_ids;
  ra.next_lp = next_log_probs; ra.next_fin = next_finished; ra.next_len = next_lengths;
  ra.B = B; ra.W = W; ra.V = N;
  ra.pen = make_penalty(length_penalty_weight);

  BeamMergeArgs m;
  m.done = done;
  m.target = launch_no * (unsigned)a.tiles_n;
  m.err = err;
  const int total = a.tiles_m * a.tiles_n;
  m.q = total >> 3;
  m.r = total & 7;
  const int slots = ceil_div(B, 8);        // image workgroups per XCD
  m.chunks = a.tiles_m < 8 ? a.tiles_m : 8;
  m.cb = ceil_div(slots, m.chunks);
  // A run goes behind the last tile row its images need by `lag` positions: the 64
  // workgroups an XCD holds at a time (32 compute units x 2) plus the skew between the XCDs'
  // runs of tile numbers, so that those tiles have finished when the run starts.
  static const char* lag_env = getenv("ICAP_BEAM_LAG");
  const int lag = lag_env ? atoi(lag_env) : 80;
  const int pw = (a.tiles_n + 7) >> 3;
  const int rows_per_image = compact_first ? 1 : W;
  for (int c = 0; c < 8; ++c) {
    long long last_image = (long long)(c + 1) * m.cb * 8 - 1;
    if (last_image > B - 1) last_image = B - 1;
    const int tmh = (int)(((last_image + 1) * rows_per_image - 1) / 128);
    const int at = (tmh + 1) * pw + lag;
    m.ins[c] = at < m.q + 1 ? at : m.q + 1;
  }
  const int items = ceil_div((time <= 0 ? 1 : W) * half_tiles, 512);
  const dim3 grid(8 * (m.q + 1 + m.chunks * m.cb)), block(512);
#define ICAP_TB_LAUNCH(K_)                                                                       \
  if (items <= 1) hipLaunchKernelGGL((dense_topk_beam_kernel<K_, 1>), grid, block, 0, s, a, ra, m); \
  else hipLaunchKernelGGL((dense_topk_beam_kernel<K_, 2>), grid, block, 0, s, a, ra, m);
  if (top_k == 1) { ICAP_TB_LAUNCH(1) }
  else if (top_k == 3) { ICAP_TB_LAUNCH(3) }
  else { ICAP_TB_LAUNCH(5) }
#undef ICAP_TB_LAUNCH
  ICAP_LAUNCH_CHECK("dense_topk_beam_kernel");
  return ICAP_OK;
}

// dense_topk_f32_strided whose launch also carries an attention step with a folded-in query
// projection (see dense_topk_attention_kernel): rows per group R <= 5 with top_k = 1 (R = 1),
// 3 (R <= 3) or 5, U <= 512, and the attention's LDS needs within the projection's tile buffers.
bool dense_topk_attention_supported(int R, int L, int U, int D, int H, int top_k) {
  if (R < 1 || R > 5 || L < 1 || U < 4 || U > 512 || U % 4 || D < 4 || D % 4 || H < 4 || H % 4)
    return false;
  if (top_k != (R <= 1 ? 1 : (R <= 3 ? 3 : 5))) return false;
  return sizeof(float) * attention_fq_lds_floats(512, R, R <= 3 ? 2 : 1, L, U, D, H) <=
         sizeof(FastLds<128, 128, 4, 2, 1>::U);
}

int dense_topk_attention_f32(const float* x, int lda, const float* w, const float* bias, int M, int K,
                             int N, int eos, int top_k, float* records, float* eos_logit,
                             const LogitMasks* masks, const float* pf, const float* features,
                             const float* att_v, float att_bv, const float* state,
                             const float* gate_w, float gate_b, const float* q_w, const float* q_b,
                             float* ctx, int G, int R, int stride_g, int stride_k, int L, int U,
                             int D, int H, int att_blocks, hipStream_t s) {
  ICAP_REQUIRE(x && w && records && eos_logit, "dense_topk_attention_f32: null pointer");
  ICAP_REQUIRE(pf && features && att_v && state && gate_w && q_w && ctx,
               "dense_topk_attention_f32: null attention pointer");
  ICAP_REQUIRE(lda >= K && lda % 4 == 0 && (long long)M * lda < (1LL << 31),
               "dense_topk_attention_f32: bad row stride %d", lda);
  ICAP_REQUIRE(dense_topk_supported(x, w, M, K, N, top_k) && aligned16(records),
               "dense_topk_attention_f32: projection shape not served (M=%d K=%d N=%d top_k=%d)", M,
               K, N, top_k);
  ICAP_REQUIRE(G > 0 && dense_topk_attention_supported(R, L, U, D, H, top_k),
               "dense_topk_attention_f32: attention shape not served (R=%d L=%d U=%d D=%d H=%d)", R,
               L, U, D, H);
  ICAP_REQUIRE(aligned16(pf) && aligned16(features) && aligned16(att_v) && aligned16(q_w) &&
                   aligned16(ctx) && aligned16(state),
               "dense_topk_attention_f32: attention operands must be 16-byte aligned");
  DenseArgs a{x, w, bias, nullptr, M, N, K, lda, N, N, ICAP_ACT_NONE, 0, 0, 0, records, eos_logit, eos};
  if (masks) a.masks = *masks;
  a.tiles_m = ceil_div(M, 128);
  a.tiles_n = ceil_div(N, 128);
  a.band_order = (size_t)M * K * sizeof(float) > (size_t)(5 << 19) ? 1 : 0;
  a.band_width = band_width_for(K, a.tiles_n);
  att_blocks = att_blocks > 0 ? (att_blocks < G ? att_blocks : G) : G;
  if (att_blocks >= 8) att_blocks &= ~7;  // the tile workgroups keep their XCD
  AttArgs t{pf, features, nullptr, att_v, att_bv, nullptr, state, gate_w, gate_b, ctx, nullptr,
            G, R, stride_g, stride_k, L, U, D, H, att_blocks, state, q_w, q_b};
  const dim3 grid(att_blocks + a.tiles_m * a.tiles_n), block(512);
  switch (R) {
    case 1: hipLaunchKernelGGL((dense_topk_attention_kernel<1, 1>), grid, block, 0, s, a, t, att_blocks); break;
    case 2: hipLaunchKernelGGL((dense_topk_attention_kernel<3, 2>), grid, block, 0, s, a, t, att_blocks); break;
    case 3: hipLaunchKernelGGL((dense_topk_attention_kernel<3, 3>), grid, block, 0, s, a, t, att_blocks); break;
    case 4: hipLaunchKernelGGL((dense_topk_attention_kernel<5, 4>), grid, block, 0, s, a, t, att_blocks); break;
    default: hipLaunchKernelGGL((dense_topk_attention_kernel<5, 5>), grid, block, 0, s, a, t, att_blocks); break;
  }
  ICAP_LAUNCH_CHECK("dense_topk_attention_kernel");
  return ICAP_OK;
}

// dense_topk_attention_f32 with the query projection as matrix tiles of the same launch
// (dense_topk_attq_kernel).  q [G*R, U] is scratch the launch fills; q_done counts finished
// query tiles over the launches of a decode (the caller zeroes it once and passes the count
// this launch must reach); err is raised if an attention workgroup's wait runs out.
// rows of a group held in registers per pass over PF (as attention_f32 chooses them: the
// fewest passes, then the fewest idle slots)
static int attq_rows_per_pass(int R) {
  if (R <= 5) return R;
  int rb = 5, best_passes = 1 << 30, best_waste = 1 << 30;
  for (int cand = 5; cand >= 3; --cand) {
    const int passes = ceil_div(R, cand), waste = passes * cand - R;
    if (passes < best_passes || (passes == best_passes && waste < best_waste)) {
      best_passes = passes; best_waste = waste; rb = cand;
    }
  }
  return rb;
}

// R up to ATT_MAX_ROWS rows per group (beams wider than 5 take several passes of 3 to 5 rows)
bool dense_topk_attq_supported(int R, int L, int U, int D, int H, int top_k) {
  if (R < 1 || R > ATT_MAX_ROWS || L < 1 || U < 128 || U > 512 || U % 128 || D < 4 || D % 4 || H < 32 || H % 32)
    return false;
  if (top_k != (R <= 1 ? 1 : (R <= 3 ? 3 : 5))) return false;
  return sizeof(float) * attention_lds_floats(512, R, attq_rows_per_pass(R), L, D) <= 64 * 1024;
}

int dense_topk_attq_tiles(int rows, int U) { return ceil_div(rows, 64) * (U / 128); }

int dense_topk_attq_f32(const float* x, int lda, const float* w, const float* bias, int M, int K,
                        int N, int eos, int top_k, float* records, float* eos_logit,
                        const LogitMasks* masks, const float* pf, const float* features,
                        const float* att_v, float att_bv, const float* state,
                        const float* gate_w, float gate_b, const float* q_w, const float* q_b,
                        float* q, float* ctx, int G, int R, int stride_g, int stride_k, int L, int U,
                        int D, int H, int att_blocks, unsigned* q_done, unsigned q_target, int* err,
                        hipStream_t s, const float* epf, const int* pf_extreme) {
  ICAP_REQUIRE(x && w && records && eos_logit, "dense_topk_attq_f32: null pointer");
  ICAP_REQUIRE((epf == nullptr) == (pf_extreme == nullptr) && (!epf || aligned16(epf)),
               "dense_topk_attq_f32: the exp table comes with its flag, 16-byte aligned");
  ICAP_REQUIRE(pf && features && att_v && state && gate_w && q_w && q && ctx && q_done && err,
               "dense_topk_attq_f32: null attention pointer");
  ICAP_REQUIRE(lda >= K && lda % 4 == 0 && (long long)M * lda < (1LL << 31),
               "dense_topk_attq_f32: bad row stride %d", lda);
  ICAP_REQUIRE(dense_topk_supported(x, w, M, K, N, top_k) && aligned16(records),
               "dense_topk_attq_f32: projection shape not served (M=%d K=%d N=%d top_k=%d)", M, K, N,
               top_k);
  ICAP_REQUIRE(G > 0 && dense_topk_attq_supported(R, L, U, D, H, top_k),
               "dense_topk_attq_f32: attention shape not served (R=%d L=%d U=%d D=%d H=%d)", R, L, U, D, H);
  ICAP_REQUIRE(aligned16(pf) && aligned16(features) && aligned16(att_v) && aligned16(q_w) &&
                   aligned16(ctx) && aligned16(state) && aligned16(q),
               "dense_topk_attq_f32: attention operands must be 16-byte aligned");
  DenseArgs a{x, w, bias, nullptr, M, N, K, lda, N, N, ICAP_ACT_NONE, 0, 0, 0, records, eos_logit, eos};
  if (masks) a.masks = *masks;
  a.tiles_m = ceil_div(M, 128);
  a.tiles_n = ceil_div(N, 128);
  a.band_order = (size_t)M * K * sizeof(float) > (size_t)(5 << 19) ? 1 : 0;
  a.band_width = band_width_for(K, a.tiles_n);
  // q = state . q_w + q_b over all G*R rows, 64 x 128 tiles
  const int rows = G * R;
  DenseArgs aq{state, q_w, q_b, q, rows, U, H, H, U, U, ICAP_ACT_NONE, 0, 0, 0, nullptr, nullptr, -1};
  aq.tiles_m = ceil_div(rows, 64);
  aq.tiles_n = U / 128;
  const int nq = aq.tiles_m * aq.tiles_n;
  // (work items of the attention: one per image and pass of rows)
  const int att_items = G * ceil_div(R, attq_rows_per_pass(R));
  att_blocks = att_blocks > 0 ? (att_blocks < att_items ? att_blocks : att_items) : att_items;
  AttArgs t{pf, features, q, att_v, att_bv, nullptr, state, gate_w, gate_b, ctx, nullptr,
            G, R, stride_g, stride_k, L, U, D, H, att_blocks, nullptr, nullptr, nullptr};
  t.epf = epf;
  t.pf_extreme = pf_extreme;
  // Where the attention workgroups sit in the grid.  In front of every tile they hold one of
  // each compute unit's two slots from the start and the launch ends on a partly filled round
  // of tile pairs; behind one full round of tiles plus half of the slots that round frees,
  // they start beside the second round's tiles and the tiles left for the end are fewer than
  // the compute units (headline shape: 156 us for any value in 704..736 against 176 us in
  // front, 169 us behind every tile).  Tuning knob ICAP_ATTQ_TILES_BEFORE.
  static const char* tb_env = getenv("ICAP_ATTQ_TILES_BEFORE");
  const int slots = 512, nq_pad = (nq + 7) & ~7;
  int tiles_before = tb_env ? atoi(tb_env) : slots + (slots - nq_pad) / 2;
  tiles_before &= ~7;
  if (tiles_before < 0) tiles_before = 0;
  if (tiles_before > a.tiles_m * a.tiles_n) tiles_before = (a.tiles_m * a.tiles_n) & ~7;
  // the tiles of the tail that are cut in two (see the kernel).  Tuning knob ICAP_TAIL_SPLIT.
  static const char* ts_env = getenv("ICAP_TAIL_SPLIT");
  const int n_tiles = a.tiles_m * a.tiles_n;
  int split = ts_env ? atoi(ts_env) : 0;
  split &= ~7;
  if (split < 0) split = 0;
  if (split > n_tiles - tiles_before) split = (n_tiles - tiles_before) & ~7;
  const int split_from = n_tiles - split;
  const dim3 grid(((nq + 7) & ~7) + ((att_blocks + 7) & ~7) + n_tiles + split), block(512);
#define ICAP_ATTQ(TK, RB_) hipLaunchKernelGGL((dense_topk_attq_kernel<TK, RB_>), grid, block, 0, s, a, aq, t, nq, att_blocks, tiles_before, split_from, q_done, q_target, err)
  switch (R <= 5 ? R : 10 + attq_rows_per_pass(R)) {
    case 1: ICAP_ATTQ(1, 1); break;
    case 2: ICAP_ATTQ(3, 2); break;
    case 3: ICAP_ATTQ(3, 3); break;
    case 4: ICAP_ATTQ(5, 4); break;
    case 5: ICAP_ATTQ(5, 5); break;
    case 13: ICAP_ATTQ(5, 3); break;  // wider beams: records of 5, passes of 3 to 5 rows
    case 14: ICAP_ATTQ(5, 4); break;
    default: ICAP_ATTQ(5, 5); break;
  }
#undef ICAP_ATTQ
  ICAP_LAUNCH_CHECK("dense_topk_attq_kernel");
  return ICAP_OK;
}

namespace {

template <int CELL>
int cell_step_f32(const char* who, const float* ctx, const float* emb, const float* table,
                  const int32_t* ids, const float* emb_proj, const float* h, const float* c,
                  const float* kernel, const float* recurrent, const float* bias, float* h_out,
                  float* c_out, int N, int C, int E, int H, int V, hipStream_t s) {
  ICAP_REQUIRE(ctx && h && kernel && recurrent && bias && h_out, "%s: null pointer", who);
  ICAP_REQUIRE(CELL != CELL_LSTM || (c && c_out), "%s: null pointer", who);
  ICAP_REQUIRE((ids && (table || emb_proj)) || (!ids && emb && !emb_proj),
               "%s: pass (table, ids), (emb_proj, ids) or emb", who);
  ICAP_REQUIRE(N > 0 && C > 0 && E > 0 && H > 0, "%s: bad shape", who);
  ICAP_REQUIRE(!ids || V > 0, "%s: V must be positive with ids", who);
  ICAP_REQUIRE(h_out != h && c_out != h, "%s: outputs must not alias h", who);
  // With the pre-multiplied embedding the GEMM runs over [ctx | h] only (K = C + H): the
  // kernel sees E = 0 and adds emb_proj[id] in its epilogue.
  const int e_gemm = emb_proj ? 0 : E;
  LstmArgs a{ctx, emb, table, ids, h, c, kernel, recurrent, bias, emb_proj, h_out, c_out,
             N, C, e_gemm, H, V, ceil_div(N, LSTM_BM), ceil_div(H, LSTM_BJ)};
  const bool vec = (C % 4 == 0) && (E % 4 == 0) && (H % 4 == 0) && aligned16(ctx) &&
                   (emb_proj || aligned16(ids ? table : emb)) && aligned16(h) &&
                   aligned16(kernel) && aligned16(recurrent);
  {
    // wide form when it applies and its 96-row patches use the chip better than 32-row ones
    static const char* form_env = getenv("ICAP_LSTM_FORM");  // tuning knob: "narrow" | "wide"
    const bool can_wide = vec && emb_proj && (C % LW_BK == 0) && (H % LW_BK == 0) &&
                          (long long)N * (C > H ? C : H) < (1LL << 31) &&
                          (long long)(C + H) * 4 * H < (1LL << 31);
    const int wide_grid = ceil_div(N, LW_BM) * (H / 32);
    bool use_wide = can_wide && wide_grid >= 192;
    if (form_env) use_wide = can_wide && form_env[0] == 'w';
    if (use_wide) {
      LstmArgs w = a;
      w.tiles_m = ceil_div(N, LW_BM);
      w.tiles_j = H / 32;
      hipLaunchKernelGGL(lstm_f32_wide_kernel<CELL>, dim3(wide_grid), dim3(LW_NT), 0, s, w);
      ICAP_LAUNCH_CHECK("lstm_f32_wide_kernel");
      return ICAP_OK;
    }
  }
  const int grid = a.tiles_m * a.tiles_j;
  {
    // few rows: the small-batch form (tuning knob ICAP_LSTM_FORM=narrow disables)
    static const char* form_env2 = getenv("ICAP_LSTM_FORM");
    const int ng = (C + H) / 128;  // groups of 8 k per wave
    const bool can_skinny = vec && emb_proj && ids && ((C + H) % 128 == 0) && (H % 8 == 0) &&
                            (ng == 2 || ng == 4 || ng == 6 || ng == 8 || ng == 12) &&
                            (long long)(C + H) * 4 * H < (1LL << 31);
    const int sk_grid = ceil_div(N, 32) * (H / 8);
    if (can_skinny && sk_grid <= 768 && !(form_env2 && form_env2[0] == 'n')) {
      LstmArgs w = a;
      w.tiles_m = ceil_div(N, 32);
      w.tiles_j = H / 8;
      const dim3 g(sk_grid), b(1024);
      if (ng == 2) hipLaunchKernelGGL((lstm_f32_skinny_kernel<CELL, 2>), g, b, 0, s, w);
      else if (ng == 4) hipLaunchKernelGGL((lstm_f32_skinny_kernel<CELL, 4>), g, b, 0, s, w);
      else if (ng == 6) hipLaunchKernelGGL((lstm_f32_skinny_kernel<CELL, 6>), g, b, 0, s, w);
      else if (ng == 8) hipLaunchKernelGGL((lstm_f32_skinny_kernel<CELL, 8>), g, b, 0, s, w);
      else hipLaunchKernelGGL((lstm_f32_skinny_kernel<CELL, 12>), g, b, 0, s, w);
      ICAP_LAUNCH_CHECK("lstm_f32_skinny_kernel");
      return ICAP_OK;
    }
  }
  if (vec)
    hipLaunchKernelGGL((lstm_f32_kernel<true, CELL>), dim3(grid), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((lstm_f32_kernel<false, CELL>), dim3(grid), dim3(256), 0, s, a);
  ICAP_LAUNCH_CHECK("lstm_f32_kernel");
  return ICAP_OK;
}

// Keras GRUCell variables (reset_after=True: kernel [CE,3H], recurrent_kernel [H,3H], bias
// [2,3H], gates z, r, h) laid out as four pre-activation columns per unit so that the LSTM
// gate product serves the GRU: z and r sum the input and the recurrent part, the candidate
// keeps them apart (x_h from the input rows only, rec_h from the recurrent rows only; the
// other rows of those two columns are zero, which adds exact zeros to the sums).
__global__ void gru_pack_kernel(const float* __restrict__ kernel, const float* __restrict__ recurrent,
                                const float* __restrict__ bias, float* __restrict__ kernel4,
                                float* __restrict__ recurrent4, float* __restrict__ bias4, int CE,
                                int H) {
  const long long total = (long long)(CE + H + 1) * 4 * H;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int col = (int)(i % (4 * H));
    const int row = (int)(i / (4 * H));
    const int g = col / H, j = col % H;
    if (row < CE) {
      kernel4[(size_t)row * 4 * H + col] = g < 3 ? kernel[(size_t)row * 3 * H + g * H + j] : 0.f;
    } else if (row < CE + H) {
      const int r = row - CE;
      const int src = g < 2 ? g : 2;  // rec_h is the third recurrent gate
      recurrent4[(size_t)r * 4 * H + col] = g == 2 ? 0.f : recurrent[(size_t)r * 3 * H + src * H + j];
    } else {
      // bias[0] is the input bias, bias[1] the recurrent bias
      const float bi = g < 3 ? bias[g * H + j] : 0.f;
      const float br = g < 2 ? bias[3 * H + g * H + j] : (g == 3 ? bias[3 * H + 2 * H + j] : 0.f);
      bias4[col] = g < 2 ? bi + br : (g == 2 ? bi : br);
    }
  }
}

}  // namespace

int lstm_step_f32(const float* ctx, const float* emb, const float* table, const int32_t* ids,
                  const float* emb_proj, const float* h, const float* c, const float* kernel,
                  const float* recurrent, const float* bias, float* h_out, float* c_out, int N,
                  int C, int E, int H, int V, hipStream_t s) {
  return cell_step_f32<CELL_LSTM>("lstm_step_f32", ctx, emb, table, ids, emb_proj, h, c, kernel,
                                  recurrent, bias, h_out, c_out, N, C, E, H, V, s);
}

int gru_pack_f32(const float* kernel, const float* recurrent, const float* bias, float* kernel4,
                 float* recurrent4, float* bias4, int CE, int H, hipStream_t s) {
  ICAP_REQUIRE(kernel && recurrent && bias && kernel4 && recurrent4 && bias4,
               "gru_pack_f32: null pointer");
  ICAP_REQUIRE(CE > 0 && H > 0, "gru_pack_f32: bad shape");
  const long long total = (long long)(CE + H + 1) * 4 * H;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(gru_pack_kernel, dim3(grid), dim3(256), 0, s, kernel, recurrent, bias, kernel4,
                     recurrent4, bias4, CE, H);
  ICAP_LAUNCH_CHECK("gru_pack_kernel");
  return ICAP_OK;
}

int gru_step_f32(const float* ctx, const float* emb, const float* table, const int32_t* ids,
                 const float* emb_proj, const float* h, const float* kernel4,
                 const float* recurrent4, const float* bias4, float* h_out, int N, int C, int E,
                 int H, int V, hipStream_t s) {
  return cell_step_f32<CELL_GRU>("gru_step_f32", ctx, emb, table, ids, emb_proj, h, nullptr, kernel4,
                                 recurrent4, bias4, h_out, nullptr, N, C, E, H, V, s);
}

}  // namespace icap

#ifdef ICAP_TIMELINE
__device__ long long icap_tl_f32[4096][8];
extern "C" int icap_debug_timeline_f32(long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(icap_tl_f32), sizeof(long long) * 8 * (n < 4096 ? n : 4096));
}
#endif
