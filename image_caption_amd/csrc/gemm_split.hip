// fp32-accurate contractions on the bf16 matrix pipe ("split" precision mode).
//
// gfx950 multiplies bf16 operands 16 times faster than f32 ones (v_mfma_f32_32x32x16_bf16
// against v_mfma_f32_32x32x2_f32).  An fp32 number is EXACTLY the sum of three bf16 numbers
// (8 + 8 + 8 mantissa bits: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)), so
//   a * b = (a_hi + a_mid + a_lo) * (b_hi + b_mid + b_lo)
// and every partial product of two bf16 values is exact in fp32.  This kernel accumulates,
// in fp32, the six partial products of weight 2^0, 2^-8 and 2^-16
//   hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid
// and drops mid*lo, lo*mid (2^-24) and lo*lo (2^-32): at most 2^-23 relative to |a*b| per
// product, the size of one fp32 rounding.  Six bf16 MFMAs replace eight f32 ones per 16 k:
// 6/16 of the matrix-pipe time of the exact-f32 kernels for results that differ from them
// by accumulation-order-sized amounts.
//
// Operands.  Activations stay fp32 in memory and are split while they are staged into LDS.
// Weights are split once per engine into three planes packed like the bf16 mode's weights:
// [3][N_pad][K] bf16, k contiguous, N padded to 128 (LSTM: gates interleaved per 32 units).
// Tile BM x 128 x 32 per stage, four waves in a 2 x 2 grid, one LDS buffer (61 KB with all
// planes, two workgroups per CU cover each other's two barriers per stage), next stage's
// global loads in flight during the MFMAs.  LDS rows are 80 bytes (64 data + 16 pad):
// conflict-free 16-byte operand reads.
#include "icap_common.h"
#include "topk_epilogue.h"

namespace icap {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;
constexpr int BN = 128;
constexpr int ROW_B = BK * 2 + 16;  // bytes per LDS tile row
constexpr int ZLD = BN + 4;         // floats per row of the gate exchange buffer
constexpr int NP = 3;               // planes

struct SplitArgs {
  const float* a0;  // A = [a0 | a1] along k; a1 may be null (K0 == K)
  const float* a1;
  int K0, lda0, lda1;
  const uint16_t* w;  // packed [3][N_pad][K]
  size_t plane;       // elements per plane = N_pad * K
  const float* bias;
  float* y;
  int M, N, K, act;
  int tiles_m, tiles_n;
  // LSTM epilogue
  const float* c_in;
  float* h_out;
  float* c_out;
  const float* emb_proj;
  const int32_t* ids;
  int H, V;
  // top-k epilogue
  float* rec;
  float* eos_x;
  int eos;
  LogitMasks masks;  // additive masks applied before the records are built (default: none)
};

// x = hi + mid + lo exactly (round-to-nearest-even conversions; the differences are exact)
__device__ __forceinline__ void split3(float x, __bf16& hi, __bf16& mid, __bf16& lo) {
  hi = (__bf16)x;
  const float r1 = x - (float)hi;
  mid = (__bf16)r1;
  lo = (__bf16)(r1 - (float)mid);
}

template <int BM, bool LSTM, int TOPK>
__global__ __launch_bounds__(256, 2) void gemm_split_kernel(SplitArgs p) {
  constexpr int NT = 256;
  constexpr int WTM = BM / 2;   // wave tile rows
  constexpr int TM = WTM / 32;  // accumulators along m
  constexpr int TN = 2;         // wave tile = WTM x 64
  constexpr int A_LD = BM * (BK / 4) / NT;       // float4 loads per thread per stage
  constexpr int B_LD = NP * BN * (BK / 8) / NT;  // 16-byte loads per thread per stage
  constexpr int A_PLANE = BM * ROW_B, B_PLANE = BN * ROW_B;
  constexpr int STAGE = NP * (A_PLANE + B_PLANE);
  constexpr int T_LD = 68;
  constexpr int Z_BYTES = LSTM ? BM * ZLD * 4 : (TOPK > 0 ? 4 * WTM * T_LD * 4 : 0);
  static_assert(TOPK == 0 || (!LSTM && BM == 128), "the top-k epilogue is built for 128-row tiles");
  constexpr int SMEM = STAGE > Z_BYTES ? STAGE : Z_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[SMEM];
  char* const as_base = smem;
  char* const bs_base = smem + NP * A_PLANE;

  int tile_m, tile_n;
  tile_of_block_band(blockIdx.x, p.tiles_m, p.tiles_n, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;

  // ---- global -> register staging ----
  const int a_c4 = tid & 7, a_r = tid >> 3;  // 8 float4 per row, 32 rows per pass
  int a_row[A_LD];
#pragma unroll
  for (int i = 0; i < A_LD; ++i) a_row[i] = min(m0 + a_r + 32 * i, p.M - 1);  // clamped, stores masked
  // B: chunk f = tid + j*256 of the stage's 3 * 128 * 4 sixteen-byte chunks
  size_t b_off[B_LD];
  int b_lds[B_LD];
#pragma unroll
  for (int j = 0; j < B_LD; ++j) {
    const int f = tid + j * NT;
    const int pl = f / (BN * 4), r = (f / 4) % BN, ch = f % 4;
    b_off[j] = (size_t)pl * p.plane + (size_t)(n0 + r) * p.K + ch * 8;
    b_lds[j] = pl * B_PLANE + r * ROW_B + ch * 16;
  }
  float4 ra[A_LD];
  uint4 rb[B_LD];
#define ICAP_SP_ADDR(KT)                                                      \
  const int k0_ = (KT) * BK;                                                  \
  const bool seg0_ = k0_ < p.K0; /* uniform over the workgroup */             \
  const float* base_ = seg0_ ? p.a0 : p.a1;                                   \
  const int ld_ = seg0_ ? p.lda0 : p.lda1;                                    \
  const int kk_ = (seg0_ ? k0_ : k0_ - p.K0) + a_c4 * 4;
#define ICAP_SP_LOAD(KT)                                                                      \
  {                                                                                           \
    ICAP_SP_ADDR(KT)                                                                          \
    _Pragma("unroll") for (int i = 0; i < A_LD; ++i) ra[i] =                                  \
        *reinterpret_cast<const float4*>(base_ + (size_t)a_row[i] * ld_ + kk_);               \
    _Pragma("unroll") for (int j = 0; j < B_LD; ++j) rb[j] =                                  \
        *reinterpret_cast<const uint4*>(p.w + b_off[j] + k0_);                                \
  }
#define ICAP_SP_STORE()                                                                       \
  {                                                                                           \
    _Pragma("unroll") for (int i = 0; i < A_LD; ++i) {                                        \
      bf16x4 h_, m_, l_;                                                                      \
      const float e_[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};                               \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                         \
        __bf16 hq_, mq_, lq_;                                                                 \
        split3(e_[q], hq_, mq_, lq_);                                                         \
        h_[q] = hq_; m_[q] = mq_; l_[q] = lq_;                                                \
      }                                                                                       \
      char* d_ = as_base + (a_r + 32 * i) * ROW_B + a_c4 * 8;                                 \
      *reinterpret_cast<bf16x4*>(d_) = h_;                                                    \
      *reinterpret_cast<bf16x4*>(d_ + A_PLANE) = m_;                                          \
      *reinterpret_cast<bf16x4*>(d_ + 2 * A_PLANE) = l_;                                      \
    }                                                                                         \
    _Pragma("unroll") for (int j = 0; j < B_LD; ++j)                                          \
        *reinterpret_cast<uint4*>(bs_base + b_lds[j]) = rb[j];                                \
  }

  // two accumulators per tile: the hi*hi products, and the five cross products, which are
  // 2^-8 and less of the first and so round 2^-8 finer when they are summed among themselves
  f32x16 acc[TM][TN], acs[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; acs[i][j][r] = 0.f; }

  const int nk = p.K / BK;
  ICAP_SP_LOAD(0)
  ICAP_SP_STORE()
  const char* as = as_base + (wm * WTM + l31) * ROW_B + half * 16;
  const char* bs = bs_base + (wn * 64 + l31) * ROW_B + half * 16;
  for (int kt = 0; kt < nk; ++kt) {
    // The next stage's loads are issued ABOVE the barrier that opens this stage's multiplies:
    // memory operations are not moved across a barrier, so the prefetch stays here and is in
    // flight during the MFMAs (placed below the barrier, the scheduler sinks the loads to
    // just above their use).  The last step repeats its own loads: no branch around them.
    ICAP_SP_LOAD(min(kt + 1, nk - 1))
    __syncthreads();  // stage kt is in LDS
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 a[NP][TM], b[NP][TN];
#pragma unroll
      for (int q = 0; q < NP; ++q) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          a[q][i] = *reinterpret_cast<const bf16x8*>(as + q * A_PLANE + i * 32 * ROW_B + ks * 32);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          b[q][j] = *reinterpret_cast<const bf16x8*>(bs + q * B_PLANE + j * 32 * ROW_B + ks * 32);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acs[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1][j], acs[i][j], 0, 0, 0);
          acs[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0][j], acs[i][j], 0, 0, 0);
          acs[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2][j], acs[i][j], 0, 0, 0);
          acs[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], acs[i][j], 0, 0, 0);
          acs[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], acs[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();  // every wave is done reading the buffer
    // The staging registers pass through an empty asm here, so that their first use - and
    // with it the wait for the loads - stays below the multiplies (the splitting arithmetic
    // is register-only and would otherwise be hoisted into the MFMA phase, wait included).
#pragma unroll
    for (int i = 0; i < A_LD; ++i) asm volatile("" : "+v"(ra[i].x), "+v"(ra[i].y), "+v"(ra[i].z), "+v"(ra[i].w));
#pragma unroll
    for (int j = 0; j < B_LD; ++j) asm volatile("" : "+v"(rb[j].x), "+v"(rb[j].y), "+v"(rb[j].z), "+v"(rb[j].w));
    ICAP_SP_STORE()
  }
  __syncthreads();  // the epilogues reuse the LDS
#undef ICAP_SP_ADDR
#undef ICAP_SP_LOAD
#undef ICAP_SP_STORE
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] += acs[i][j][r];

  if constexpr (TOPK > 0) {
    // the logits never reach memory: patch to LDS, lane r folds row r into its record
    constexpr int RL = 2 + 2 * TOPK;
    float* patch = reinterpret_cast<float*>(smem) + wave * WTM * T_LD;
    const int colbase = n0 + wn * 64;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = colbase + j * 32 + l31;
      const bool n_ok = n < p.N;
      const float bv = (p.bias && n_ok) ? p.bias[n] : 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int lr = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          patch[lr * T_LD + j * 32 + l31] = n_ok ? acc[i][j][r] + bv : -INFINITY;
        }
    }
    __syncthreads();
    float x[64];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(patch + lane * T_LD + 4 * q);
      x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
    }
    const int row = m0 + wm * WTM + lane;
    const bool masked = p.masks.any();
    if (masked) apply_logit_masks<64>(x, colbase, row, p.M, p.masks);
    TopkRow<TOPK> rec;
    if constexpr (TOPK > 1) rec = topk_scan_keyed<TOPK, 64>(x, patch + lane * T_LD, colbase, p.eos, masked);
    else rec = topk_scan<TOPK, 64>(x, colbase, p.eos);
    if (row < p.M) {
      if (rec.has_eos) p.eos_x[row] = rec.eos_v;
      topk_store<TOPK>(p.rec + ((size_t)row * (2 * p.tiles_n) + 2 * tile_n + wn) * RL, rec);
    }
  } else if (!LSTM) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * 64 + j * 32 + l31;
      const bool n_ok = n < p.N;
      const float bv = (p.bias && n_ok) ? p.bias[n] : 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          float v = acc[i][j][r] + bv;
          if (p.act == ICAP_ACT_RELU) v = fmaxf(v, 0.f);
          if (n_ok && m < p.M) p.y[(size_t)m * p.N + n] = v;
        }
    }
  } else {
    // gates meet in LDS: column g*32 + u of the tile is gate g of unit n0/4 + u
    float* zs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          zs[row * ZLD + wn * 64 + j * 32 + l31] = acc[i][j][r];
        }
    __syncthreads();
    const int H = p.H;
    const int u = tid & 31;
    const int unit = (n0 >> 2) + u;
    float bz[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bz[g] = p.bias ? p.bias[g * H + unit] : 0.f;
    for (int row = tid >> 5; row < BM; row += NT / 32) {
      const int m = m0 + row;
      if (m >= p.M) break;
      float z[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) z[g] = zs[row * ZLD + g * 32 + u];
      if (p.emb_proj) {
        int id = p.ids[m];
        id = min(max(id, 0), p.V - 1);
        const float* ep = p.emb_proj + (size_t)id * 4 * H + unit;
#pragma unroll
        for (int g = 0; g < 4; ++g) z[g] += ep[g * H];
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) z[g] += bz[g];
      const size_t o = (size_t)m * H + unit;
      const float cn = fast_sigmoid(z[1]) * p.c_in[o] + fast_sigmoid(z[0]) * fast_tanh(z[2]);
      p.c_out[o] = cn;
      p.h_out[o] = fast_sigmoid(z[3]) * fast_tanh(cn);
    }
  }
}

// w[K_src][N] fp32 -> three bf16 planes packed[pl][n'][k_off + k]
__global__ __launch_bounds__(256) void pack_split_kernel(const float* __restrict__ w,
                                                         uint16_t* __restrict__ packed, int K_src,
                                                         int N, int k_off, int K_total,
                                                         int gate_units, size_t plane) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)K_src * N) return;
  const int k = (int)(idx / N), n = (int)(idx - (long long)k * N);
  int np = n;
  if (gate_units > 0) {
    const int g = n / gate_units, u = n - g * gate_units;
    np = (u >> 5) * 128 + g * 32 + (u & 31);
  }
  __bf16 hi, mid, lo;
  split3(w[idx], hi, mid, lo);
  const size_t o = (size_t)np * K_total + k_off + k;
  packed[o] = __builtin_bit_cast(uint16_t, hi);
  packed[plane + o] = __builtin_bit_cast(uint16_t, mid);
  packed[2 * plane + o] = __builtin_bit_cast(uint16_t, lo);
}

size_t plane_elems(int K, int N) { return (size_t)ceil_div(N, BN) * BN * K; }

template <int BM, bool LSTM, int TOPK>
int launch(SplitArgs& a, hipStream_t s, const char* name) {
  a.tiles_m = ceil_div(a.M, BM);
  a.tiles_n = ceil_div(a.N, BN);
  a.plane = plane_elems(a.K, a.N);
  hipLaunchKernelGGL((gemm_split_kernel<BM, LSTM, TOPK>), dim3(a.tiles_m * a.tiles_n), dim3(256), 0,
                     s, a);
  ICAP_LAUNCH_CHECK(name);
  return ICAP_OK;
}

}  // namespace

size_t packed_split_bytes(int K, int N) {
  if (K <= 0 || N <= 0) return 0;
  return NP * plane_elems(K, N) * sizeof(uint16_t);
}

int pack_weight_split(const float* w, void* packed, int K_src, int N, int k_offset, int K_total,
                      int gate_units, hipStream_t s) {
  ICAP_REQUIRE(w && packed, "pack_weight_split: null pointer");
  ICAP_REQUIRE(K_src > 0 && N > 0 && k_offset >= 0 && k_offset + K_src <= K_total,
               "pack_weight_split: rows [%d,%d) outside [0,%d)", k_offset, k_offset + K_src, K_total);
  ICAP_REQUIRE(aligned16(packed), "pack_weight_split: packed buffer must be 16-byte aligned");
  if (gate_units > 0)
    ICAP_REQUIRE(N == 4 * gate_units && gate_units % 32 == 0,
                 "pack_weight_split: gate layout needs N == 4*units and units %% 32 == 0");
  const size_t plane = plane_elems(K_total, N);
  const int n_pad = ceil_div(N, BN) * BN;
  if (n_pad != N && k_offset == 0)  // zero rows for the padded columns of every plane
    for (int pl = 0; pl < NP; ++pl)
      ICAP_HIP(hipMemsetAsync(static_cast<uint16_t*>(packed) + pl * plane + (size_t)N * K_total, 0,
                              (size_t)(n_pad - N) * K_total * sizeof(uint16_t), s));
  const long long total = (long long)K_src * N;
  hipLaunchKernelGGL(pack_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w,
                     static_cast<uint16_t*>(packed), K_src, N, k_offset, K_total, gate_units, plane);
  ICAP_LAUNCH_CHECK("pack_split_kernel");
  return ICAP_OK;
}

static void fill_dense(SplitArgs& a, const float* x, const void* packed_w, const float* bias,
                       int M, int K, int N) {
  a.a0 = x; a.a1 = nullptr; a.K0 = K; a.lda0 = K; a.lda1 = 0;
  a.w = static_cast<const uint16_t*>(packed_w);
  a.bias = bias; a.M = M; a.N = N; a.K = K; a.act = ICAP_ACT_NONE; a.eos = -1;
}

int dense_split(const float* x, const void* packed_w, const float* bias, float* y, int M, int K,
                int N, int act, hipStream_t s) {
  ICAP_REQUIRE(x && packed_w && y, "dense_split: null pointer");
  ICAP_REQUIRE(M > 0 && K > 0 && N > 0, "dense_split: bad shape M=%d K=%d N=%d", M, K, N);
  ICAP_REQUIRE(K % BK == 0, "dense_split: K=%d must be a multiple of %d", K, BK);
  ICAP_REQUIRE(aligned16(x) && aligned16(packed_w), "dense_split: operands must be 16-byte aligned");
  ICAP_REQUIRE(act == ICAP_ACT_NONE || act == ICAP_ACT_RELU, "dense_split: unknown activation %d", act);
  SplitArgs a = {};
  fill_dense(a, x, packed_w, bias, M, K, N);
  a.y = y; a.act = act;
  if ((long long)ceil_div(M, 128) * ceil_div(N, BN) >= 256)
    return launch<128, false, 0>(a, s, "gemm_split_kernel<128>");
  return launch<64, false, 0>(a, s, "gemm_split_kernel<64>");
}

bool dense_topk_split_supported(const float* x, int M, int K, int N, int top_k) {
  return (top_k == 1 || top_k == 3 || top_k == 5) && M > 0 && K > 0 && N > 0 && K % BK == 0 &&
         aligned16(x);
}

int dense_topk_split(const float* x, const void* packed_w, const float* bias, int M, int K, int N,
                     int eos, int top_k, float* records, float* eos_logit, const LogitMasks* masks,
                     hipStream_t s) {
  ICAP_REQUIRE(x && packed_w && records && eos_logit, "dense_topk_split: null pointer");
  ICAP_REQUIRE(dense_topk_split_supported(x, M, K, N, top_k),
               "dense_topk_split: needs top_k in {1,3,5}, K %% 32 == 0 and a 16-byte aligned x "
               "(M=%d K=%d N=%d top_k=%d)", M, K, N, top_k);
  ICAP_REQUIRE(aligned16(packed_w) && aligned16(records), "dense_topk_split: unaligned buffer");
  SplitArgs a = {};
  fill_dense(a, x, packed_w, bias, M, K, N);
  a.rec = records; a.eos_x = eos_logit; a.eos = eos;
  if (masks) a.masks = *masks;
  if (top_k == 1) return launch<128, false, 1>(a, s, "gemm_split_kernel<topk>");
  if (top_k == 3) return launch<128, false, 3>(a, s, "gemm_split_kernel<topk>");
  return launch<128, false, 5>(a, s, "gemm_split_kernel<topk>");
}

int lstm_step_split(const float* ctx, const float* h, const float* c, const void* packed_w,
                    const float* bias, const float* emb_proj, const int32_t* ids, float* h_out,
                    float* c_out, int N, int C, int H, int V, hipStream_t s) {
  ICAP_REQUIRE(ctx && h && c && packed_w && h_out && c_out, "lstm_step_split: null pointer");
  ICAP_REQUIRE(N > 0 && C > 0 && H > 0, "lstm_step_split: bad shape");
  ICAP_REQUIRE(C % BK == 0 && H % BK == 0, "lstm_step_split: C=%d and H=%d must be multiples of %d",
               C, H, BK);
  ICAP_REQUIRE((emb_proj == nullptr) == (ids == nullptr),
               "lstm_step_split: emb_proj and ids come together");
  ICAP_REQUIRE(!emb_proj || V > 0, "lstm_step_split: vocabulary size missing");
  ICAP_REQUIRE(h_out != h, "lstm_step_split: h_out must not alias h");
  ICAP_REQUIRE(aligned16(ctx) && aligned16(h) && aligned16(packed_w),
               "lstm_step_split: operands must be 16-byte aligned");
  SplitArgs a = {};
  a.a0 = ctx; a.a1 = h; a.K0 = C; a.lda0 = C; a.lda1 = H;
  a.w = static_cast<const uint16_t*>(packed_w);
  a.bias = bias; a.y = nullptr; a.M = N; a.N = 4 * H; a.K = C + H; a.act = ICAP_ACT_NONE;
  a.c_in = c; a.h_out = h_out; a.c_out = c_out; a.emb_proj = emb_proj; a.ids = ids; a.H = H; a.V = V;
  a.eos = -1;
  return launch<64, true, 0>(a, s, "gemm_split_kernel<lstm>");
}

}  // namespace icap
