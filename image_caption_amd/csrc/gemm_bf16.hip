// bf16-operand contractions for the reduced-precision mode: y = act(x . W + b) and the
// LSTM gate product with its cell update, on v_mfma_f32_32x32x16_bf16 (fp32 accumulate).
//
// Activations stay fp32 in memory and are rounded to bf16 (round-to-nearest-even) while
// they are staged into LDS.  Weights are packed once per engine: transposed to
// [N_pad][K] bf16 with k contiguous (both MFMA operands then read 8 consecutive k with
// one ds_read_b128), N padded to a multiple of 128 with zero rows.  The LSTM packing also
// interleaves the gates in groups of 32 units, so a 128-column tile holds i,f,g,o of the
// same 32 units and the cell update runs in the epilogue of the same kernel.
//
// Tile: BM x 128 x 64 per stage, waves in a 2 x 2 grid, LDS double-buffered, the global
// loads of stage k+1 in flight during the MFMAs of stage k, one barrier per stage.  LDS
// rows are 128 bytes with their eight 16-byte chunks XOR-swizzled by (row >> 1) & 7: the
// lanes of a ds_read_b128 group (sixteen rows, one chunk index) then touch every bank once,
// and without padding a 64-row stage pair is 48 KB (three workgroups per compute unit) and
// a 128-row one 64 KB.
#include <stdlib.h>

#include "icap_common.h"
#include "topk_epilogue.h"
#include "attention_body.h"

#ifdef ICAP_TIMELINE
// Experiment builds only (-DICAP_TIMELINE): every workgroup of gemm_bf16_kernel leaves the
// 100 MHz clock at four points of its life, read back through icap_debug_timeline.
__device__ long long g_timeline[8192][4];
// ICAP_TIMELINE = 2 stamps only the launches with the top-k record epilogue (inside a decode)
#define ICAP_STAMP(i) if ((ICAP_TIMELINE != 2 || TOPK > 0) && threadIdx.x == 0 && bid < 8192) g_timeline[bid][i] = wall_clock64();
extern "C" int icap_debug_timeline(long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_timeline), sizeof(long long) * 4 * (n < 8192 ? n : 8192));
}
#else
#define ICAP_STAMP(i)
#endif

namespace icap {
namespace {

// arrivals per compute unit (index: XCC id, shader engine / array, CU id), counted up for ever:
// only the parity of the count is used
__device__ unsigned g_cu_arrivals[4096];

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 64;
constexpr int BN = 128;
constexpr int ROW_B = BK * 2;  // bytes per LDS tile row: 8 chunks of 16 bytes, chunk c of row r at c ^ ((r >> 1) & 7)
constexpr int ZLD = BN + 4;         // floats per row of the gate exchange buffer

struct Bf16Args {
  const float* a0;  // A = [a0 | a1] along k; a1 may be null (K0 == K)
  const float* a1;
  int K0, lda0, lda1;
  const uint16_t* w;  // packed [N_pad][K]
  const float* bias;
  float* y;
  int M, N, K, act;
  int tiles_m, tiles_n;
  // LSTM epilogue
  const float* c_in;
  float* h_out;
  float* c_out;
  const float* emb_proj;  // [V,4H] token-only gate pre-activations, may be null
  const int32_t* ids;
  int H, V;
  int gru;  // cell update: 0 LSTM (i, f, g, o), 1 GRU on four-column operands (z, r, x_h, rec_h)
  // top-k epilogue (dense_topk_bf16): per (row, 64 columns) records instead of y
  float* rec;
  float* eos_x;
  int eos;
  LogitMasks masks;  // additive masks applied before the records are built (default: none)
  // A already in bf16 ([M][K], row stride lda16 elements; A16 kernels) / a bf16 copy of y
  const uint16_t* a16 = nullptr;
  int lda16 = 0;
  uint16_t* y16 = nullptr;  // [M][N], written beside y when not null
  // Stagger of the first round of workgroups (top-k projection): of the workgroups that start
  // together on a compute unit, every second one (by order of arrival, counted per unit) first
  // waits stagger_ticks of the 100 MHz clock.  0: off.
  int stagger_ticks = 0;
  int stagger_wgs = 0;  // workgroups of the first round (blockIdx below this take part)
  int y16_blocked = 0;      // y16 in the operand-block layout of vocab_bf16.hip instead of row-major
};

// A16: the A operand is already bf16 in memory (written by the producing kernel's epilogue,
// same round-to-nearest-even as the staging conversion, so the products are the same): half
// the operand traffic, and at the vocab projection's shape the 2.6 MB activation panel then
// stays in an XCD's L2 beside its band of weight panels.
// LDS bytes of a tile workgroup: two operand stages, reused by the epilogues
template <int BM, bool LSTM, int TOPK>
constexpr int gemm_bf16_smem_bytes() {
  constexpr int stages = 2 * (BM * ROW_B + BN * ROW_B);
  constexpr int z = LSTM ? BM * ZLD * 4 : (TOPK > 0 ? 4 * (BM / 2) * 68 * 4 : 0);
  return stages > z ? stages : z;
}

// One tile by a 256-thread workgroup; bid = position in the tile grid, smem = the workgroup's
// LDS (gemm_bf16_smem_bytes).  COH: the plain epilogue stores y with device scope, for a
// reader inside the same launch (the query tiles of vocab_bf16_attq_kernel).
// PAIR: two K stages per trip of the loop, the loads of both in flight over the multiplies of
// two stages (the two LDS buffers then hold the pair being multiplied, so a second barrier per
// trip separates their last read from the stores of the next pair; K / 64 must be even).  An
// experiment that answered a question: a trip of the plain loop takes about a microsecond,
// as long as a load from a busy L2 - but halving the number of exposed latencies does not
// shorten the loop, so the loop is bound by operand throughput, not by latency.
template <int BM, bool LSTM, int TOPK = 0, bool A16 = false, bool COH = false, bool PAIR = false>
__device__ __forceinline__ void gemm_bf16_body(const Bf16Args& p, const int bid, char* const smem) {
  constexpr int NT = 256;
  constexpr int WTM = BM / 2;    // wave tile rows
  constexpr int TM = WTM / 32;   // accumulators along m
  constexpr int TN = 2;          // wave tile = WTM x 64
  constexpr int A_LD = A16 ? BM * (BK / 8) / NT : BM * (BK / 4) / NT;  // 16-byte loads per thread per stage
  constexpr int B_LD = BN * (BK / 8) / NT;  // 16-byte loads per thread per stage
  static_assert(!A16 || !LSTM, "the bf16 A operand is built for the plain products");
  constexpr int A_BYTES = BM * ROW_B, B_BYTES = BN * ROW_B;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int T_LD = 68;  // floats per row of a wave's transposition patch (top-k epilogue)
  static_assert(TOPK == 0 || (!LSTM && BM == 128), "the top-k epilogue is built for 128-row tiles");

  ICAP_STAMP(0)
  if constexpr (TOPK > 0) {
    // The K loop of these tiles waits on operand delivery and their record epilogue on the
    // vector unit: two workgroups that share a compute unit and start together go through
    // both phases in lockstep and use one resource at a time.  Delaying one of them once puts
    // the K loop of one beside the epilogue of the other; the offset carries over to the
    // later rounds, whose workgroups start when their predecessors end.
    if (p.stagger_ticks > 0 && bid < p.stagger_wgs) {
      // (the word lives in the staging buffer: a second LDS object would make the compiler
      // drain the memory queue in front of every K stage's LDS reads)
      volatile unsigned* s_arrival = reinterpret_cast<volatile unsigned*>(smem);
      if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((7 << 11) | (8 << 6) | 4);   // HW_ID[15:8]: CU, SH, SE
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20); // XCC_ID[3:0]
        *s_arrival = atomicAdd(&g_cu_arrivals[((xcc & 15u) << 8) | (hw & 255u)], 1u);
      }
      __syncthreads();
      const bool late = (*s_arrival & 1u) != 0;
      __syncthreads();  // (the staging stores below may overwrite the word)
      if (late) {
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < p.stagger_ticks) __builtin_amdgcn_s_sleep(8);
      }
    }
  }
  int tile_m, tile_n;
  tile_of_block_band(bid, p.tiles_m, p.tiles_n, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;

  // ---- global -> register staging ----
  const int a_c4 = tid & 15, a_r = tid >> 4;  // 16 rows per pass
  const int b_ch = tid & 7, b_r = tid >> 3;   // 32 rows per pass
  // swizzled byte offsets inside an LDS row (a pass advances the row by 16 or 32: same swizzle)
  const int b_sw = (b_ch ^ ((b_r >> 1) & 7)) << 4;
  const int a_sw = (((a_c4 >> 1) ^ ((a_r >> 1) & 7)) << 4) | ((a_c4 & 1) << 3);
  int rd_sw[BK / 16];  // chunk 2*ks + half of this lane's row (wave tiles start on multiples of 16 rows)
#pragma unroll
  for (int ks = 0; ks < BK / 16; ++ks) rd_sw[ks] = ((2 * ks + half) ^ ((l31 >> 1) & 7)) << 4;
  int a_row[A_LD];
#pragma unroll
  for (int i = 0; i < A_LD; ++i)  // clamped, stores are masked
    a_row[i] = A16 ? min(m0 + b_r + 32 * i, p.M - 1) : min(m0 + a_r + 16 * i, p.M - 1);
  float4 ra[A_LD];  // (A16: the same 16 bytes hold 8 bf16)
  uint4 rb[B_LD];
  float4 ra2[PAIR ? A_LD : 1];  // PAIR: the second stage of the pair in flight
  uint4 rb2[PAIR ? B_LD : 1];
  // (macros, not lambdas: arrays captured by a lambda are not split into registers)
#define ICAP_BF16_LOAD_INTO(RA, RB, KT)                                                            \
  {                                                                                                \
    const int k0_ = (KT) * BK;                                                                     \
    const bool seg0_ = k0_ < p.K0; /* uniform over the workgroup */                                \
    const float* base_ = seg0_ ? p.a0 : p.a1;                                                      \
    const int ld_ = seg0_ ? p.lda0 : p.lda1;                                                       \
    const int kk_ = (seg0_ ? k0_ : k0_ - p.K0) + a_c4 * 4;                                         \
    _Pragma("unroll") for (int i = 0; i < A_LD; ++i) RA[i] = A16                                   \
        ? *reinterpret_cast<const float4*>(p.a16 + (size_t)a_row[i] * p.lda16 + k0_ + b_ch * 8)    \
        : *reinterpret_cast<const float4*>(base_ + (size_t)a_row[i] * ld_ + kk_);                  \
    _Pragma("unroll") for (int j = 0; j < B_LD; ++j) RB[j] = *reinterpret_cast<const uint4*>(      \
        p.w + (size_t)(n0 + b_r + 32 * j) * p.K + k0_ + b_ch * 8);                                 \
  }
#define ICAP_BF16_STORE_FROM(RA, RB, BUF)                                                          \
  {                                                                                                \
    char* as_ = smem + (BUF) * STAGE;                                                              \
    char* bs_ = as_ + A_BYTES;                                                                     \
    _Pragma("unroll") for (int i = 0; i < A_LD; ++i) {                                             \
      if (A16) {                                                                                   \
        *reinterpret_cast<float4*>(as_ + (b_r + 32 * i) * ROW_B + b_sw) = RA[i];                   \
      } else {                                                                                     \
        bf16x4 v_;                                                                                 \
        v_[0] = (__bf16)RA[i].x; v_[1] = (__bf16)RA[i].y;                                          \
        v_[2] = (__bf16)RA[i].z; v_[3] = (__bf16)RA[i].w;                                          \
        *reinterpret_cast<bf16x4*>(as_ + (a_r + 16 * i) * ROW_B + a_sw) = v_;                      \
      }                                                                                            \
    }                                                                                              \
    _Pragma("unroll") for (int j = 0; j < B_LD; ++j)                                               \
        *reinterpret_cast<uint4*>(bs_ + (b_r + 32 * j) * ROW_B + b_sw) = RB[j];                    \
  }
#define ICAP_BF16_LOAD(KT) ICAP_BF16_LOAD_INTO(ra, rb, KT)
#define ICAP_BF16_STORE(BUF) ICAP_BF16_STORE_FROM(ra, rb, BUF)
#define ICAP_BF16_PIN(RA, RB)                                                                      \
  _Pragma("unroll") for (int i = 0; i < A_LD; ++i)                                                 \
      asm volatile("" : "+v"(RA[i].x), "+v"(RA[i].y), "+v"(RA[i].z), "+v"(RA[i].w));               \
  _Pragma("unroll") for (int j = 0; j < B_LD; ++j)                                                 \
      asm volatile("" : "+v"(RB[j].x), "+v"(RB[j].y), "+v"(RB[j].z), "+v"(RB[j].w));

  // LSTM / GRU epilogue operands of this thread's (row, unit) pairs - the token-only gate
  // pre-activations (a gather by id) and the previous state - are fetched here, ahead of the
  // K loop: inside the epilogue's row loop every iteration would wait out its own id ->
  // row -> value chain behind the previous iteration's stores.
  // (64-row tiles; 128-row tiles hold too many pairs per thread to keep them across the loop
  // and fetch them in one batch right behind it instead)
  constexpr int NQ = LSTM ? BM / (NT / 32) : 1;
  constexpr bool HOIST = NQ <= 8;
  float ep_pre[NQ][4], st_pre[NQ];
  int id_pre[NQ];
#define ICAP_BF16_CELL_IDS                                                                         \
  _Pragma("unroll") for (int q = 0; q < NQ; ++q) {                                                 \
    const int m_ = min(m0 + (tid >> 5) + q * (NT / 32), p.M - 1); /* clamped rows are never stored */ \
    id_pre[q] = p.emb_proj ? min(max(p.ids[m_], 0), p.V - 1) : 0;                                  \
  }
#define ICAP_BF16_CELL_OPERANDS                                                                    \
  {                                                                                                \
    const int unit_ = (n0 >> 2) + (tid & 31);                                                      \
    _Pragma("unroll") for (int q = 0; q < NQ; ++q) {                                               \
      const int m_ = min(m0 + (tid >> 5) + q * (NT / 32), p.M - 1);                                \
      const size_t o_ = (size_t)m_ * p.H + unit_;                                                  \
      st_pre[q] = p.gru ? p.a1[o_] : p.c_in[o_]; /* a1 is the previous state h [N,H] */            \
      if (p.emb_proj) { /* uniform */                                                              \
        const float* ep_ = p.emb_proj + (size_t)id_pre[q] * 4 * p.H + unit_;                       \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) ep_pre[q][g] = ep_[g * p.H];                 \
      } else {                                                                                     \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) ep_pre[q][g] = 0.f;                          \
      }                                                                                            \
    }                                                                                              \
  }
  // (the ids ride along with the first stage's loads, the rows they select with the second's)
  if constexpr (LSTM && HOIST) ICAP_BF16_CELL_IDS

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // One barrier per stage, below the next stage's loads: memory operations are not moved
  // across a barrier, so the prefetch stays at the top of the stage, in flight during the
  // multiplies; the staging registers then pass through an empty asm so that their first
  // use (a register-only conversion) and its wait stay below the multiplies.
  const int nk = p.K / BK;
  auto multiply_stage = [&](int buf) {
    const char* as = smem + buf * STAGE + (wm * WTM + l31) * ROW_B;
    const char* bs = smem + buf * STAGE + A_BYTES + (wn * 64 + l31) * ROW_B;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const bf16x8*>(as + i * 32 * ROW_B + rd_sw[ks]);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const bf16x8*>(bs + j * 32 * ROW_B + rd_sw[ks]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  };
  if constexpr (PAIR) {
    ICAP_BF16_LOAD_INTO(ra, rb, 0)
    ICAP_BF16_LOAD_INTO(ra2, rb2, 1)
    ICAP_BF16_STORE_FROM(ra, rb, 0)
    ICAP_BF16_STORE_FROM(ra2, rb2, 1)
    if constexpr (LSTM && HOIST) ICAP_BF16_CELL_OPERANDS
    ICAP_STAMP(1)
    for (int kt = 0; kt < nk; kt += 2) {
      // (the last trip re-loads its own pair: no branch around the loads)
      ICAP_BF16_LOAD_INTO(ra, rb, min(kt + 2, nk - 2))
      ICAP_BF16_LOAD_INTO(ra2, rb2, min(kt + 3, nk - 1))
      __syncthreads();  // the pair (kt, kt + 1) is in LDS
      multiply_stage(0);
      multiply_stage(1);
      ICAP_BF16_PIN(ra, rb)
      ICAP_BF16_PIN(ra2, rb2)
      __syncthreads();  // nobody reads the pair any more
      ICAP_BF16_STORE_FROM(ra, rb, 0)
      ICAP_BF16_STORE_FROM(ra2, rb2, 1)
    }
  } else {
    ICAP_BF16_LOAD(0)
    ICAP_BF16_STORE(0)
    if constexpr (LSTM && HOIST) ICAP_BF16_CELL_OPERANDS
    ICAP_STAMP(1)
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
      ICAP_BF16_LOAD(min(kt + 1, nk - 1))  // the last one is a harmless repeat: no branch
      __syncthreads();
      multiply_stage(cur);
      ICAP_BF16_PIN(ra, rb)
      ICAP_BF16_STORE(cur ^ 1)
      cur ^= 1;
    }
  }
  __syncthreads();  // the epilogues reuse the LDS
  ICAP_STAMP(2)

#undef ICAP_BF16_LOAD
#undef ICAP_BF16_STORE
#undef ICAP_BF16_LOAD_INTO
#undef ICAP_BF16_STORE_FROM
#undef ICAP_BF16_PIN

  if constexpr (TOPK > 0) {
    // The logits never reach memory: every wave writes its 64 x 64 patch (+ bias, -inf
    // beyond N) to LDS row-major (the staging buffers are idle after the last barrier),
    // then lane r folds row r of the patch into the row's record for these 64 columns.
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
          if (n_ok && m < p.M) {
            if constexpr (COH) __hip_atomic_store(p.y + (size_t)m * p.N + n, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else p.y[(size_t)m * p.N + n] = v;
            if (p.y16) {
              // blocked: [m/32][n/16][(n%16)/8][m%32][n%8] - a wave's MFMA A operand for a k-step
              // of 16 is then one contiguous kilobyte
              const size_t o = p.y16_blocked
                  ? ((((size_t)(m >> 5) * (p.N >> 4) + (n >> 4)) * 2 + ((n >> 3) & 1)) * 32 + (m & 31)) * 8 + (n & 7)
                  : (size_t)m * p.N + n;
              p.y16[o] = __builtin_bit_cast(uint16_t, (__bf16)v);
            }
          }
        }
    }
  } else {
    if constexpr (!HOIST) {
      ICAP_BF16_CELL_IDS
      ICAP_BF16_CELL_OPERANDS
    }
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
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int row = (tid >> 5) + q * (NT / 32);
      const int m = m0 + row;
      float z[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) z[g] = (zs[row * ZLD + g * 32 + u] + bz[g]) + ep_pre[q][g];
      const size_t o = (size_t)m * H + unit;
      if (m < p.M) {
        if (p.gru) {
          const float zg = fast_sigmoid(z[0]), rg = fast_sigmoid(z[1]);
          const float hh = fast_tanh(z[2] + rg * z[3]);
          p.h_out[o] = zg * st_pre[q] + (1.f - zg) * hh;
        } else {
          const float cn = fast_sigmoid(z[1]) * st_pre[q] + fast_sigmoid(z[0]) * fast_tanh(z[2]);
          p.c_out[o] = cn;
          p.h_out[o] = fast_sigmoid(z[3]) * fast_tanh(cn);
        }
      }
    }
  }
  ICAP_STAMP(3)
}

template <int BM, bool LSTM, int TOPK = 0, bool A16 = false, bool PAIR = false>
__global__ __launch_bounds__(256, BM == 64 ? 3 : 2) void gemm_bf16_kernel(Bf16Args p) {
  __shared__ __attribute__((aligned(16))) char smem[gemm_bf16_smem_bytes<BM, LSTM, TOPK>()];
  gemm_bf16_body<BM, LSTM, TOPK, A16, false, PAIR>(p, blockIdx.x, smem);
}

// The record projection on bf16 activations carrying the NEXT decode step's query projection
// and attention in the same launch (the bf16 counterpart of dense_topk_attq_kernel): that
// attention depends only on the recurrent state the cell kernel has written, not on this
// projection.  One launch on one stream replaces the side stream, its event dependencies (a
// wait and a record on the main stream per step, several microseconds each) and the query
// kernel.  Grid order: nq tiles of 64 x 128 of q = state . W2 + b2 (padded to a multiple of
// 8 workgroups, as is the attention section, so that the vocab tiles keep their XCD), the
// first tiles_before vocab tiles, the attention workgroups, the other vocab tiles.  The
// query tiles store q with device scope and count themselves on *q_done; an attention
// workgroup waits (one lane polling, with a sleep) until the count reaches q_target and
// reads q with device scope.  Workgroups start in grid order, so the query tiles never wait
// for a slot held by a waiting attention workgroup; should a wait run out all the same
// (2^16 polls, a fraction of a second) the workgroup raises *err and leaves its rows
// untouched - the caller turns that into an invalid result rather than a wrong one.
template <int TOPK, int RB>
__global__ __launch_bounds__(256, 2) void vocab_bf16_attq_kernel(Bf16Args p, Bf16Args pq, AttArgs a,
                                                                 int nq, int att_blocks,
                                                                 int tiles_before, unsigned* q_done,
                                                                 unsigned q_target, int* err) {
  __shared__ __attribute__((aligned(16))) char smem[gemm_bf16_smem_bytes<128, false, TOPK>()];
  static_assert(gemm_bf16_smem_bytes<64, false, 0>() <= gemm_bf16_smem_bytes<128, false, TOPK>(),
                "the query tile's LDS lives in the vocab tile's");
  const int nq_pad = (nq + 7) & ~7, att_pad = (att_blocks + 7) & ~7;
  int b = blockIdx.x;
  int tile_id = -1;
  if (b >= nq_pad) {
    if (b < nq_pad + tiles_before) { tile_id = b - nq_pad; }
    else if (b >= nq_pad + tiles_before + att_pad) { tile_id = b - nq_pad - att_pad; }
    else { b -= tiles_before; }
  }
  if (tile_id >= 0) {
    gemm_bf16_body<128, false, TOPK, true>(p, tile_id, smem);
    return;
  }
  if (b < nq_pad) {
    if (b >= nq) return;
    gemm_bf16_body<64, false, 0, false, true>(pq, b, smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every store of the tile acknowledged
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(q_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  if (b - nq_pad >= att_blocks) return;
  int* s_ok = reinterpret_cast<int*>(smem);
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
  if (!ok) return;
  __builtin_amdgcn_s_setprio(3);
  attention_body<256, RB, 2, true, ATT_ACT_TANH, true>(a, smem, b - nq_pad, att_blocks);
}

// w[K_src][N] fp32 -> packed[n'][k_off + k] bf16, n' = n or the gate-interleaved column.
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w,
                                                          uint16_t* __restrict__ packed, int K_src,
                                                          int N, int k_off, int K_total,
                                                          int gate_units) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)K_src * N) return;
  const int k = (int)(idx / N), n = (int)(idx - (long long)k * N);
  int np = n;
  if (gate_units > 0) {
    const int g = n / gate_units, u = n - g * gate_units;
    np = (u >> 5) * 128 + g * 32 + (u & 31);
  }
  const __bf16 v = (__bf16)w[idx];
  packed[(size_t)np * K_total + k_off + k] = __builtin_bit_cast(uint16_t, v);
}

template <int BM, bool LSTM>
int launch(Bf16Args& a, hipStream_t s, const char* name) {
  a.tiles_m = ceil_div(a.M, BM);
  a.tiles_n = ceil_div(a.N, BN);
  hipLaunchKernelGGL((gemm_bf16_kernel<BM, LSTM>), dim3(a.tiles_m * a.tiles_n), dim3(256), 0, s, a);
  ICAP_LAUNCH_CHECK(name);
  return ICAP_OK;
}

}  // namespace

size_t packed_weight_bytes(int K, int N) {
  if (K <= 0 || N <= 0) return 0;
  return (size_t)ceil_div(N, BN) * BN * K * sizeof(uint16_t);
}

int pack_weight_bf16(const float* w, void* packed, int K_src, int N, int k_offset, int K_total,
                     int gate_units, hipStream_t s) {
  ICAP_REQUIRE(w && packed, "pack_weight_bf16: null pointer");
  ICAP_REQUIRE(K_src > 0 && N > 0 && k_offset >= 0 && k_offset + K_src <= K_total,
               "pack_weight_bf16: rows [%d,%d) outside [0,%d)", k_offset, k_offset + K_src, K_total);
  ICAP_REQUIRE(aligned16(packed), "pack_weight_bf16: packed buffer must be 16-byte aligned");
  if (gate_units > 0)
    ICAP_REQUIRE(N == 4 * gate_units && gate_units % 32 == 0,
                 "pack_weight_bf16: gate layout needs N == 4*units and units %% 32 == 0");
  const int n_pad = ceil_div(N, BN) * BN;
  if (n_pad != N && k_offset == 0)  // zero rows for the padded columns
    ICAP_HIP(hipMemsetAsync(static_cast<uint16_t*>(packed) + (size_t)N * K_total, 0,
                            (size_t)(n_pad - N) * K_total * sizeof(uint16_t), s));
  const long long total = (long long)K_src * N;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w,
                     static_cast<uint16_t*>(packed), K_src, N, k_offset, K_total, gate_units);
  ICAP_LAUNCH_CHECK("pack_weight_kernel");
  return ICAP_OK;
}

int dense_bf16(const float* x, const void* packed_w, const float* bias, float* y, int M, int K,
               int N, int act, hipStream_t s) {
  ICAP_REQUIRE(x && packed_w && y, "dense_bf16: null pointer");
  ICAP_REQUIRE(M > 0 && K > 0 && N > 0, "dense_bf16: bad shape M=%d K=%d N=%d", M, K, N);
  ICAP_REQUIRE(K % BK == 0, "dense_bf16: K=%d must be a multiple of %d", K, BK);
  ICAP_REQUIRE(aligned16(x) && aligned16(packed_w), "dense_bf16: operands must be 16-byte aligned");
  ICAP_REQUIRE(act == ICAP_ACT_NONE || act == ICAP_ACT_RELU, "dense_bf16: unknown activation %d", act);
  Bf16Args a = {};
  a.a0 = x; a.a1 = nullptr; a.K0 = K; a.lda0 = K; a.lda1 = 0;
  a.w = static_cast<const uint16_t*>(packed_w);
  a.bias = bias; a.y = y; a.M = M; a.N = N; a.K = K; a.act = act;
  // small problems: 64-row tiles keep more compute units busy
  if ((long long)ceil_div(M, 128) * ceil_div(N, BN) >= 256) return launch<128, false>(a, s, "gemm_bf16_kernel<128>");
  return launch<64, false>(a, s, "gemm_bf16_kernel<64>");
}

// dense_bf16 that also writes its result rounded to bf16 ([M][N]): the A operand of a
// following dense_topk_bf16_a16.
int dense_bf16_dual(const float* x, const void* packed_w, const float* bias, float* y, void* y16,
                    int y16_blocked, int M, int K, int N, int act, hipStream_t s) {
  ICAP_REQUIRE(!y16_blocked || N % 16 == 0, "dense_bf16_dual: the blocked layout needs N %% 16 == 0");
  ICAP_REQUIRE(x && packed_w && y && y16, "dense_bf16_dual: null pointer");
  ICAP_REQUIRE(M > 0 && K > 0 && N > 0 && K % BK == 0, "dense_bf16_dual: bad shape M=%d K=%d N=%d", M, K, N);
  ICAP_REQUIRE(aligned16(x) && aligned16(packed_w), "dense_bf16_dual: operands must be 16-byte aligned");
  Bf16Args a = {};
  a.a0 = x; a.a1 = nullptr; a.K0 = K; a.lda0 = K; a.lda1 = 0;
  a.w = static_cast<const uint16_t*>(packed_w);
  a.bias = bias; a.y = y; a.M = M; a.N = N; a.K = K; a.act = act;
  a.y16 = static_cast<uint16_t*>(y16);
  a.y16_blocked = y16_blocked;
  if ((long long)ceil_div(M, 128) * ceil_div(N, BN) >= 256) return launch<128, false>(a, s, "gemm_bf16_kernel<128>");
  return launch<64, false>(a, s, "gemm_bf16_kernel<64>");
}

bool dense_topk_bf16_supported(const float* x, int M, int K, int N, int top_k) {
  return (top_k == 1 || top_k == 3 || top_k == 5) && M > 0 && K > 0 && N > 0 && K % BK == 0 &&
         aligned16(x);
}

// icap_dense_topk_f32 with bf16 operands: same records (see include/icap.h), packed weights.
int dense_topk_bf16(const float* x, const void* packed_w, const float* bias, int M, int K, int N,
                    int eos, int top_k, float* records, float* eos_logit, const LogitMasks* masks,
                     hipStream_t s) {
  ICAP_REQUIRE(x && packed_w && records && eos_logit, "dense_topk_bf16: null pointer");
  ICAP_REQUIRE(dense_topk_bf16_supported(x, M, K, N, top_k),
               "dense_topk_bf16: needs top_k in {1,3,5}, K %% 64 == 0 and a 16-byte aligned x "
               "(M=%d K=%d N=%d top_k=%d)", M, K, N, top_k);
  ICAP_REQUIRE(aligned16(packed_w) && aligned16(records), "dense_topk_bf16: unaligned buffer");
  Bf16Args a = {};
  a.a0 = x; a.a1 = nullptr; a.K0 = K; a.lda0 = K; a.lda1 = 0;
  a.w = static_cast<const uint16_t*>(packed_w);
  a.bias = bias; a.y = nullptr; a.M = M; a.N = N; a.K = K; a.act = ICAP_ACT_NONE;
  a.rec = records; a.eos_x = eos_logit; a.eos = eos;
  if (masks) a.masks = *masks;
  a.tiles_m = ceil_div(M, 128);
  a.tiles_n = ceil_div(N, BN);
  const dim3 grid(a.tiles_m * a.tiles_n), block(256);
  if (top_k == 1) hipLaunchKernelGGL((gemm_bf16_kernel<128, false, 1>), grid, block, 0, s, a);
  else if (top_k == 3) hipLaunchKernelGGL((gemm_bf16_kernel<128, false, 3>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((gemm_bf16_kernel<128, false, 5>), grid, block, 0, s, a);
  ICAP_LAUNCH_CHECK("gemm_bf16_kernel<topk>");
  return ICAP_OK;
}

bool vocab_bf16_resident_supported(int M, int K, int N, int top_k);
int vocab_bf16_resident(const void*, const void*, const float*, int, int, int, int, int, float*, float*,
                        const LogitMasks*, int, hipStream_t);

// dense_topk_bf16 on an activation matrix that is already bf16 (from dense_bf16_dual): same
// records.  blocked: x16 in the operand-block layout, taken by the kernel with the weight
// panel resident in LDS (vocab_bf16.hip; served for K = 256 or 512, see
// dense_topk_bf16_a16_blocked; opt-in with ICAP_BF16_RESIDENT=1: alone it is a little faster
// than the staged tiles, beside the side-stream attention it is slower, because its one
// workgroup per CU with 128 KB of LDS leaves that kernel nowhere to run);
// otherwise x16 is row-major [M][K].
bool dense_topk_bf16_a16_blocked(int M, int K, int N, int top_k) {
  const char* res_env = getenv("ICAP_BF16_RESIDENT");  // (read per call: a test switches it)
  return res_env && res_env[0] == '1' && vocab_bf16_resident_supported(M, K, N, top_k);
}

int dense_topk_bf16_a16(const void* x16, int blocked, const void* packed_w, const float* bias, int M,
                        int K, int N, int eos, int top_k, float* records, float* eos_logit,
                        const LogitMasks* masks, hipStream_t s) {
  ICAP_REQUIRE(x16 && packed_w && records && eos_logit, "dense_topk_bf16_a16: null pointer");
  ICAP_REQUIRE((top_k == 1 || top_k == 3 || top_k == 5) && M > 0 && N > 0 && K > 0 && K % BK == 0,
               "dense_topk_bf16_a16: needs top_k in {1,3,5} and K %% 64 == 0 (M=%d K=%d N=%d top_k=%d)",
               M, K, N, top_k);
  ICAP_REQUIRE(aligned16(x16) && aligned16(packed_w) && aligned16(records),
               "dense_topk_bf16_a16: unaligned buffer");
  ICAP_REQUIRE(!blocked || vocab_bf16_resident_supported(M, K, N, top_k),
               "dense_topk_bf16_a16: the blocked activation layout is the resident kernel's");
  {
    if (blocked) {
      static int cus = 0;  // compute units of the device (the first caller's; one GPU model per process)
      if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
          cus = n;
        else
          cus = 256;
      }
      return vocab_bf16_resident(x16, packed_w, bias, M, K, N, eos, top_k, records, eos_logit, masks,
                                 cus, s);
    }
  }
  Bf16Args a = {};
  a.a16 = static_cast<const uint16_t*>(x16); a.lda16 = K; a.K0 = K;
  a.w = static_cast<const uint16_t*>(packed_w);
  a.bias = bias; a.y = nullptr; a.M = M; a.N = N; a.K = K; a.act = ICAP_ACT_NONE;
  a.rec = records; a.eos_x = eos_logit; a.eos = eos;
  if (masks) a.masks = *masks;
  a.tiles_m = ceil_div(M, 128);
  a.tiles_n = ceil_div(N, BN);
  {
    static const char* st_env = getenv("ICAP_BF16_STAGGER_US");  // tuning knob: microseconds, 0 = off
    const double us = st_env ? atof(st_env) : 0.0;
    a.stagger_ticks = (int)(us * 100.0);
    a.stagger_wgs = 512;  // two workgroups on each of 256 compute units
  }
  const dim3 grid(a.tiles_m * a.tiles_n), block(256);
  // Two K stages per loop trip (ICAP_BF16_PAIR=1, opt-in): measured no gain - the K loop of a
  // tile takes 7.3-7.6 us either way (device timelines), i.e. it runs at the rate the operands
  // arrive (256 KB per tile, 512 tiles at a time: 18 TB/s out of the L2s), not at the latency
  // of a stage; 188k against 192k captions/s at B = 512, beam 5.
  static const char* pair_env = getenv("ICAP_BF16_PAIR");
  const bool pair = (K / BK) % 2 == 0 && pair_env && pair_env[0] == '1';
  if (pair) {
    if (top_k == 1) hipLaunchKernelGGL((gemm_bf16_kernel<128, false, 1, true, true>), grid, block, 0, s, a);
    else if (top_k == 3) hipLaunchKernelGGL((gemm_bf16_kernel<128, false, 3, true, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((gemm_bf16_kernel<128, false, 5, true, true>), grid, block, 0, s, a);
  } else {
    if (top_k == 1) hipLaunchKernelGGL((gemm_bf16_kernel<128, false, 1, true>), grid, block, 0, s, a);
    else if (top_k == 3) hipLaunchKernelGGL((gemm_bf16_kernel<128, false, 3, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((gemm_bf16_kernel<128, false, 5, true>), grid, block, 0, s, a);
  }
  ICAP_LAUNCH_CHECK("gemm_bf16_kernel<topk,a16>");
  return ICAP_OK;
}

// dense_topk_bf16_a16 (row-major x16) with the next step's query projection and attention in
// the same launch (vocab_bf16_attq_kernel).  q [G*R, U] is scratch the launch fills; q_done
// counts finished query tiles over the launches of a decode (the caller zeroes it once and
// passes the count this launch must reach); err is raised if an attention workgroup's wait
// runs out.
bool dense_topk_bf16_attq_supported(int R, int L, int U, int D, int H, int top_k) {
  if (R < 1 || R > 5 || L < 1 || U < 4 || U > 512 || U % 4 || D < 4 || D % 4 || H < BK || H % BK)
    return false;
  if (top_k != (R <= 1 ? 1 : (R <= 3 ? 3 : 5))) return false;
  return sizeof(float) * attention_lds_floats(256, R, R, L, D) <= 64 * 1024;
}

int dense_topk_bf16_attq_tiles(int rows, int U) { return ceil_div(rows, 64) * ceil_div(U, BN); }

int dense_topk_bf16_attq(const void* x16, const void* packed_w, const float* bias, int M, int K, int N,
                         int eos, int top_k, float* records, float* eos_logit,
                         const LogitMasks* masks, const float* pf, const float* features,
                         const float* att_v, float att_bv, const float* state, const float* gate_w,
                         float gate_b, const void* packed_q_w, const float* q_b, float* q, float* ctx,
                         int G, int R, int stride_g, int stride_k, int L, int U, int D, int H,
                         int att_blocks, unsigned* q_done, unsigned q_target, int* err,
                         hipStream_t s, const float* epf, const int* pf_extreme) {
  ICAP_REQUIRE(x16 && packed_w && records && eos_logit, "dense_topk_bf16_attq: null pointer");
  ICAP_REQUIRE(pf && features && att_v && state && gate_w && packed_q_w && q && ctx && q_done && err,
               "dense_topk_bf16_attq: null attention pointer");
  ICAP_REQUIRE((top_k == 1 || top_k == 3 || top_k == 5) && M > 0 && N > 0 && K > 0 && K % BK == 0,
               "dense_topk_bf16_attq: needs top_k in {1,3,5} and K %% 64 == 0 (M=%d K=%d N=%d top_k=%d)",
               M, K, N, top_k);
  ICAP_REQUIRE(G > 0 && dense_topk_bf16_attq_supported(R, L, U, D, H, top_k),
               "dense_topk_bf16_attq: attention shape not served (R=%d L=%d U=%d D=%d H=%d)", R, L, U, D, H);
  ICAP_REQUIRE(aligned16(x16) && aligned16(packed_w) && aligned16(records) && aligned16(pf) &&
                   aligned16(features) && aligned16(att_v) && aligned16(packed_q_w) && aligned16(ctx) &&
                   aligned16(state) && aligned16(q),
               "dense_topk_bf16_attq: operands must be 16-byte aligned");
  Bf16Args a = {};
  a.a16 = static_cast<const uint16_t*>(x16); a.lda16 = K; a.K0 = K;
  a.w = static_cast<const uint16_t*>(packed_w);
  a.bias = bias; a.y = nullptr; a.M = M; a.N = N; a.K = K; a.act = ICAP_ACT_NONE;
  a.rec = records; a.eos_x = eos_logit; a.eos = eos;
  if (masks) a.masks = *masks;
  a.tiles_m = ceil_div(M, 128);
  a.tiles_n = ceil_div(N, BN);
  // q = state . q_w + q_b over all G*R rows, 64 x 128 tiles
  const int rows = G * R;
  Bf16Args aq = {};
  aq.a0 = state; aq.a1 = nullptr; aq.K0 = H; aq.lda0 = H; aq.lda1 = 0;
  aq.w = static_cast<const uint16_t*>(packed_q_w);
  aq.bias = q_b; aq.y = q; aq.M = rows; aq.N = U; aq.K = H; aq.act = ICAP_ACT_NONE;
  aq.tiles_m = ceil_div(rows, 64);
  aq.tiles_n = ceil_div(U, BN);
  const int nq = aq.tiles_m * aq.tiles_n;
  att_blocks = att_blocks > 0 ? (att_blocks < G ? att_blocks : G) : G;
  AttArgs t{pf, features, q, att_v, att_bv, nullptr, state, gate_w, gate_b, ctx, nullptr,
            G, R, stride_g, stride_k, L, U, D, H, att_blocks, nullptr, nullptr, nullptr};
  t.epf = epf;
  t.pf_extreme = pf_extreme;
  const int tiles = a.tiles_m * a.tiles_n;
  const dim3 grid(((nq + 7) & ~7) + ((att_blocks + 7) & ~7) + tiles), block(256);
  // Where the attention workgroups sit in the grid (tuning knob ICAP_BF16_ATTQ_TILES_BEFORE):
  // behind one round of resident tiles (two per compute unit) less the query tiles.
  static const char* tb_env = getenv("ICAP_BF16_ATTQ_TILES_BEFORE");
  const int slots = 512, nq_pad = (nq + 7) & ~7;
  int tiles_before = tb_env ? atoi(tb_env) : slots - nq_pad;
  tiles_before &= ~7;
  if (tiles_before < 0) tiles_before = 0;
  if (tiles_before > tiles) tiles_before = tiles & ~7;
#define ICAP_ATTQ16(TK, RB_) hipLaunchKernelGGL((vocab_bf16_attq_kernel<TK, RB_>), grid, block, 0, s, a, aq, t, nq, att_blocks, tiles_before, q_done, q_target, err)
  switch (R) {
    case 1: ICAP_ATTQ16(1, 1); break;
    case 2: ICAP_ATTQ16(3, 2); break;
    case 3: ICAP_ATTQ16(3, 3); break;
    case 4: ICAP_ATTQ16(5, 4); break;
    default: ICAP_ATTQ16(5, 5); break;
  }
#undef ICAP_ATTQ16
  ICAP_LAUNCH_CHECK("vocab_bf16_attq_kernel");
  return ICAP_OK;
}

namespace {
int cell_step_bf16(int gru, const float* ctx, const float* h, const float* c, const void* packed_w,
                   const float* bias, const float* emb_proj, const int32_t* ids, float* h_out,
                   float* c_out, int N, int C, int H, int V, hipStream_t s);
}

int lstm_step_bf16(const float* ctx, const float* h, const float* c, const void* packed_w,
                   const float* bias, const float* emb_proj, const int32_t* ids, float* h_out,
                   float* c_out, int N, int C, int H, int V, hipStream_t s) {
  ICAP_REQUIRE(c && c_out, "lstm_step_bf16: null pointer");
  return cell_step_bf16(0, ctx, h, c, packed_w, bias, emb_proj, ids, h_out, c_out, N, C, H, V, s);
}

// GRU step with bf16 operands: packed_w holds the four-column operands of icap_gru_pack_f32
// (kernel4[0:C] then recurrent4, gates interleaved), bias4 in [4H] order.
int gru_step_bf16(const float* ctx, const float* h, const void* packed_w, const float* bias4,
                  const float* emb_proj, const int32_t* ids, float* h_out, int N, int C, int H, int V,
                  hipStream_t s) {
  return cell_step_bf16(1, ctx, h, nullptr, packed_w, bias4, emb_proj, ids, h_out, nullptr, N, C, H, V,
                        s);
}

// The cell step in two launches, for a schedule that wants the gate product off its critical
// path: the product [ctx | h] . W does not depend on the token the beam step is about to pick,
// only the token's gate rows emb_proj[id] and the cell update do.
//   cell_gates_bf16   z[N][4H] = [ctx | h] . W   (columns in the packed, gate-interleaved order:
//                     column (u / 32) * 128 + g * 32 + u % 32 is gate g of unit u; no bias)
//   cell_update_f32   (z + bias) + emb_proj[id], then the LSTM / GRU update of cell_step_bf16's
//                     epilogue, operation for operation: the two launches give the bits of the one.
int cell_gates_bf16(const float* ctx, const float* h, const void* packed_w, float* z, int N, int C,
                    int H, hipStream_t s) {
  ICAP_REQUIRE(ctx && h && packed_w && z, "cell_gates_bf16: null pointer");
  ICAP_REQUIRE(N > 0 && C > 0 && H > 0 && C % BK == 0 && H % BK == 0,
               "cell_gates_bf16: C=%d and H=%d must be positive multiples of %d", C, H, BK);
  ICAP_REQUIRE(aligned16(ctx) && aligned16(h) && aligned16(packed_w),
               "cell_gates_bf16: operands must be 16-byte aligned");
  Bf16Args a = {};
  a.a0 = ctx; a.a1 = h; a.K0 = C; a.lda0 = C; a.lda1 = H;
  a.w = static_cast<const uint16_t*>(packed_w);
  a.bias = nullptr; a.y = z; a.M = N; a.N = 4 * H; a.K = C + H; a.act = ICAP_ACT_NONE;
  const long long tiles64 = (long long)ceil_div(N, 64) * ceil_div(4 * H, BN);
  if (tiles64 > 768) return launch<128, false>(a, s, "gemm_bf16_kernel<gates,128>");
  return launch<64, false>(a, s, "gemm_bf16_kernel<gates>");
}

namespace {
__global__ __launch_bounds__(256) void cell_update_kernel(const float* __restrict__ z,
                                                          const float* __restrict__ bias,
                                                          const float* __restrict__ emb_proj,
                                                          const int32_t* __restrict__ ids,
                                                          const float* __restrict__ state_in,
                                                          float* __restrict__ h_out,
                                                          float* __restrict__ c_out, int N, int H, int V,
                                                          int gru) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)N * H) return;
  const int m = (int)(idx / H), unit = (int)(idx - (long long)m * H);
  const float* zrow = z + (size_t)m * 4 * H + (unit >> 5) * 128 + (unit & 31);
  const int id = emb_proj ? min(max(ids[m], 0), V - 1) : 0;
  const float* ep = emb_proj ? emb_proj + (size_t)id * 4 * H + unit : nullptr;
  float zz[4];
#pragma unroll
  for (int g = 0; g < 4; ++g)
    zz[g] = (zrow[g * 32] + (bias ? bias[g * H + unit] : 0.f)) + (ep ? ep[g * H] : 0.f);
  const float st = state_in[idx];
  if (gru) {
    const float zg = fast_sigmoid(zz[0]), rg = fast_sigmoid(zz[1]);
    const float hh = fast_tanh(zz[2] + rg * zz[3]);
    h_out[idx] = zg * st + (1.f - zg) * hh;
  } else {
    const float cn = fast_sigmoid(zz[1]) * st + fast_sigmoid(zz[0]) * fast_tanh(zz[2]);
    c_out[idx] = cn;
    h_out[idx] = fast_sigmoid(zz[3]) * fast_tanh(cn);
  }
}
}  // namespace

// state_in: the previous cell state c (LSTM; c_out may alias it) or the previous h (GRU, c_out null)
int cell_update_f32(int gru, const float* z, const float* bias, const float* emb_proj,
                    const int32_t* ids, const float* state_in, float* h_out, float* c_out, int N,
                    int H, int V, hipStream_t s) {
  ICAP_REQUIRE(z && state_in && h_out && (gru || c_out), "cell_update_f32: null pointer");
  ICAP_REQUIRE(N > 0 && H > 0 && H % 32 == 0, "cell_update_f32: H=%d must be a positive multiple of 32", H);
  ICAP_REQUIRE((emb_proj == nullptr) == (ids == nullptr), "cell_update_f32: emb_proj and ids come together");
  ICAP_REQUIRE(!emb_proj || V > 0, "cell_update_f32: vocabulary size missing");
  ICAP_REQUIRE(!gru || h_out != state_in, "cell_update_f32: h_out must not alias the previous h");
  const long long total = (long long)N * H;
  hipLaunchKernelGGL(cell_update_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, z, bias,
                     emb_proj, ids, state_in, h_out, c_out, N, H, V, gru);
  ICAP_LAUNCH_CHECK("cell_update_kernel");
  return ICAP_OK;
}

namespace {
int cell_step_bf16(int gru, const float* ctx, const float* h, const float* c, const void* packed_w,
                   const float* bias, const float* emb_proj, const int32_t* ids, float* h_out,
                   float* c_out, int N, int C, int H, int V, hipStream_t s) {
  ICAP_REQUIRE(ctx && h && packed_w && h_out, "lstm_step_bf16: null pointer");
  ICAP_REQUIRE(N > 0 && C > 0 && H > 0, "lstm_step_bf16: bad shape");
  ICAP_REQUIRE(C % BK == 0 && H % BK == 0, "lstm_step_bf16: C=%d and H=%d must be multiples of %d", C,
               H, BK);
  ICAP_REQUIRE((emb_proj == nullptr) == (ids == nullptr),
               "lstm_step_bf16: emb_proj and ids come together");
  ICAP_REQUIRE(!emb_proj || V > 0, "lstm_step_bf16: vocabulary size missing");
  ICAP_REQUIRE(h_out != h, "lstm_step_bf16: h_out must not alias h");
  ICAP_REQUIRE(aligned16(ctx) && aligned16(h) && aligned16(packed_w),
               "lstm_step_bf16: operands must be 16-byte aligned");
  Bf16Args a = {};
  a.a0 = ctx; a.a1 = h; a.K0 = C; a.lda0 = C; a.lda1 = H;
  a.w = static_cast<const uint16_t*>(packed_w);
  a.bias = bias; a.y = nullptr; a.M = N; a.N = 4 * H; a.K = C + H; a.act = ICAP_ACT_NONE;
  a.c_in = c; a.h_out = h_out; a.c_out = c_out; a.emb_proj = emb_proj; a.ids = ids; a.H = H; a.V = V;
  a.gru = gru;
  // 64-row tiles while they fit the chip at once (three workgroups per compute unit); beyond
  // that a second, partly filled round of workgroups costs more than 128-row tiles do (the K
  // loop runs at the rate operands arrive from L2, and taller tiles fetch the weights half as often)
  static const char* bm_env = getenv("ICAP_BF16_CELL_BM");  // tuning knob: "64" | "128"
  const long long tiles64 = (long long)ceil_div(N, 64) * ceil_div(4 * H, BN);
  const bool tall = bm_env ? atoi(bm_env) == 128 : tiles64 > 768;
  if (tall) return launch<128, true>(a, s, "gemm_bf16_kernel<lstm,128>");
  return launch<64, true>(a, s, "gemm_bf16_kernel<lstm>");
}
}  // namespace

}  // namespace icap
