// The vocab projection of the bf16 mode with its top-k record epilogue, built around what
// bounds it at K = 512: not the matrix pipe (26 GFLOP are 10 us of v_mfma_f32_32x32x16_bf16) but
// operand traffic and latency - a 128 x 128 x 512 tile is 1.7 us of multiplies fed by 256 KB
// of operands, and a staged, barrier-per-stage pipeline waits out a memory latency per stage.
//
// Here a workgroup (8 waves, one per compute unit) keeps ONE 128-column weight panel resident
// in LDS for its whole life ([128][K] bf16 = 128 KB, read from memory once per workgroup) and
// its waves run decoupled, without a barrier: a wave takes a block of 32 activation rows,
// holds their K values in registers in the MFMA operand layout (bf16 activations written by
// the producing kernel, 16 bytes per lane and MFMA, the next half of K or the next block
// prefetched into a second register set), multiplies them against the panel's four 32-column
// blocks (one ds_read_b128 per MFMA, conflict-free through an XOR swizzle of the 16-byte
// chunks) and reduces the 32 x 128 logits to the rows' records at once, through a 4 KB
// swizzled transposition patch of its own.  LDS: 128 KB panel + 8 x 4 KB patches = all 160 KB.
#include "icap_common.h"
#include "topk_epilogue.h"

namespace icap {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int VB_NT = 512, VB_WAVES = 8, VB_BN = 128;
constexpr int VB_PANEL_BYTES = 131072, VB_PATCH_BYTES = 4096;

struct VocabArgs {
  const uint16_t* a16;  // bf16 [ceil(M/32)][K/16][2][32][8]: block rb, k-step s = one kilobyte, lane-ordered
  const uint16_t* w;    // packed [N_pad][K] bf16
  const float* bias;
  float* rec;    // [M][2*tiles_n][2 + 2*TOPK]
  float* eos_x;  // [M]
  int M, N, K, eos;
  int tiles_n, groups;  // grid = tiles_n * groups; a group's waves stride the row blocks
  LogitMasks masks;
};

// lane-wise select on a wave mask (the 64-bit result of a vector compare): mask ? a : b.  As
// inline asm because the compiler otherwise turns the insertion's selects on constants into
// branches (an exec-masked region and a jump per select: five times the instructions).
__device__ __forceinline__ float sel_f(unsigned long long mask, float a, float b) {
  float r;
  asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(mask));
  return r;
}
__device__ __forceinline__ int sel_i(unsigned long long mask, int a, int b) {
  int r;
  asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(mask));
  return r;
}

// Inserts (v, c) into a descending list; strict '>' keeps the earlier (lower) column on ties.
// The list is sorted, so the new entry k is the median of (entry k-1, entry k, v): one
// v_med3_f32 per level; the columns follow with two selects per level.
template <int TOPK>
__device__ __forceinline__ void topk_insert(float (&tx)[TOPK], int (&tc)[TOPK], float v, int c) {
  unsigned long long gt[TOPK];
#pragma unroll
  for (int k = 0; k < TOPK; ++k) gt[k] = __builtin_amdgcn_ballot_w64(v > tx[k]);
#pragma unroll
  for (int k = TOPK - 1; k >= 1; --k) {
    tc[k] = sel_i(gt[k - 1], tc[k - 1], sel_i(gt[k], c, tc[k]));
    tx[k] = __builtin_amdgcn_fmed3f(tx[k - 1], tx[k], v);
  }
  tc[0] = sel_i(gt[0], c, tc[0]);
  tx[0] = __builtin_fmaxf(tx[0], v);
}

template <int TOPK, int KH>  // KH = K / 256
__global__ __launch_bounds__(VB_NT, 1) void vocab_bf16_resident_kernel(VocabArgs p) {
  constexpr int K = KH * 256;
  constexpr int ROW_BYTES = K * 2;          // one panel column: K bf16
  constexpr int CHUNKS = K / 8;             // 16-byte chunks per panel column
  constexpr int RL = 2 + 2 * TOPK;
  __shared__ __attribute__((aligned(16))) char smem[VB_PANEL_BYTES + VB_WAVES * VB_PATCH_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int tile_n = blockIdx.x % p.tiles_n;
  const int group = blockIdx.x / p.tiles_n;
  const int n0 = tile_n * VB_BN;

  // ---- the weight panel, once: chunk ch of column n goes to chunk ch ^ (n & 15).  Every
  // load of a thread is issued before its first store (one memory latency, not sixteen) ----
  {
    constexpr int PER = VB_BN * CHUNKS / VB_NT;  // 8 (K = 256) or 16 (K = 512)
    uint4 v[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int i = tid + q * VB_NT;
      const int n = i / CHUNKS, ch = i - n * CHUNKS;
      v[q] = *reinterpret_cast<const uint4*>(p.w + (size_t)(n0 + n) * K + ch * 8);
    }
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int i = tid + q * VB_NT;
      const int n = i / CHUNKS, ch = i - n * CHUNKS;
      *reinterpret_cast<uint4*>(smem + n * ROW_BYTES + ((ch ^ (n & 15)) << 4)) = v[q];
    }
  }
  __syncthreads();

  float* patch = reinterpret_cast<float*>(smem + VB_PANEL_BYTES + wave * VB_PATCH_BYTES);
  const char* bcol = smem + l31 * ROW_BYTES;  // this lane's column inside a 32-column block
  const int swz = l31 & 15;
  const int n_rb = (p.M + 31) >> 5;
  const int rb_step = VB_WAVES * p.groups;

  // A operand: lane (l31, half) holds 8 consecutive k of row l31 per MFMA: chunk 2*s + half
  // of the row for k-step s.  A rolling window of 16 k-steps in registers: the slot a step
  // has consumed is refilled at once with the chunk 16 steps ahead (of this block, then of
  // the wave's next one), so 64 MFMAs (2048 cycles) cover every load.
  constexpr int STEPS = K / 16;  // 16 (K = 256) or 32 (K = 512)
  uint4 win[16];
  auto a_chunk = [&](int rb_, int s_) -> uint4 {  // (rows past M hold whatever: their records are never stored)
    return reinterpret_cast<const uint4*>(p.a16)[((size_t)rb_ * STEPS + s_) * 64 + lane];
  };
  // Panel fragment of k-step s = 8*sh + sl: chunk (16*sh + 2*sl + half) ^ swz.  The swizzle
  // touches the low four bits only, so the address is one of eight lane-dependent bases (per
  // sl) plus constants - eight address registers instead of one per step.
  const char* bsl[8];
#pragma unroll
  for (int sl = 0; sl < 8; ++sl) bsl[sl] = bcol + ((((2 * sl) | half) ^ swz) << 4);

  int rb = group * VB_WAVES + wave;
  if (rb < n_rb) {
#pragma unroll
    for (int s = 0; s < 16; ++s) win[s] = a_chunk(rb, s);
  }
  for (; rb < n_rb; rb += rb_step) {
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // (past the wave's last block the prefetch repeats that block's rows: no branch around loads)
    const int rb_next = min(rb + rb_step, n_rb - 1);
    // the panel fragments of step s + 1 are read before the multiplies of step s (explicit
    // double buffer; the scheduling barriers keep the compiler from hoisting further reads
    // and running out of registers)
    bf16x8 bq[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      bq[0][j] = *reinterpret_cast<const bf16x8*>(bsl[0] + j * 32 * ROW_BYTES);
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      const bf16x8 av = __builtin_bit_cast(bf16x8, win[s & 15]);
      // refill the slot: 16 steps ahead in this block, or the start of the next block
      win[s & 15] = (s + 16 < STEPS) ? a_chunk(rb, s + 16) : a_chunk(rb_next, s + 16 - STEPS);
      if (s + 1 < STEPS) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          bq[(s + 1) & 1][j] = *reinterpret_cast<const bf16x8*>(
              bsl[(s + 1) & 7] + ((s + 1) >> 3) * 256 + j * 32 * ROW_BYTES);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bq[s & 1][j], acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- records: two 64-column half tiles, each two 32 x 32 accumulator blocks through the
    // wave's patch.  Element (row r, column c) of a block sits at r*32 + ((c/4) ^ ((r/2)&7))*4
    // + c%4: the 32 lanes of a store (one row) and the 16 lanes of a 16-byte load group (rows
    // of both parities, distinct r/2 & 7 per parity) touch every bank once.
    // (The lane's coordinates pass through an empty asm in every iteration: everything the
    // records code derives from them - sixteen patch addresses, sixty-four column numbers and
    // their comparisons with the eos column - is the same for every row block, and hoisted
    // out of the loop it costs a hundred registers; recomputed it costs a few dozen adds.)
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    const int l31_o = lane_o & 31, half_o = lane_o >> 5;
    const int lr = lane_o >> 1, part = lane_o & 1;
    const int grow = rb * 32 + lr;
    const int rswz = (lr >> 1) & 7;
    // patch store address of accumulator register r: row rho = (r&3) + 8*(r>>2) + 4*half,
    // chunk (l31>>2) ^ ((rho>>1)&7) = wt ^ kr with kr in {0,1,4,5} a constant of r
    const int wt = (l31_o >> 2) ^ (2 * half_o);
    const int wlow = (l31_o & 3) + 128 * half_o;
#pragma unroll
    for (int ht2 = 0; ht2 < 2; ++ht2) {
      TopkRow<TOPK> rec;
#pragma unroll
      for (int k = 0; k < TOPK; ++k) { rec.tx[k] = -INFINITY; rec.tc[k] = -1; }
      rec.eos_v = 0.f;
      rec.has_eos = false;
      float m = -INFINITY, sum = 0.f;
#pragma unroll
      for (int jb = 0; jb < 2; ++jb) {
        const int j = ht2 * 2 + jb;
        {
          const int col = n0 + j * 32 + l31_o;
          const bool ok = col < p.N;
          const float bv = (p.bias && ok) ? p.bias[col] : 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int kr = ((r >> 1) & 1) | (((r >> 2) & 1) << 2);
            patch[((r & 3) + 8 * (r >> 2)) * 32 + ((wt ^ kr) << 2) + wlow] =
                ok ? acc[j][r] + bv : -INFINITY;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        float x[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v = *reinterpret_cast<const float4*>(patch + lr * 32 + (((part * 4 + q) ^ rswz) << 2));
          x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the next block's stores stay behind these loads)
        const int cbase = n0 + j * 32 + part * 16;
        if (p.masks.any()) apply_logit_masks<16>(x, cbase, grow, p.M, p.masks);
        // the eos logit, if this lane's 16 columns hold it (selects, no branch)
        {
          const int e_eos = p.eos - cbase;
          float ev = x[0];
#pragma unroll
          for (int e = 1; e < 16; ++e) ev = (e_eos == e) ? x[e] : ev;
          const bool hit = (unsigned)e_eos < 16u;
          rec.eos_v = hit ? ev : rec.eos_v;
          rec.has_eos = rec.has_eos || hit;
        }
        // the list carries the element's position inside the half tile's 32 columns of this
        // lane (a constant per insertion); the column number is formed once, at the end
        float bm = x[0];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          topk_insert<TOPK>(rec.tx, rec.tc, x[e], jb * 16 + e);
          bm = fmaxf(bm, x[e]);
        }
        const float nm = fmaxf(m, bm);
        const float ms = (nm > -INFINITY) ? nm : 0.f;
        float add = 0.f;
#pragma unroll
        for (int e = 0; e < 16; e += 4)
          add += (fast_exp(x[e] - ms) + fast_exp(x[e + 1] - ms)) + (fast_exp(x[e + 2] - ms) + fast_exp(x[e + 3] - ms));
        sum = ((m > -INFINITY) ? sum * fast_exp(m - ms) : 0.f) + add;
        m = nm;
        __builtin_amdgcn_sched_barrier(0);  // one block at a time: the blocks' chains stay apart
      }
      rec.m = m;
      rec.sum = sum;
      {  // positions -> columns: position q = jb*16 + e is column n0 + (2*ht2 + jb)*32 + part*16 + e
        const int c0 = n0 + ht2 * 64 + part * 16;
#pragma unroll
        for (int k = 0; k < TOPK; ++k) {
          const int q = rec.tc[k];
          rec.tc[k] = q < 0 ? TOPK_EMPTY : c0 + (q & 15) + ((q >> 4) << 5);
        }
      }
      if (rec.has_eos && grow < p.M) p.eos_x[grow] = rec.eos_v;
      rec = topk_merge_neighbour<TOPK>(rec);
      if (part == 0 && grow < p.M)
        topk_store<TOPK>(p.rec + ((size_t)grow * (2 * p.tiles_n) + 2 * tile_n + ht2) * RL, rec);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <int TOPK>
int launch_resident(const VocabArgs& a, hipStream_t s) {
  const dim3 grid(a.tiles_n * a.groups), block(VB_NT);
  if (a.K == 512) hipLaunchKernelGGL((vocab_bf16_resident_kernel<TOPK, 2>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((vocab_bf16_resident_kernel<TOPK, 1>), grid, block, 0, s, a);
  ICAP_LAUNCH_CHECK("vocab_bf16_resident_kernel");
  return ICAP_OK;
}

}  // namespace

bool vocab_bf16_resident_supported(int M, int K, int N, int top_k) {
  return (top_k == 1 || top_k == 3 || top_k == 5) && M > 0 && N > 0 && (K == 256 || K == 512) &&
         (long long)M * K < (1LL << 31);
}

// x16: the activations as bf16 in the operand-block layout (dense_bf16_dual, blocked; its
// buffer covers ceil(M/32)*32 rows), packed_w [N_pad][K] bf16 (icap_pack_weight_bf16) -> the
// records of dense_topk_f32.  compute_units sizes the grid (one workgroup per compute unit).
int vocab_bf16_resident(const void* x16, const void* packed_w, const float* bias, int M, int K, int N,
                        int eos, int top_k, float* records, float* eos_logit, const LogitMasks* masks,
                        int compute_units, hipStream_t s) {
  ICAP_REQUIRE(x16 && packed_w && records && eos_logit, "vocab_bf16_resident: null pointer");
  ICAP_REQUIRE(vocab_bf16_resident_supported(M, K, N, top_k),
               "vocab_bf16_resident: shape not served (M=%d K=%d N=%d top_k=%d)", M, K, N, top_k);
  ICAP_REQUIRE(aligned16(x16) && aligned16(packed_w) && aligned16(records),
               "vocab_bf16_resident: unaligned buffer");
  VocabArgs a;
  a.a16 = static_cast<const uint16_t*>(x16);
  a.w = static_cast<const uint16_t*>(packed_w);
  a.bias = bias; a.rec = records; a.eos_x = eos_logit;
  a.M = M; a.N = N; a.K = K; a.eos = eos;
  a.tiles_n = ceil_div(N, VB_BN);
  // as many workgroups per weight panel as fill the chip, but no more than there are groups
  // of eight row blocks
  const int cus = compute_units > 0 ? compute_units : 256;
  const int max_groups = ceil_div(ceil_div(M, 32), VB_WAVES);
  int groups = cus / a.tiles_n;
  if (groups < 1) groups = 1;
  if (groups > max_groups) groups = max_groups;
  a.groups = groups;
  if (masks) a.masks = *masks;
  if (top_k == 1) return launch_resident<1>(a, s);
  if (top_k == 3) return launch_resident<3>(a, s);
  return launch_resident<5>(a, s);
}

}  // namespace icap
