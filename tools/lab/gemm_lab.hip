// Ablation lab for the f32 MFMA GEMM main loop (not part of the library).
// Build: hipcc -O3 --offload-arch=gfx950 -I../../include -I../../image_caption_amd/csrc gemm_lab.hip -o gemm_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
struct Args { const float* A; const float* B; float* C; int M, N, K, lda, ldb, ldc, tiles_m, tiles_n; };

__device__ __forceinline__ void tile_of_block(int pid, int tiles_m, int tiles_n, int& tm, int& tn) {
  const int total = tiles_m * tiles_n; const int q = total >> 3, r = total & 7; const int xcd = pid & 7;
  const int sid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (pid >> 3);
  tm = sid % tiles_m; tn = sid / tiles_m;
}
// VARIANT bits: 1 = skip global loads in loop, 2 = skip MFMA, 4 = skip LDS store+barrier in loop, 8 = skip epilogue stores,
// 32 = (with 16) the B tile goes from global memory straight into LDS (global_load_lds_dwordx4), only A is register-staged,
// 16 = the library's schedule: next-step loads above the per-step barrier, staging registers pinned below the multiplies
template <int BM, int BN, int WM, int WN, int VARIANT, int MINW, int BK = 32>
__global__ __launch_bounds__(WM* WN * 64, MINW)
#ifdef WPE
    __attribute__((amdgpu_waves_per_eu(WM * WN * MINW / 4, WM * WN * MINW / 4)))
#endif
    void k(Args p) {
  constexpr int NT = WM * WN * 64, TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int A_LD = (BM * BK / 4) / NT, B_LD = (BK * BN / 4) / NT;
  __shared__ float4 As[2][BK / 4][BM + 1];
  __shared__ float Bs[2][BK][BN];
  int tile_m, tile_n; tile_of_block(blockIdx.x, p.tiles_m, p.tiles_n, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WN, wn = wave % WN, l31 = lane & 31, half = lane >> 5;
  int off_a[A_LD], off_b[B_LD];
#pragma unroll
  for (int i = 0; i < A_LD; ++i) { const int f = tid + i * NT; const int gm = min(m0 + f / (BK / 4), p.M - 1); off_a[i] = gm * p.lda + 4 * (f % (BK / 4)); }
#pragma unroll
  for (int i = 0; i < B_LD; ++i) { const int f = tid + i * NT; const int col = min(n0 + 4 * (f % (BN / 4)), p.N - 4); off_b[i] = (f / (BN / 4)) * p.ldb + col; }
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float4 ra[A_LD], rb[B_LD];
#define LOADT(a_, b_) _Pragma("unroll") for (int i = 0; i < A_LD; ++i) ra[i] = *reinterpret_cast<const float4*>((a_) + off_a[i]); \
  _Pragma("unroll") for (int i = 0; i < B_LD; ++i) rb[i] = *reinterpret_cast<const float4*>((b_) + off_b[i]);
#define STORET(buf_) _Pragma("unroll") for (int i = 0; i < A_LD; ++i) { const int f = tid + i * NT; As[buf_][f % (BK / 4)][f / (BK / 4)] = ra[i]; } \
  _Pragma("unroll") for (int i = 0; i < B_LD; ++i) { const int f = tid + i * NT; *reinterpret_cast<float4*>(&Bs[buf_][f / (BN / 4)][4 * (f % (BN / 4))]) = rb[i]; }
#define LOADA(a_) _Pragma("unroll") for (int i = 0; i < A_LD; ++i) ra[i] = *reinterpret_cast<const float4*>((a_) + off_a[i]);
#define STOREA(buf_) _Pragma("unroll") for (int i = 0; i < A_LD; ++i) { const int f = tid + i * NT; As[buf_][f % (BK / 4)][f / (BK / 4)] = ra[i]; }
#if defined(__HIP_DEVICE_COMPILE__)
#define GLDS(src_, dst_) __builtin_amdgcn_global_load_lds((src_), (dst_), 16, 0, 0)
#define WAIT_VM0() __builtin_amdgcn_s_waitcnt(0x0F70)
#else
#define GLDS(src_, dst_) ((void)0)
#define WAIT_VM0() ((void)0)
#endif
#define DIRECTB(b_, buf_) _Pragma("unroll") for (int i = 0; i < B_LD; ++i) { \
    const int f0 = (tid & ~63) + i * NT; /* first float4 of this wave's run: 64 consecutive float4 = 2 rows */ \
    GLDS((b_) + off_b[i], &Bs[buf_][f0 / (BN / 4)][4 * (f0 % (BN / 4))]); }
  const int nk = p.K / BK;
  const float* a_tile = p.A; const float* b_tile = p.B;
  if (VARIANT & 32) {
    LOADA(a_tile) DIRECTB(b_tile, 0) STOREA(0)
    WAIT_VM0();
    for (int t = 0; t < nk; ++t) {
      const int buf = t & 1;
      const bool more = t + 1 < nk; a_tile += more ? BK : 0; b_tile += more ? (size_t)BK * p.ldb : 0;
      LOADA(a_tile)
      __syncthreads();
      DIRECTB(b_tile, buf ^ 1)
#pragma unroll
      for (int g = 0; g < BK / 8; ++g) {
        float4 a[TM]; float b[TN][4];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = As[buf][2 * g + half][(wm * TM + i) * 32 + l31];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) b[j][s2] = Bs[buf][8 * g + 4 * half + s2][(wn * TN + j) * 32 + l31];
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
#pragma unroll
      for (int i = 0; i < A_LD; ++i) asm volatile("" : "+v"(ra[i].x), "+v"(ra[i].y), "+v"(ra[i].z), "+v"(ra[i].w));
      STOREA(buf ^ 1)
      WAIT_VM0();
    }
  } else {
  LOADT(a_tile, b_tile) STORET(0) if (VARIANT & 4) { STORET(1) }
  if (!(VARIANT & 16)) __syncthreads();
  for (int t = 0; t < nk; ++t) {
    const int buf = t & 1;
    if (!(VARIANT & 1)) { const bool more = t + 1 < nk; a_tile += more ? BK : 0; b_tile += more ? (size_t)BK * p.ldb : 0; LOADT(a_tile, b_tile) }
    if (VARIANT & 16) __syncthreads();
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      float4 a[TM]; float b[TN][4];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[buf][2 * g + half][(wm * TM + i) * 32 + l31];
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int s = 0; s < 4; ++s) b[j][s] = Bs[buf][8 * g + 4 * half + s][(wn * TN + j) * 32 + l31];
      if (!(VARIANT & 2)) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j][1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j][2], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j][3], acc[i][j], 0, 0, 0);
          }
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) { acc[i][j][0] += a[i].x * b[j][0] + a[i].y * b[j][1] + a[i].z * b[j][2] + a[i].w * b[j][3]; }
      }
    }
    if (VARIANT & 16) {
#pragma unroll
      for (int i = 0; i < A_LD; ++i) asm volatile("" : "+v"(ra[i].x), "+v"(ra[i].y), "+v"(ra[i].z), "+v"(ra[i].w));
#pragma unroll
      for (int i = 0; i < B_LD; ++i) asm volatile("" : "+v"(rb[i].x), "+v"(rb[i].y), "+v"(rb[i].z), "+v"(rb[i].w));
      STORET(buf ^ 1)
    } else if (!(VARIANT & 4)) { STORET(buf ^ 1) __syncthreads(); }
  }
  }
  if (VARIANT & 8) {
    float sacc = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
    if (sacc == 123.456f) p.C[tid] = sacc + ra[0].x + rb[0].x;
    return;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + (wn * TN + j) * 32 + l31;
    if (col >= p.N) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (row < p.M) p.C[(size_t)row * p.ldc + col] = acc[i][j][r] + ((VARIANT & 1) ? ra[0].x * 0.f : 0.f);
      }
  }
}

template <int BM, int BN, int WM, int WN, int VARIANT, int MINW, int BK = 32>
float run(Args a, int iters, const char* name) {
  a.tiles_m = (a.M + BM - 1) / BM; a.tiles_n = (a.N + BN - 1) / BN;
  const int grid = a.tiles_m * a.tiles_n;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 80; ++i) hipLaunchKernelGGL((k<BM, BN, WM, WN, VARIANT, MINW, BK>), dim3(grid), dim3(WM * WN * 64), 0, 0, a);
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k<BM, BN, WM, WN, VARIANT, MINW, BK>), dim3(grid), dim3(WM * WN * 64), 0, 0, a);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const float us = ms * 1e3f / iters;
  printf("%-44s grid %5d  %8.1f us  %6.1f TF\n", name, grid, us, 2.0 * a.M * a.N * a.K / us / 1e6);
  return us;
}

int main(int argc, char** argv) {
  int M = argc > 1 ? atoi(argv[1]) : 1536, N = argc > 2 ? atoi(argv[2]) : 10000, K = argc > 3 ? atoi(argv[3]) : 512;
  std::vector<float> hA((size_t)M * K), hB((size_t)K * N);
  srand(1);
  for (auto& v : hA) v = (rand() / (float)RAND_MAX) * 2 - 1;
  for (auto& v : hB) v = (rand() / (float)RAND_MAX) * 2 - 1;
  float *A, *B, *C;
  hipMalloc(&A, hA.size() * 4); hipMalloc(&B, hB.size() * 4); hipMalloc(&C, (size_t)M * N * 4);
  hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice); hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
  Args a{A, B, C, M, N, K, K, N, N, 0, 0};
  printf("M=%d N=%d K=%d\n", M, N, K);
  run<128, 128, 2, 2, 0, 1>(a, 20, "128x128 full");
  // correctness spot check
  std::vector<float> hC((size_t)M * N); hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
  double maxerr = 0; for (int t = 0; t < 200; ++t) { int r = rand() % M, c = rand() % N; double s = 0; for (int kk = 0; kk < K; ++kk) s += (double)hA[(size_t)r * K + kk] * hB[(size_t)kk * N + c]; maxerr = fmax(maxerr, fabs(s - hC[(size_t)r * N + c])); }
  printf("spot-check max abs err %.3e\n", maxerr);
#define R(BM,BN,WM,WN,MW) run<BM,BN,WM,WN,0,MW>(a, 20, #BM "x" #BN " waves " #WM "x" #WN " minw " #MW);
  R(128,128,4,2,2) R(128,128,2,4,2) R(128,128,4,4,2) R(128,128,4,4,4)
  R(256,128,4,4,1) R(256,128,4,4,2) R(128,256,4,4,2) R(256,256,4,4,1) R(256,256,4,4,2)
  R(64,128,2,4,2) R(128,64,4,2,2) R(64,64,2,2,4)
  R(128,128,4,2,3) R(128,128,4,2,4)
#define S(BM,BN,WM,WN,MW) run<BM,BN,WM,WN,16,MW>(a, 20, "sched " #BM "x" #BN " waves " #WM "x" #WN " minw " #MW);
  S(128,128,4,2,2) S(128,128,2,4,2) S(128,128,4,4,2) S(256,128,4,4,1) S(128,256,4,4,1) S(128,256,2,4,1) S(256,256,4,4,1)
  S(64,128,2,4,4) S(128,64,4,2,4) S(64,64,2,2,4) S(128,128,4,2,3) S(256,128,4,2,1) S(128,256,4,2,1)
#define SK(BM,BN,WM,WN,MW,BKK) run<BM,BN,WM,WN,16,MW,BKK>(a, 20, "sched " #BM "x" #BN " waves " #WM "x" #WN " minw " #MW " BK " #BKK);
  SK(128,128,4,2,2,32) SK(128,128,4,2,3,16) SK(128,128,4,2,4,16) SK(128,128,4,2,2,16) SK(128,128,4,2,1,64) SK(128,128,2,2,4,16) SK(128,128,2,2,6,16) SK(128,128,2,2,3,32)
  SK(128,128,4,2,2,32)
  {
    run<128,128,4,2,48,2>(a, 20, "direct-B sched 128x128 4x2");
    std::vector<float> hC2((size_t)M * N); hipMemcpy(hC2.data(), C, hC2.size() * 4, hipMemcpyDeviceToHost);
    double me = 0; for (int t = 0; t < 400; ++t) { int r = rand() % M, c = rand() % N; double s2 = 0; for (int kk = 0; kk < K; ++kk) s2 += (double)hA[(size_t)r * K + kk] * hB[(size_t)kk * N + c]; me = fmax(me, fabs(s2 - hC2[(size_t)r * N + c])); }
    printf("direct-B spot-check max abs err %.3e\n", me);
    run<128,128,4,2,16,2>(a, 20, "sched 128x128 4x2 (reference)");
    run<128,128,4,2,48,2>(a, 20, "direct-B sched 128x128 4x2");
    run<128,128,4,4,48,2>(a, 20, "direct-B sched 128x128 4x4");
    run<128,128,2,4,48,2>(a, 20, "direct-B sched 128x128 2x4");
  }
  run<128,128,4,2,8,2>(a, 20, "128x128 4x2 no epilogue");
  run<128,128,4,2,13,2>(a, 20, "128x128 4x2 MFMA+LDS reads only");
  run<128,128,4,2,4,2>(a, 20, "128x128 4x2 no LDS store/barrier");
  return 0;
}
