// Bare matrix-pipe rate: every wave issues v_mfma_f32_32x32x2_f32 (or the bf16 32x32x16) on
// operands held in registers, `chains` independent accumulators per wave, `waves` waves per
// SIMD, one workgroup per compute unit.  Prints TFLOP/s against the 157.3 / 2500 nominal peaks:
// what a kernel can reach at most once operand delivery costs nothing.
//   hipcc -O3 --offload-arch=gfx950 tools/lab/mfma_peak_lab.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int CHAINS, bool BF16>
__global__ void mfma_loop(float* out, int iters, float seed) {
  f32x16 acc[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float a = seed + threadIdx.x, b = seed * 0.5f + threadIdx.x;
  bf16x8 a8, b8;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(a + i); b8[i] = (__bf16)(b - i); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        if constexpr (BF16) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[c], 0, 0, 0);
        else acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
      }
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  if (s == 12345.678f) out[0] = s;  // keeps the loop alive
}

template <int CHAINS, bool BF16>
void run(int waves_per_simd, int cus, float* d) {
  const int threads = 64 * 4 * waves_per_simd, iters = 2000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((mfma_loop<CHAINS, BF16>), dim3(cus), dim3(threads), 0, 0, d, iters, 1.f);
  (void)hipEventRecord(e0, 0);
  const int reps = 5;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((mfma_loop<CHAINS, BF16>), dim3(cus), dim3(threads), 0, 0, d, iters, 1.f);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops_per_mfma = BF16 ? 2.0 * 32 * 32 * 16 : 2.0 * 32 * 32 * 2;
  const double flops = flops_per_mfma * 8 * CHAINS * (double)iters * (threads / 64) * cus * reps;
  const double tf = flops / (ms * 1e-3) / 1e12;
  printf("%s chains %d waves/SIMD %d: %.1f us per launch, %.1f TFLOP/s = %.2f of %s\n", BF16 ? "bf16 32x32x16" : "f32 32x32x2",
         CHAINS, waves_per_simd, ms * 1e3 / reps, tf, tf / (BF16 ? 2500.0 : 157.3), BF16 ? "2500" : "157.3");
}

int main() {
  float* d;
  (void)hipMalloc(&d, 64);
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  printf("%d compute units, clock %d MHz\n", cus, prop.clockRate / 1000);
  for (int w = 1; w <= 4; ++w) { run<1, false>(w, cus, d); run<2, false>(w, cus, d); }
  run<4, false>(1, cus, d);
  for (int w = 1; w <= 2; ++w) { run<1, true>(w, cus, d); run<2, true>(w, cus, d); run<4, true>(w, cus, d); }
  (void)hipFree(d);
  return 0;
}
