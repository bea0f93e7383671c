// Times the library's own entry points back to back from C++ (no Python in the loop).
// Build: hipcc -O3 --offload-arch=gfx950 -I../../include abi_lab.hip -L../../image_caption_amd -licap -Wl,-rpath,'$ORIGIN/../../image_caption_amd' -o abi_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "icap.h"

int main() {
  const int M = 1536, N = 10000, K = 512;
  std::vector<float> hA((size_t)M * K), hB((size_t)K * N), hb(N, 0.f);
  srand(1);
  for (auto& v : hA) v = (rand() / (float)RAND_MAX) * 2 - 1;
  for (auto& v : hB) v = (rand() / (float)RAND_MAX) * 2 - 1;
  float *A, *B, *C, *bias, *rec, *eosx;
  hipMalloc(&A, hA.size() * 4); hipMalloc(&B, hB.size() * 4); hipMalloc(&C, (size_t)M * N * 4);
  hipMalloc(&bias, N * 4); hipMalloc(&rec, (size_t)M * icap_dense_topk_half_tiles(N) * 8 * 4); hipMalloc(&eosx, M * 4);
  hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      for (int i = 0; i < 3; ++i) mode ? icap_dense_topk_f32(A, B, bias, M, K, N, 4, 3, rec, eosx, 0) : icap_dense_f32(A, B, bias, C, M, K, N, 0, 0);
      hipEventRecord(e0, 0);
      for (int i = 0; i < 20; ++i) mode ? icap_dense_topk_f32(A, B, bias, M, K, N, 4, 3, rec, eosx, 0) : icap_dense_f32(A, B, bias, C, M, K, N, 0, 0);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      printf("%s: %.1f us  %.1f TF\n", mode ? "icap_dense_topk_f32" : "icap_dense_f32", ms * 1e3 / 20, 2.0 * M * N * K / (ms * 1e3 / 20) / 1e6);
    }
  }
  return 0;
}
