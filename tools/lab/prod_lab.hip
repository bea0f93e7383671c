// Runs the library's own GEMM kernel templates under the lab's back-to-back harness (the
// library source is included into this translation unit).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../include -I../../image_caption_amd/csrc prod_lab.hip -o prod_lab
#include "../../image_caption_amd/csrc/gemm_f32.hip"
#include <vector>
namespace icap {
std::string& last_error() { static std::string m; return m; }
int fail(int code, const char* fmt, ...) { (void)fmt; return code; }
}
using namespace icap;

template <typename F>
static void timeit(const char* name, F f, double flops) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 150; ++i) f();
  hipEventRecord(e0, 0);
  for (int i = 0; i < 20; ++i) f();
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  printf("%-40s %8.1f us  %6.1f TF\n", name, ms * 1e3 / 20, flops / (ms * 1e3 / 20) / 1e6);
}

int main() {
  const int M = 1536, N = 10000, K = 512;
  std::vector<float> hA((size_t)M * K), hB((size_t)K * N);
  srand(1);
  for (auto& v : hA) v = (rand() / (float)RAND_MAX) * 2 - 1;
  for (auto& v : hB) v = (rand() / (float)RAND_MAX) * 2 - 1;
  float *A, *B, *C, *bias;
  hipMalloc(&A, hA.size() * 4); hipMalloc(&B, hB.size() * 4); hipMalloc(&C, (size_t)M * N * 4); hipMalloc(&bias, N * 4);
  hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
  hipMemset(bias, 0, N * 4);
  const double fl = 2.0 * M * N * K;
  for (int order = 0; order < 2; ++order)
    for (int wb = 0; wb < 2; ++wb) {
      DenseArgs a{A, B, wb ? bias : nullptr, C, M, N, K, K, N, N, ICAP_ACT_NONE, 0, 0, 0, nullptr, nullptr, -1};
      a.tiles_m = ceil_div(M, 128); a.tiles_n = ceil_div(N, 128); a.band_order = order;
      char name[64]; snprintf(name, sizeof name, "prod fast kernel order=%d bias=%d", order, wb);
      timeit(name, [&] { hipLaunchKernelGGL((dense_f32_fast_kernel<128, 128, 4, 2, 2, 0>), dim3(a.tiles_m * a.tiles_n), dim3(512), 0, 0, a); }, fl);
    }
  timeit("dense_f32()", [&] { dense_f32(A, B, bias, C, M, K, N, 0, 0); }, fl);
  float *rec, *eosx;
  hipMalloc(&rec, (size_t)M * dense_topk_half_tiles(N) * 8 * 4); hipMalloc(&eosx, M * 4);
  timeit("dense_topk_f32()", [&] { dense_topk_f32(A, B, bias, M, K, N, 4, 3, rec, eosx, nullptr, 0); }, fl);
  timeit("dense_f32() again", [&] { dense_f32(A, B, bias, C, M, K, N, 0, 0); }, fl);
  return 0;
}
