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
  {
    // output projections of the transformer path: tall-skinny products
    const int M2 = 3072, N2 = 512;
    for (int K2 : {4096, 2048}) {
      float *A2, *B2, *C2;
      hipMalloc(&A2, (size_t)M2 * K2 * 4); hipMalloc(&B2, (size_t)K2 * N2 * 4); hipMalloc(&C2, (size_t)M2 * N2 * 4);
      hipMemset(A2, 0, (size_t)M2 * K2 * 4); hipMemset(B2, 0, (size_t)K2 * N2 * 4);
      const double f2 = 2.0 * M2 * N2 * K2;
      printf("M=%d N=%d K=%d\n", M2, N2, K2);
      timeit("dense_f32() dispatch", [&] { dense_f32(A2, B2, bias, C2, M2, K2, N2, 0, 0); }, f2);
      DenseArgs a{A2, B2, bias, C2, M2, N2, K2, K2, N2, N2, ICAP_ACT_NONE, 0, 0, 0, nullptr, nullptr, -1};
      auto fast = [&](auto kern, int bm, int bn, int nt, const char* nm) {
        DenseArgs b = a; b.tiles_m = ceil_div(M2, bm); b.tiles_n = ceil_div(N2, bn); b.band_order = 0;
        timeit(nm, [&] { hipLaunchKernelGGL(kern, dim3(b.tiles_m * b.tiles_n), dim3(nt), 0, 0, b); }, f2);
      };
      fast(dense_f32_fast_kernel<128, 128, 4, 2, 2, 0>, 128, 128, 512, "fast 128x128 (96 tiles)");
      fast(dense_f32_fast_kernel<64, 64, 2, 2, 2, 0>, 64, 64, 256, "fast 64x64 minw2 (384 tiles)");
      fast(dense_f32_fast_kernel<64, 64, 2, 2, 4, 0>, 64, 64, 256, "fast 64x64 minw4 (384 tiles)");
      fast(dense_f32_fast_kernel<128, 64, 4, 2, 2, 0>, 128, 64, 512, "fast 128x64 8 waves (192 tiles)");
      fast(dense_f32_fast_kernel<64, 128, 2, 4, 2, 0>, 64, 128, 512, "fast 64x128 8 waves (192 tiles)");
      fast(dense_f32_fast_kernel<128, 64, 4, 1, 2, 0>, 128, 64, 256, "fast 128x64 4 waves (192 tiles)");
      fast(dense_f32_fast_kernel<64, 64, 2, 1, 4, 0>, 64, 64, 128, "fast 64x64 2 waves (384 tiles)");
    }
  }
  {
    const int M3 = 1536, N3 = 512, K3 = 512;
    float *A3, *B3, *C3;
    hipMalloc(&A3, (size_t)M3 * K3 * 4); hipMalloc(&B3, (size_t)K3 * N3 * 4); hipMalloc(&C3, (size_t)M3 * N3 * 4);
    hipMemset(A3, 0, (size_t)M3 * K3 * 4); hipMemset(B3, 0, (size_t)K3 * N3 * 4);
    const double f3 = 2.0 * M3 * N3 * K3;
    printf("M=%d N=%d K=%d\n", M3, N3, K3);
    timeit("dense_f32() dispatch (shortk)", [&] { dense_f32(A3, B3, bias, C3, M3, K3, N3, 0, 0); }, f3);
    DenseArgs a{A3, B3, bias, C3, M3, N3, K3, K3, N3, N3, ICAP_ACT_NONE, 0, 0, 0, nullptr, nullptr, -1};
    auto fast = [&](auto kern, int bm, int bn, int nt, const char* nm) {
      DenseArgs b = a; b.tiles_m = ceil_div(M3, bm); b.tiles_n = ceil_div(N3, bn); b.band_order = 0;
      timeit(nm, [&] { hipLaunchKernelGGL(kern, dim3(b.tiles_m * b.tiles_n), dim3(nt), 0, 0, b); }, f3);
    };
    fast(dense_f32_fast_kernel<64, 64, 2, 2, 2, 0>, 64, 64, 256, "fast 64x64 4 waves (192 tiles)");
    fast(dense_f32_fast_kernel<64, 128, 2, 4, 2, 0>, 64, 128, 512, "fast 64x128 8 waves (96 tiles)");
    fast(dense_f32_fast_kernel<128, 64, 4, 2, 2, 0>, 128, 64, 512, "fast 128x64 8 waves (96 tiles)");
    fast(dense_f32_fast_kernel<32, 64, 1, 2, 4, 0>, 32, 64, 128, "fast 32x64 2 waves (384 tiles)");
    fast(dense_f32_fast_kernel<64, 32, 2, 1, 4, 0>, 64, 32, 128, "fast 64x32 2 waves (384 tiles)");
    fast(dense_f32_fast_kernel<32, 32, 1, 1, 8, 0>, 32, 32, 64, "fast 32x32 1 wave (768 tiles)");
  }
  return 0;
}
