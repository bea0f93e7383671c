// Measures the bubble that stream-ordering primitives put between dependent kernels on
// gfx950: plain back-to-back launches, hipEventRecord (timing / no timing) between
// launches, events attached to the dispatch itself (hipExtLaunchKernelGGL), a two-stream
// fork/join per step, and the same sequence replayed from a captured hipGraph.
// Build: hipcc -O2 --offload-arch=gfx950 gap_lab.hip -o gap_lab
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x)                                                              \
  do {                                                                     \
    hipError_t e_ = (x);                                                   \
    if (e_ != hipSuccess) {                                                \
      printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
      return 1;                                                            \
    }                                                                      \
  } while (0)

// Spins for about `cycles` shader clocks; every wave exits.
__global__ void spin_kernel(float* out, long long cycles) {
  const long long t0 = wall_clock64();
  float v = threadIdx.x;
  while (wall_clock64() - t0 < cycles) v = v * 1.0001f + 1.0f;
  if (v == 12345.678f) out[0] = v;
}

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch())
      .count();
}

int main() {
  float* buf;
  CK(hipMalloc(&buf, 4096));
  hipStream_t s, side;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
  const int steps = 200, per_step = 5;
  const long long cyc = 2000;  // 20 us at the 100 MHz wall clock
  const dim3 grid(256), block(256);
  std::vector<hipEvent_t> tim(2 * steps * per_step), fast(2 * steps * per_step);
  for (auto& e : tim) CK(hipEventCreate(&e));
  for (auto& e : fast) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));

  auto run = [&](const char* name, auto&& body) -> int {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipDeviceSynchronize());
      const double t0 = now_us();
      if (body() != 0) return 1;
      CK(hipDeviceSynchronize());
      const double t1 = now_us();
      if (rep == 2) printf("%-46s %8.1f us total  %7.2f us per kernel\n", name, t1 - t0, (t1 - t0) / (steps * per_step));
    }
    return 0;
  };

  if (run("plain back-to-back", [&]() -> int {
        for (int i = 0; i < steps * per_step; ++i) spin_kernel<<<grid, block, 0, s>>>(buf, cyc);
        return 0;
      }))
    return 1;
  if (run("hipEventRecord(timing) after every kernel", [&]() -> int {
        for (int i = 0; i < steps * per_step; ++i) {
          spin_kernel<<<grid, block, 0, s>>>(buf, cyc);
          CK(hipEventRecord(tim[i], s));
        }
        return 0;
      }))
    return 1;
  if (run("hipEventRecord(no timing) after every kernel", [&]() -> int {
        for (int i = 0; i < steps * per_step; ++i) {
          spin_kernel<<<grid, block, 0, s>>>(buf, cyc);
          CK(hipEventRecord(fast[i], s));
        }
        return 0;
      }))
    return 1;
  if (run("timing pair around 1 of 5 kernels", [&]() -> int {
        for (int i = 0; i < steps * per_step; ++i) {
          if (i % per_step == 0) CK(hipEventRecord(tim[2 * i], s));
          spin_kernel<<<grid, block, 0, s>>>(buf, cyc);
          if (i % per_step == 0) CK(hipEventRecord(tim[2 * i + 1], s));
        }
        return 0;
      }))
    return 1;
  if (run("ext launch start/stop on 1 of 5 kernels", [&]() -> int {
        for (int i = 0; i < steps * per_step; ++i) {
          if (i % per_step == 0)
            hipExtLaunchKernelGGL(spin_kernel, grid, block, 0, s, tim[2 * i], tim[2 * i + 1], 0, buf, cyc);
          else
            spin_kernel<<<grid, block, 0, s>>>(buf, cyc);
        }
        return 0;
      }))
    return 1;
  {
    float ms = 0;
    CK(hipEventElapsedTime(&ms, tim[0], tim[1]));
    printf("   ext-launch event pair says kernel = %.2f us\n", ms * 1e3);
  }
  // Per step: k0 on s; fork: side runs 2 kernels, s runs 3 kernels; join before next k0.
  if (run("fork/join with hipEventRecord (7 kernels/step)", [&]() -> int {
        for (int i = 0; i < steps; ++i) {
          if (i) CK(hipStreamWaitEvent(s, fast[2 * i - 1], 0));
          spin_kernel<<<grid, block, 0, s>>>(buf, cyc);
          CK(hipEventRecord(fast[2 * i], s));
          CK(hipStreamWaitEvent(side, fast[2 * i], 0));
          spin_kernel<<<grid, block, 0, side>>>(buf, cyc);
          spin_kernel<<<grid, block, 0, side>>>(buf, cyc);
          CK(hipEventRecord(fast[2 * i + 1], side));
          for (int k = 0; k < 4; ++k) spin_kernel<<<grid, block, 0, s>>>(buf, cyc);
        }
        return 0;
      }))
    return 1;
  if (run("fork/join with ext-launch stop events", [&]() -> int {
        for (int i = 0; i < steps; ++i) {
          if (i) CK(hipStreamWaitEvent(s, fast[2 * i - 1], 0));
          hipExtLaunchKernelGGL(spin_kernel, grid, block, 0, s, nullptr, fast[2 * i], 0, buf, cyc);
          CK(hipStreamWaitEvent(side, fast[2 * i], 0));
          spin_kernel<<<grid, block, 0, side>>>(buf, cyc);
          hipExtLaunchKernelGGL(spin_kernel, grid, block, 0, side, nullptr, fast[2 * i + 1], 0, buf, cyc);
          for (int k = 0; k < 4; ++k) spin_kernel<<<grid, block, 0, s>>>(buf, cyc);
        }
        return 0;
      }))
    return 1;
  printf("   (fork/join critical path = 5 kernels per step; per-kernel column divides by 5)\n");

  // Graph replay of the fork/join sequence.
  hipGraph_t graph;
  hipGraphExec_t exec;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
  for (int i = 0; i < steps; ++i) {
    if (i) CK(hipStreamWaitEvent(s, fast[2 * i - 1], 0));
    spin_kernel<<<grid, block, 0, s>>>(buf, cyc);
    CK(hipEventRecord(fast[2 * i], s));
    CK(hipStreamWaitEvent(side, fast[2 * i], 0));
    spin_kernel<<<grid, block, 0, side>>>(buf, cyc);
    spin_kernel<<<grid, block, 0, side>>>(buf, cyc);
    CK(hipEventRecord(fast[2 * i + 1], side));
    for (int k = 0; k < 4; ++k) spin_kernel<<<grid, block, 0, s>>>(buf, cyc);
  }
  CK(hipStreamWaitEvent(s, fast[2 * steps - 1], 0));
  CK(hipStreamEndCapture(s, &graph));
  CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  if (run("fork/join replayed from a hipGraph", [&]() -> int {
        CK(hipGraphLaunch(exec, s));
        return 0;
      }))
    return 1;
  // Linear graph (no fork): 5 kernels per step.
  hipGraph_t g2;
  hipGraphExec_t e2;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
  for (int i = 0; i < steps * per_step; ++i) spin_kernel<<<grid, block, 0, s>>>(buf, cyc);
  CK(hipStreamEndCapture(s, &g2));
  CK(hipGraphInstantiate(&e2, g2, nullptr, nullptr, 0));
  if (run("linear chain replayed from a hipGraph", [&]() -> int {
        CK(hipGraphLaunch(e2, s));
        return 0;
      }))
    return 1;
  CK(hipGraphExecDestroy(exec));
  CK(hipGraphDestroy(graph));
  CK(hipGraphExecDestroy(e2));
  CK(hipGraphDestroy(g2));
  printf("done\n");
  return 0;
}
