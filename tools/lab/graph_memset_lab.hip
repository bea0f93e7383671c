// Does a memset node of a captured, single-stream (linear) graph run on EVERY replay, in order
// with the kernels around it?  Graph: memset(word, 0, bytes) -> bump kernel (word += 1, out = word).
// Expected out == 1 after every replay.
//   graph_memset_lab <filler kernels between> <replay on null stream 0|1> [bytes of the memset, default 8]
// Build: hipcc -O2 --offload-arch=gfx950 tools/lab/graph_memset_lab.hip -o tools/lab/graph_memset_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
__global__ void bump(unsigned* word, unsigned* out) { *out = atomicAdd(word, 1u) + 1u; }
__global__ void filler(float* p, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
int main(int argc, char** argv) {
  const int nodes_between = argc > 1 ? atoi(argv[1]) : 0;  // filler kernels between memset and bump
  const bool on_null = argc > 2 && atoi(argv[2]) != 0;      // replay on the null stream
  const int bytes = argc > 3 ? atoi(argv[3]) : 8;           // (18: a size that is no multiple of 4)
  unsigned *word, *out; float* buf;
  CK(hipMalloc(&word, 256)); CK(hipMalloc(&out, 256)); CK(hipMalloc(&buf, 1 << 20));
  CK(hipMemset(word, 0, 256)); CK(hipMemset(out, 0, 256)); CK(hipMemset(buf, 0, 1 << 20));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  CK(hipMemsetAsync(word, 0, bytes, s));
  for (int i = 0; i < nodes_between; ++i) hipLaunchKernelGGL(filler, dim3(1024), dim3(256), 0, s, buf, 1 << 18);
  hipLaunchKernelGGL(bump, dim3(1), dim3(1), 0, s, word, out);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  int bad = 0;
  for (int r = 0; r < 6; ++r) {
    hipStream_t ls = on_null ? nullptr : s;
    CK(hipGraphLaunch(ge, ls));
    CK(hipStreamSynchronize(ls));
    unsigned h = 0; CK(hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost));
    printf("replay %d: out = %u (%s)\n", r, h, h == 1 ? "memset ran" : "memset did NOT run before the kernel");
    bad += h != 1;
  }
  printf("graph_memset_lab: fillers %d, %d bytes, %s stream: %s\n", nodes_between, bytes, on_null ? "null" : "own", bad ? "STALE" : "ok");
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return 0;
}
