// What does one dependent dispatch cost?  N trivial kernels back to back in one stream, (a) launched one by one,
// (b) captured once in a hipGraph and replayed.  (The large-D chain is ~31 dependent launches of 6-17 us each.)
//   hipcc --offload-arch=gfx950 -O3 launch_probe.hip -o launch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void tiny(float* p, int i) { if (threadIdx.x == 0 && blockIdx.x == 0) p[i & 63] += 1.0f; }
__global__ void wide(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.0f; }
int main() {
  float* d; CK(hipMalloc((void**)&d, 64 << 20)); CK(hipMemset(d, 0, 64 << 20));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int N = 200;
  for (int mode = 0; mode < 2; ++mode) {  // 0: one workgroup, 1: 512 workgroups x 256 threads touching 512 KB
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < N; ++i) { if (mode == 0) tiny<<<1, 64, 0, st>>>(d, i); else wide<<<512, 256, 0, st>>>(d, 512 * 256); }
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 2) printf("%s kernels, stream launches : %.2f us per dependent launch\n", mode ? "512-workgroup" : "1-workgroup  ", ms * 1e3 / N);
    }
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < N; ++i) { if (mode == 0) tiny<<<1, 64, 0, st>>>(d, i); else wide<<<512, 256, 0, st>>>(d, 512 * 256); }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, st));
      CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 2) printf("%s kernels, graph replay    : %.2f us per dependent launch\n", mode ? "512-workgroup" : "1-workgroup  ", ms * 1e3 / N);
    }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
