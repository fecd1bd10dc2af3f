// What does a software grid barrier cost next to a dependent dispatch (tools/launch_probe: 2.7-3.0 us in a stream)?
// G co-resident workgroups run NB rounds of: write 16 KB each, release, arrive on a counter, spin, acquire, read a
// neighbour's 16 KB.  Spins are bounded: a scheduling surprise ends the run with an error instead of hanging the GPU.
//   hipcc --offload-arch=gfx950 -O3 gridbar_probe.hip -o gridbar_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void k(float* data, unsigned* counter, unsigned base, int nb, int payload, int* err, float* sink) {
  const int G = gridDim.x, w = blockIdx.x, tid = threadIdx.x;
  float acc = 0.f;
  for (int b = 0; b < nb; ++b) {
    if (payload) {
      float4* mine = reinterpret_cast<float4*>(data + (size_t)w * 4096);
      for (int i = tid; i < 1024; i += 256) mine[i] = make_float4(b + w, 1.f, 2.f, 3.f);
    }
    __syncthreads();
    if (tid == 0) {
      __threadfence();  // release: this workgroup's stores are visible device-wide
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = base + (unsigned)(b + 1) * (unsigned)G;
      long long spins = 0;
      while ((int)(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > 4000000LL) { *err = 1; break; }
      }
      __threadfence();  // acquire
    }
    __syncthreads();
    if (payload) {
      const float4* other = reinterpret_cast<const float4*>(data + (size_t)((w + 97) % G) * 4096);
      for (int i = tid; i < 1024; i += 256) { float4 v = other[i]; acc += v.x; if (v.x != (float)(b + (w + 97) % G)) *err = 2; }
    }
  }
  if (acc == -1.f) sink[0] = acc;
}

int main() {
  float *d, *sink; unsigned* c; int* err;
  CK(hipMalloc((void**)&d, (size_t)1024 * 16384)); CK(hipMalloc((void**)&sink, 64)); CK(hipMalloc((void**)&c, 64)); CK(hipMalloc((void**)&err, 4));
  CK(hipMemset(c, 0, 64)); CK(hipMemset(err, 0, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  unsigned base = 0;
  const int nb = 200;
  for (int G : {32, 128, 256})
    for (int payload = 0; payload < 2; ++payload) {
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        k<<<G, 256>>>(d, c, base, nb, payload, err, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        base += (unsigned)nb * (unsigned)G;
      }
      int h; CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
      printf("%3d workgroups, %s: %.2f us per barrier round%s\n", G, payload ? "16 KB written + a neighbour's 16 KB read" : "barrier only                           ",
             ms * 1e3 / nb, h ? (h == 1 ? "  [SPIN LIMIT]" : "  [STALE DATA]") : "");
      CK(hipMemset(err, 0, 4));
    }
  return 0;
}
