// Diagnostic: where do the cycles of phase_chol (128 x 128 blocked Cholesky in LDS/registers) go?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DBLR_STAMPS -I../bayesianlinearregressors.jl_amd/csrc chol_bench.hip -o chol_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include "blr_fused_small.hpp"
#ifndef BLR_STAMPS
namespace blr { __device__ unsigned long long g_stamps[8]; }
#endif
using namespace blr;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <typename T>
__global__ __launch_bounds__(256, 2) void k(const T* A, T* out, int* info, int reps) {
  using C = SmallCfg<T, 8>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* P = reinterpret_cast<T*>(smem);
  T* bvec = reinterpret_cast<T*>(smem + C::OFF_B);
  const int tid = threadIdx.x;
  int rc = 0;
  for (int rep = 0; rep < reps; ++rep) {
    __syncthreads();
    for (int idx = tid; idx < 128 * 128; idx += 256) { int c = idx >> 7, r = idx & 127; if (r >= c) P[pidx(r, c)] = A[c * 128 + r]; }
    if (tid < 128) bvec[tid] = T(1);
    __syncthreads();
    rc = phase_chol<T, 8>(smem, 128, 1);
  }
  if (tid == 0) info[blockIdx.x] = rc;
  for (int idx = tid; idx < 128 * 128; idx += 256) { int c = idx >> 7, r = idx & 127; if (r >= c) out[c * 128 + r] = P[pidx(r, c)]; }
}

template <typename T>
int run(const char* name) {
  using C = SmallCfg<T, 8>;
  std::vector<T> A(128 * 128);
  for (int c = 0; c < 128; ++c) for (int r = 0; r < 128; ++r) A[c * 128 + r] = (T)((r == c ? 200.0 : 0.0) + std::cos(0.37 * (r + 1) * (c + 1)) );
  for (int c = 0; c < 128; ++c) for (int r = 0; r < c; ++r) A[c * 128 + r] = A[r * 128 + c];
  T *dA, *dO; int* dI;
  CK(hipMalloc((void**)&dA, A.size() * sizeof(T))); CK(hipMalloc((void**)&dO, A.size() * sizeof(T))); CK(hipMalloc((void**)&dI, 4096));
  CK(hipMemcpy(dA, A.data(), A.size() * sizeof(T), hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<T>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 50;
  unsigned long long zero[8] = {0};
  for (int grid : {1, 256, 512}) {
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), zero, sizeof(zero)));
    float ms;
    k<T><<<grid, 256, C::LDS_BYTES>>>(dA, dO, dI, 2);
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), zero, sizeof(zero)));
    CK(hipEventRecord(e0));
    k<T><<<grid, 256, C::LDS_BYTES>>>(dA, dO, dI, reps);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long st[8];
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st)));
    int info; CK(hipMemcpy(&info, dI, 4, hipMemcpyDeviceToHost));
    printf("%s grid=%d: %.2f us per factorisation (info=%d)", name, grid, ms * 1e3 / reps, info);
    if (grid == 1) {
      printf("  cycles/fact: load-tiles %llu | barrier %llu | (a) store panel %llu | (b) eliminate %llu | writeback+barrier %llu | (c) trailing %llu",
             st[0] / reps, st[1] / reps, st[2] / reps, st[3] / reps, st[4] / reps, st[5] / reps);
    }
    printf("\n");
  }
  return 0;
}
int main() { run<float>("f32"); run<double>("f64"); return 0; }
