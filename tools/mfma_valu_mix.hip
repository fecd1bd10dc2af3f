// Microbenchmark: do f64 MFMA and f64 VALU FMA overlap on gfx950?  Per iteration each wave issues 8 MFMAs
// (independent accumulators) and, interleaved after each MFMA, NV independent v_fma_f64.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int NM, int NV>
__global__ __launch_bounds__(256) void k_mix(double* out, int iters, double a0, double b0) {
  d4 acc[NM > 0 ? NM : 1];
  for (int i = 0; i < (NM > 0 ? NM : 1); ++i) acc[i] = d4{0, 0, 0, 0};
  double x[NV > 0 ? NV : 1];
  for (int i = 0; i < (NV > 0 ? NV : 1); ++i) x[i] = a0 + i + threadIdx.x;
  double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (NM > 0) acc[i % (NM > 0 ? NM : 1)] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i % (NM > 0 ? NM : 1)], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV; ++v) x[v] = __builtin_fma(x[v], 1.0000001, a);
    }
  }
  double s = 0;
  for (int i = 0; i < (NM > 0 ? NM : 1); ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < (NV > 0 ? NV : 1); ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM, int NV>
int run(int cus, int w, int iters, double* buf) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int grid = cus * w;
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    k_mix<NM, NV><<<grid, 256>>>(buf, iters, 1.0, 2.0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  double mf = (NM > 0) ? (double)grid * 4 * iters * 8 * 2048.0 : 0.0;
  double vf = (double)grid * 256 * iters * 8 * NV * 2.0;
  printf("w=%d  mfma/iter=%d valu/mfma=%2d : %8.3f ms  MFMA %6.1f TF + VALU %6.1f TF = %6.1f TF\n", w, NM > 0 ? 8 : 0, NV, ms,
         mf / ms / 1e9, vf / ms / 1e9, (mf + vf) / ms / 1e9);
  return 0;
}

int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount;
  double* buf;
  CK(hipMalloc((void**)&buf, (size_t)cus * 8 * 256 * 8));
  const int iters = 4000;
  for (int w = 1; w <= 2; ++w) {
    run<8, 0>(cus, w, iters, buf);
    run<8, 4>(cus, w, iters, buf);
    run<8, 8>(cus, w, iters, buf);
    run<8, 12>(cus, w, iters, buf);
    run<8, 16>(cus, w, iters, buf);
    run<8, 24>(cus, w, iters, buf);
    run<0, 16>(cus, w, iters, buf);
  }
  return 0;
}
