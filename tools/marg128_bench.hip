// Development harness for the D = 128 marginal stream (marg_image_kernel + marginals_gemm_kernel, blr_marginals.hpp): B regressors,
// N inputs each, fp64; timing of the stream kernel alone for a given number of workgroups per regressor.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../bayesianlinearregressors.jl_amd/csrc marg128_bench.hip -o marg128_bench
//   ./marg128_bench [B] [N] [per_reg] [reps]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "blr_marginals.hpp"
using namespace blr;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
  typedef double T;
  const int D = 128;
  const int B = argc > 1 ? atoi(argv[1]) : 64, N = argc > 2 ? atoi(argv[2]) : 4096, per_reg = argc > 3 ? atoi(argv[3]) : 16, reps = argc > 4 ? atoi(argv[4]) : 20;
  std::vector<T> X((size_t)D * N), U((size_t)D * D, 0.0), mw(D, 0.5);
  unsigned long long st = 88172645463325252ULL;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0 - 0.5; };
  for (auto& v : X) v = rnd();
  for (int j = 0; j < D; ++j) for (int i = 0; i <= j; ++i) U[(size_t)j * D + i] = (i == j) ? 2.0 : rnd() / std::sqrt((double)D);
  T *dX, *dU, *dmw, *ds, *dimg, *dmean, *dvar; int32_t* dinfo;
  CK(hipMalloc((void**)&dX, (size_t)B * D * N * 8)); CK(hipMalloc((void**)&dU, U.size() * 8)); CK(hipMalloc((void**)&dmw, D * 8)); CK(hipMalloc((void**)&ds, 8));
  CK(hipMalloc((void**)&dimg, (size_t)B * MargGemmCfg<T>::IMG_ELEMS * 8)); CK(hipMalloc((void**)&dmean, (size_t)B * N * 8)); CK(hipMalloc((void**)&dvar, (size_t)B * N * 8));
  CK(hipMalloc((void**)&dinfo, (size_t)B * 4)); CK(hipMemset(dinfo, 0, (size_t)B * 4));
  for (int b = 0; b < B; ++b) CK(hipMemcpy(dX + (size_t)b * D * N, X.data(), X.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dU, U.data(), U.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dmw, mw.data(), D * 8, hipMemcpyHostToDevice)); const T s_iso = 0.1; CK(hipMemcpy(ds, &s_iso, 8, hipMemcpyHostToDevice));
  using G = MargGemmCfg<T>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(marg_image_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, TrsmCfg<T>::LDS_BYTES));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(marginals_gemm_kernel<T, false>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
  MarginalArgs<T> a{};
  a.X = dX; a.ldx = D; a.strideX = (int64_t)D * N; a.layout = LAYOUT_COLVECS; a.s = ds; a.strides = 0; a.noise_kind = NOISE_ISOTROPIC;
  a.mw = dmw; a.stridemw = 0; a.U = dU; a.ldu = D; a.strideU = 0; a.prior_kind = PRIOR_UPPER_FACTOR;
  a.mean = dmean; a.stridemean = N; a.var = dvar; a.stridevar = N; a.info = dinfo; a.D = D; a.N = N; a.reg0 = 0;
  hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  float ms_img = 0, ms_gemm = 0;
  for (int r = 0; r < reps + 3; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(marg_image_kernel<T>, dim3(B, 2), dim3(kThreads), TrsmCfg<T>::LDS_BYTES, 0, (const T*)dU, (int64_t)D, (int64_t)0, D, dimg, (const int32_t*)dinfo, 0);
    CK(hipEventRecord(e1));
    hipLaunchKernelGGL((marginals_gemm_kernel<T, false>), dim3(per_reg, B), dim3(kThreads), G::LDS_BYTES, 0, a, (const T*)dimg);
    CK(hipEventRecord(e2)); CK(hipEventSynchronize(e2));
    float t1, t2; CK(hipEventElapsedTime(&t1, e0, e1)); CK(hipEventElapsedTime(&t2, e1, e2));
    if (r >= 3) { ms_img += t1 / reps; ms_gemm += t2 / reps; }
  }
  CK(hipGetLastError());
  std::vector<T> var(N); CK(hipMemcpy(var.data(), dvar + (size_t)(B - 1) * N, (size_t)N * 8, hipMemcpyDeviceToHost));
  double worst = 0;
  for (int n : {0, 1, 15, 16, N / 2 + 5, N - 1}) {
    std::vector<double> z(D);
    double sq = 0;
    for (int j = 0; j < D; ++j) {
      double acc = X[(size_t)n * D + j];
      for (int d = 0; d < j; ++d) acc -= U[(size_t)j * D + d] * z[d];
      z[j] = acc / U[(size_t)j * D + j];
      sq += z[j] * z[j];
    }
    worst = std::fmax(worst, std::fabs(var[n] - (sq + 0.1)) / (sq + 0.1));
  }
  printf("B=%d N=%d f64, %d workgroups per regressor: image %.1f us + stream %.1f us = %.2f G marginals/s (stream alone %.1f TFLOP/s of N D^2) | var max rel err %.2e\n",
         B, N, per_reg, 1e3 * ms_img, 1e3 * ms_gemm, (double)B * N / (ms_img + ms_gemm) / 1e6, (double)B * N * D * D / ms_gemm / 1e9, worst);
  return 0;
}
