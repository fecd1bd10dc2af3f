// Development harness for marg_blocksub_kernel (blr_marginals.hpp): one factor of order D, N inputs, fp32; timing only (parity is
// tests/test_gpu_parity.py::test_large_d_marginals through the ABI).  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../bayesianlinearregressors.jl_amd/csrc marg_bench.hip -o marg_bench [-DBLR_MB_EXP=n]
//   ./marg_bench [D] [N] [reps]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "blr_marginals.hpp"
using namespace blr;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
  typedef float T;
  const int D = argc > 1 ? atoi(argv[1]) : 1024, N = argc > 2 ? atoi(argv[2]) : 65536, reps = argc > 3 ? atoi(argv[3]) : 20;
  const int DP = (D + 127) / 128 * 128, NC = DP / 128;
  std::vector<T> X((size_t)D * N), U((size_t)D * D, 0.f), mw(D, 0.5f);
  unsigned long long st = 88172645463325252ULL;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0 - 0.5; };
  for (auto& v : X) v = (T)rnd();
  for (int j = 0; j < D; ++j) for (int i = 0; i <= j; ++i) U[(size_t)j * D + i] = (i == j) ? 2.0f : (T)(rnd() / std::sqrt((double)D));
  T *dX, *dU, *dmw, *ds, *dimg, *dmean, *dvar; int32_t* dinfo;
  CK(hipMalloc((void**)&dX, X.size() * 4)); CK(hipMalloc((void**)&dU, U.size() * 4)); CK(hipMalloc((void**)&dmw, D * 4)); CK(hipMalloc((void**)&ds, 4));
  CK(hipMalloc((void**)&dimg, (size_t)NC * MargGemmCfg<T>::IMG_ELEMS * 4)); CK(hipMalloc((void**)&dmean, (size_t)N * 4)); CK(hipMalloc((void**)&dvar, (size_t)N * 4));
  CK(hipMalloc((void**)&dinfo, 4)); CK(hipMemset(dinfo, 0, 4));
  CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dU, U.data(), U.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dmw, mw.data(), D * 4, hipMemcpyHostToDevice)); const T s_iso = 0.1f; CK(hipMemcpy(ds, &s_iso, 4, hipMemcpyHostToDevice));
#ifndef MB_RT
#define MB_RT 32
#endif
  using MB = MargBlockCfg<T, MB_RT>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(marg_image_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, TrsmCfg<T>::LDS_BYTES));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(marg_blocksub_kernel<T, MB_RT>), hipFuncAttributeMaxDynamicSharedMemorySize, MB::kMaxLds));
  hipLaunchKernelGGL(marg_image_kernel<T>, dim3(NC, 2), dim3(kThreads), TrsmCfg<T>::LDS_BYTES, 0, (const T*)dU, (int64_t)D, (int64_t)128 * (D + 1), 128, dimg,
                     (const int32_t*)dinfo, 0, D, NC, (int64_t)0);
  MargBlockArgs<T> m{};
  m.X = dX; m.ldx = D; m.U = dU; m.ldu = D; m.img = dimg; m.mw = dmw; m.s = ds; m.noise_kind = NOISE_ISOTROPIC; m.mean = dmean; m.var = dvar;
  m.info = dinfo; m.D = D; m.Dx = D; m.DP = DP; m.N = N;  // (one regressor: the batch strides stay zero)
  int cus = 256; { hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0)); cus = pr.multiProcessorCount; }
  const int ntiles = (N + MB_RT - 1) / MB_RT, slots = cus * (32 / MB_RT), grid = ntiles < slots ? ntiles : slots;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((marg_blocksub_kernel<T, MB_RT>), dim3(grid), dim3(MB::THREADS), MB::lds_bytes(DP), 0, m);
  CK(hipDeviceSynchronize()); CK(hipGetLastError());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((marg_blocksub_kernel<T, MB_RT>), dim3(grid), dim3(MB::THREADS), MB::lds_bytes(DP), 0, m);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
#ifdef BLR_MB_STAMPS
  {
    unsigned long long z[8][8];
    CK(hipMemcpyFromSymbol(z, HIP_SYMBOL(g_mbstamps), sizeof(z)));
    for (int w = 0; w < 8; ++w)
      printf("  wave %d of WG 0, cycles per launch: product MFMA %8llu | factor wait %7llu | RMW %6llu | barriers %7llu | image wait %6llu | diag %6llu | input wait %6llu | other %7llu\n",
             w, z[w][0], z[w][1], z[w][2], z[w][3], z[w][4], z[w][5], z[w][6], z[w][7]);
  }
#endif
  std::vector<T> var(N); CK(hipMemcpy(var.data(), dvar, (size_t)N * 4, hipMemcpyDeviceToHost));
  // host check of a few inputs: z = L^-1 x, L = U'
  double worst = 0;
  for (int n : {0, 1, 31, 32, N / 2 + 5, N - 1}) {
    std::vector<double> z(D);
    double sq = 0;
    for (int j = 0; j < D; ++j) {
      double acc = X[(size_t)n * D + j];
      for (int d = 0; d < j; ++d) acc -= (double)U[(size_t)j * D + d] * z[d];
      z[j] = acc / U[(size_t)j * D + j];
      sq += z[j] * z[j];
    }
    worst = std::fmax(worst, std::fabs(var[n] - (sq + 0.1)) / (sq + 0.1));
  }
  printf("D=%d N=%d f32: %.3f ms per call = %.1f TFLOP/s (N D^2), %.2f TB/s of X | var max rel err on 6 inputs %.2e\n", D, N, ms,
         (double)N * D * D / ms / 1e9, (double)N * D * 4 / ms / 1e9, worst);
  return 0;
}
