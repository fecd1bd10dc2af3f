// Panel step of the large-D Cholesky in isolation: panel_chain_kernel (blr_panel.hpp).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../bayesianlinearregressors.jl_amd/csrc panel_bench.hip -o panel_bench
// Checks L_pp and X L_pp^-T against a host double-precision factorisation, then times the launch (A restored before every one).
// With -DBLR_STAMPS: section sums of the chain wave and of update wave 0, and the time line of workgroup 0's last launch.
// (The kernel it replaced, round 2's panel_factor_kernel: 36.8 us f32 / 51.1 us f64 for the same panel.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "blr_large.hpp"
using namespace blr;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#ifndef PB_NW
#define PB_NW BLR_PANEL_WAVES
#endif
#ifndef PB_NBT
#define PB_NBT 8   // 16 x 16 tiles along the block's edge (128-column panel)
#endif

template <typename T>
int run(const char* name, int nwg) {
  constexpr int W = 16 * PB_NBT;  // panel width
  const int nrows = W + ChainCfg<T, PB_NW, BLR_PANEL_ER, PB_NBT>::ER * nwg;
  const int64_t lda = nrows;
  std::vector<double> A((size_t)lda * W);
  srand(7);
  // SPD block: G G' / 16 + I, rows below random
  std::vector<double> G(W * W);
  for (auto& g : G) g = (rand() / (double)RAND_MAX) - 0.5;
  for (int c = 0; c < W; ++c)
    for (int r2 = 0; r2 < W; ++r2) {
      double s = 0;
      for (int k = 0; k < W; ++k) s += G[r2 * W + k] * G[c * W + k];
      A[(size_t)c * lda + r2] = s / 16.0 + (r2 == c ? 1.0 : 0.0);
    }
  for (int c = 0; c < W; ++c)
    for (int r2 = W; r2 < nrows; ++r2) A[(size_t)c * lda + r2] = (rand() / (double)RAND_MAX) - 0.5;
  // host reference
  std::vector<double> R = A;
  for (int c = 0; c < W; ++c) {
    double d = R[(size_t)c * lda + c];
    for (int k = 0; k < c; ++k) d -= R[(size_t)k * lda + c] * R[(size_t)k * lda + c];
    d = std::sqrt(d);
    R[(size_t)c * lda + c] = d;
    for (int r2 = c + 1; r2 < nrows; ++r2) {
      double s = R[(size_t)c * lda + r2];
      for (int k = 0; k < c; ++k) s -= R[(size_t)k * lda + r2] * R[(size_t)k * lda + c];
      R[(size_t)c * lda + r2] = s / d;
    }
  }
  std::vector<T> hA(A.size());
  for (size_t i = 0; i < A.size(); ++i) hA[i] = (T)A[i];
  T *dA, *dW;
  int32_t* dinfo;
  unsigned* darr;
  CK(hipMalloc((void**)&dA, hA.size() * sizeof(T)));
  CK(hipMalloc((void**)&dW, hA.size() * sizeof(T)));
  CK(hipMalloc((void**)&dinfo, 64));
  CK(hipMalloc((void**)&darr, 2 * 128 * 4));  // two banks of arrival words (blr_panel.hpp)
  CK(hipMemcpy(dA, hA.data(), hA.size() * sizeof(T), hipMemcpyHostToDevice));
  CK(hipMemset(dinfo, 0, 64));
  CK(hipMemset(darr, 0, 2 * 128 * 4));
  unsigned launches = 0;
  using CC = ChainCfg<T, PB_NW, BLR_PANEL_ER, PB_NBT>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(panel_chain_kernel<T, PB_NW, BLR_PANEL_ER, PB_NBT>), hipFuncAttributeMaxDynamicSharedMemorySize, CC::LDS_BYTES));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int which = 1; which < 2; ++which) {
    double tot = 0;
    const int reps = 20;
    std::vector<T> out(hA.size());
    for (int it = 0; it < reps + 2; ++it) {
      CK(hipMemcpy(dW, dA, hA.size() * sizeof(T), hipMemcpyDeviceToDevice));
      CK(hipEventRecord(e0));
      panel_chain_kernel<T, PB_NW, BLR_PANEL_ER, PB_NBT><<<nwg, 64 * PB_NW, CC::LDS_BYTES>>>(dW, lda, 0, nrows, dinfo, darr + 128 * (launches & 1), (unsigned)nwg, 0, 0, darr + 128 * ((launches & 1) ^ 1)); ++launches;
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (it >= 2) tot += ms;
    }
    CK(hipGetLastError());
    CK(hipMemcpy(out.data(), dW, out.size() * sizeof(T), hipMemcpyDeviceToHost));
    int info;
    CK(hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0;
    size_t worst = 0;
    for (int c = 0; c < W; ++c)
      for (int r2 = c; r2 < nrows; ++r2) {
        const size_t i = (size_t)c * lda + r2;
        const double e = std::fabs((double)out[i] - R[i]);
        if (!(e <= maxerr)) { maxerr = e; worst = i; }
        maxref = std::fmax(maxref, std::fabs(R[i]));
      }
#ifdef BLR_STAMPS
    if (which == 1) {
      unsigned long long st[8];
      CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st)));
      const int n = reps + 2;
      printf("  chain wave cycles per launch: prologue %llu | tile->rows %llu | factor %llu | publish+B1 %llu | solve+apply %llu | B2 %llu | - %llu\n",
             st[0] / n, st[1] / n, st[2] / n, st[3] / n, st[4] / n, st[5] / n, st[6] / n);
    }
    if (which == 1) {
      unsigned long long st[8];
      CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps2), sizeof(st)));
      const int n = reps + 2;
      printf("  update wave 0 cycles per launch: (din scan) %llu | pre-write %llu | B1 wait %llu | solve %llu | B2 wait %llu | trailing %llu\n",
             st[0] / n, st[1] / n, st[2] / n, st[3] / n, st[4] / n, st[5] / n);
    }
    if (which == 1 && sizeof(T) == 4) {
      static unsigned long long tl[8][80];
      CK(hipMemcpyFromSymbol(tl, HIP_SYMBOL(g_tl), sizeof(tl)));
      unsigned long long t0 = ~0ull;
      for (int w = 0; w < PB_NW; ++w) if (tl[w][0] && tl[w][0] < t0) t0 = tl[w][0];
      printf("  time line of the last launch, workgroup 0 (cycles since the first wave's start)\n");
      printf("  wave  start | per step J: [ready for B1, past B1, ready for B2, past B2] ... | loop end, kernel end\n");
      for (int w = 0; w < PB_NW; ++w) {
        printf("  w%d %6llu |", w, tl[w][0] - t0);
        for (int J = 0; J < PB_NBT; ++J) printf(" [%llu %llu %llu %llu]", tl[w][1 + 4 * J] - t0, tl[w][2 + 4 * J] - t0, tl[w][3 + 4 * J] - t0, tl[w][4 + 4 * J] - t0);
        printf(" | %llu %llu\n", tl[w][72] - t0, tl[w][73] - t0);
      }
    }
    { unsigned long long zero[8] = {0}; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), zero, sizeof(zero))); CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps2), zero, sizeof(zero))); }
#endif
    printf("%s %s nwg=%d: %.2f us  max|err| = %.3e (at row %zu col %zu, ref %.4g)  info=%d\n", name, "chain ", nwg,
           tot * 1e3 / reps, maxerr, worst % lda, worst / lda, R[worst], info);
  }
  // failure path: a non-positive pivot at column 37 must come back as info = 38 and leave the block alone
  {
    std::vector<T> hB = hA;
    hB[(size_t)37 * lda + 37] = (T)-1.0;
    CK(hipMemcpy(dW, hB.data(), hB.size() * sizeof(T), hipMemcpyHostToDevice));
    panel_chain_kernel<T, PB_NW, BLR_PANEL_ER, PB_NBT><<<nwg, 64 * PB_NW, CC::LDS_BYTES>>>(dW, lda, 0, nrows, dinfo, darr + 128 * (launches & 1), (unsigned)nwg, 0, 0, darr + 128 * ((launches & 1) ^ 1)); ++launches;
    CK(hipDeviceSynchronize());
    int info;
    CK(hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost));
    std::vector<T> out(hB.size());
    CK(hipMemcpy(out.data(), dW, out.size() * sizeof(T), hipMemcpyDeviceToHost));
    size_t changed = 0;
    for (size_t i = 0; i < out.size(); ++i) changed += out[i] != hB[i];
    printf("%s chain  bad pivot at column 37: info=%d (want 38), %zu entries changed\n", name, info, changed);
    CK(hipMemset(dinfo, 0, 64));
  }
  return 0;
}

int main(int argc, char** argv) {
  const int nwg = argc > 1 ? atoi(argv[1]) : 30;
  if (run<float>("f32", nwg)) return 1;
  if (run<double>("f64", nwg)) return 1;
  return 0;
}
