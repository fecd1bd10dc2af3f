// Development harness for fused_i8_kernel (blr_fused_i8.hpp): B synthetic regressors at D = 128, fp64, aligned ColVecs, isotropic
// noise, diagonal prior; timed with HIP events; every output compared with fused_small_kernel<double, 8, 4> on the same inputs
// (and the first regressors' evidence with a plain host evaluation).  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../bayesianlinearregressors.jl_amd/csrc i8_gram.hip -o i8_gram
//   ./i8_gram [B] [N] [reps] [mode]     mode 0: N(0,1) inputs   1: one far outlier in every other regressor (repair path)   2: rows of very different scale   3: X = 0   4: ~10 entries per regressor 6 x larger (wrapped values, repair path)   5: NaN in every other regressor (hand-back path)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "blr_fused_i8.hpp"
using namespace blr;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  typedef double T;
  constexpr int D = 128;
  const int B = argc > 1 ? atoi(argv[1]) : 4096;
  const int N = argc > 2 ? atoi(argv[2]) : 4096;
  const int reps = argc > 3 ? atoi(argv[3]) : 10;
  const int mode = argc > 4 ? atoi(argv[4]) : 0;
  const int BU = std::min(B, 32);
  std::vector<T> X((size_t)BU * N * D), y((size_t)BU * N), mw(D, T(0)), dpr(D);
  unsigned long long st = 88172645463325252ULL;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0; };
  auto gauss = [&]() { return std::sqrt(-2.0 * std::log(rnd() + 1e-300)) * std::cos(6.283185307179586 * rnd()); };
  for (auto& v : X) v = gauss();
  for (auto& v : y) v = 3.0 * gauss();
  for (int i = 0; i < D; ++i) dpr[i] = 0.5 + rnd();
  if (getenv("I8_MW")) for (int i = 0; i < D; ++i) mw[i] = gauss();  // a prior mean (folded in after the stream)
  if (mode == 2)
    for (int b = 0; b < BU; ++b)
      for (int n = 0; n < N; ++n)
        for (int i = 0; i < D; ++i) X[((size_t)b * N + n) * D + i] *= std::ldexp(1.0, (i % 7) * 9 - 27);  // rows from 2^-27 to 2^27
  if (mode == 3) std::fill(X.begin(), X.end(), 0.0);  // zero operands: what the clock does without data toggling
  if (mode == 1)
    for (int b = 0; b < BU; b += 2) X[((size_t)b * N + N / 2) * D + 5] = 1.0e6;  // far beyond the capacity of its row in every other regressor
  if (mode == 4)
    for (auto& v : X) if (rnd() < 2.0e-5) v *= 6.0;
  if (mode == 5)
    for (int b = 0; b < BU; b += 2) X[((size_t)b * N + N / 2) * D + 5] = std::nan("");
  const T s_iso = 0.1;
  T *dX, *dy, *ds, *dmw, *dpri; CK(hipMalloc((void**)&dX, (size_t)B * N * D * 8)); CK(hipMalloc((void**)&dy, (size_t)B * N * 8));
  CK(hipMalloc((void**)&ds, 8)); CK(hipMalloc((void**)&dmw, D * 8)); CK(hipMalloc((void**)&dpri, D * 8));
  for (int b0 = 0; b0 < B; b0 += BU) {
    const int nb = std::min(BU, B - b0);
    CK(hipMemcpy(dX + (size_t)b0 * N * D, X.data(), (size_t)nb * N * D * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dy + (size_t)b0 * N, y.data(), (size_t)nb * N * 8, hipMemcpyHostToDevice));
  }
  CK(hipMemcpy(ds, &s_iso, 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dmw, mw.data(), D * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dpri, dpr.data(), D * 8, hipMemcpyHostToDevice));
  struct Out { T *mwp, *Tp, *Lp; double* lp; int32_t* info; } o[2];
  for (int k = 0; k < 2; ++k) {
    CK(hipMalloc((void**)&o[k].mwp, (size_t)B * D * 8)); CK(hipMalloc((void**)&o[k].Tp, (size_t)B * D * D * 8));
    CK(hipMalloc((void**)&o[k].Lp, (size_t)BU * D * D * 8));
    CK(hipMalloc((void**)&o[k].lp, (size_t)B * 8)); CK(hipMalloc((void**)&o[k].info, (size_t)B * 4));
    CK(hipMemset(o[k].info, 0xff, (size_t)B * 4));
  }
  auto args = [&](int k) {
    PosteriorArgs<T> a{};
    a.X = dX; a.ldx = D; a.strideX = (int64_t)N * D; a.y = dy; a.stridey = N; a.s = ds; a.strides = 0;
    a.mw = dmw; a.stridemw = 0; a.Lw = dpri; a.ldl = 1; a.strideLw = 0;
    a.mw_post = o[k].mwp; a.stride_mwpost = D; a.T_post = o[k].Tp; a.ldt = D; a.strideT = D * D; a.Lw_post = nullptr;
    a.logpdf = o[k].lp; a.info = o[k].info; a.layout = LAYOUT_COLVECS; a.noise_kind = NOISE_ISOTROPIC;
    a.prior_kind = PRIOR_DIAGONAL; a.D = D; a.N = N; a.B = B; a.vec_ok = 1;
    return a;
  };
  using SC = SmallCfg<T, 8>;
#ifndef I8_NG
#define I8_NG 6
#endif
  auto ki8 = fused_i8_kernel<false, false, I8_NG>;
  auto k64 = fused_small_kernel<T, 8, 4>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k64), hipFuncAttributeMaxDynamicSharedMemorySize, SC::LDS_BYTES));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(ki8), hipFuncAttributeMaxDynamicSharedMemorySize, I8Cfg::LDS_BYTES));
  const int stag_first = getenv("I8_STAG_FIRST") ? atoi(getenv("I8_STAG_FIRST")) : (B >= 1024 ? 256 : 0);
  const int stag_ticks = getenv("I8_STAG_TICKS") ? atoi(getenv("I8_STAG_TICKS")) : 25000;  // 250 us
  unsigned long long* dcnt; CK(hipMalloc((void**)&dcnt, 64)); CK(hipMemset(dcnt, 0, 64));
  auto run_i8 = [&](bool lp_only) {
    PosteriorArgs<T> a = args(0);
    a.i8_handed_tot = dcnt; a.i8_handed_slice = dcnt + 1;
    if (lp_only) { a.mw_post = nullptr; a.T_post = nullptr; }
    hipLaunchKernelGGL(ki8, dim3(B), dim3(kI8Threads), I8Cfg::LDS_BYTES, 0, a);
    a.retry_only = 1;
    hipLaunchKernelGGL(k64, dim3(B), dim3(kThreads), SC::LDS_BYTES, 0, a);
  };
  auto run_f64 = [&]() { PosteriorArgs<T> a = args(1); hipLaunchKernelGGL(k64, dim3(B), dim3(kThreads), SC::LDS_BYTES, 0, a); };
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms_i8, ms_f64;
  for (int w = 0; w < 2; ++w) run_i8(false);
  CK(hipDeviceSynchronize()); CK(hipGetLastError());
  CK(hipEventRecord(e0));
  const bool lponly = getenv("I8_LPONLY") != nullptr;  // timing experiment: no mw', no T written
  for (int r = 0; r < reps; ++r) run_i8(lponly);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_i8, e0, e1)); ms_i8 /= reps;
  { PosteriorArgs<T> a = args(0); a.Lw_post = o[0].Lp; a.ldlp = D; a.strideLp = D * D; a.B = BU;
    hipLaunchKernelGGL(ki8, dim3(BU), dim3(kI8Threads), I8Cfg::LDS_BYTES, 0, a); a.retry_only = 1;
    hipLaunchKernelGGL(k64, dim3(BU), dim3(kThreads), SC::LDS_BYTES, 0, a); }
  run_f64(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < std::max(1, reps / 3); ++r) run_f64();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_f64, e0, e1)); ms_f64 /= std::max(1, reps / 3);
  { PosteriorArgs<T> a = args(1); a.Lw_post = o[1].Lp; a.ldlp = D; a.strideLp = D * D; a.B = BU; hipLaunchKernelGGL(k64, dim3(BU), dim3(kThreads), SC::LDS_BYTES, 0, a); }
  CK(hipDeviceSynchronize());
#ifdef BLR_I8_STAMPS
  {
    unsigned long long z[8][20] = {};
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_i8stamps), z, sizeof(z)));
    PosteriorArgs<T> a = args(0);
    hipLaunchKernelGGL(ki8, dim3(B), dim3(kI8Threads), I8Cfg::LDS_BYTES, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipMemcpyFromSymbol(z, HIP_SYMBOL(g_i8stamps), sizeof(z)));
    if (getenv("I8_SUSTAINED")) {  // >= 2 s of back-to-back launches, then the in-kernel clocks of 20 more (cycles / 100 MHz ticks per workgroup)
      const double secs = atof(getenv("I8_SUSTAINED"));
      hipEvent_t s0, s1; CK(hipEventCreate(&s0)); CK(hipEventCreate(&s1));
      float el = 0; CK(hipEventRecord(s0));
      int nl = 0;
      while (el < secs * 1e3) { for (int q = 0; q < 16; ++q) hipLaunchKernelGGL(ki8, dim3(B), dim3(kI8Threads), I8Cfg::LDS_BYTES, 0, a); nl += 16;
        CK(hipEventRecord(s1)); CK(hipEventSynchronize(s1)); CK(hipEventElapsedTime(&el, s0, s1)); }
      unsigned long long zc[4] = {}; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_i8clk), zc, sizeof(zc)));
      CK(hipEventRecord(s0));
      for (int q = 0; q < 20; ++q) hipLaunchKernelGGL(ki8, dim3(B), dim3(kI8Threads), I8Cfg::LDS_BYTES, 0, a);
      CK(hipEventRecord(s1)); CK(hipEventSynchronize(s1)); CK(hipEventElapsedTime(&el, s0, s1));
      CK(hipMemcpyFromSymbol(zc, HIP_SYMBOL(g_i8clk), sizeof(zc)));
      printf("  sustained (mode %d, %d launches = %.1f s of pre-heat): %.3f ms per launch of %d = %.3f M updates/s; in-kernel clock %.3f GHz (%.0f k cycles, %.1f us per regressor, %llu workgroups sampled)\n",
             mode, nl, secs, el / 20, B, B / (el / 20) / 1e3, (double)zc[0] / (double)zc[1] * 0.1, (double)zc[0] / zc[2] / 1e3, (double)zc[1] / zc[2] / 100.0, zc[2]);
    }
    const double nwg = (B + 256) / 257, nk = N / 32 * nwg;  // (sums over the workgroups with blockIdx % 257 == 0: different replicas of the inputs)
    for (int w = 0; w < 8; ++w)
      printf("  wave %d, mean over %d workgroups, cycles per k-step: MFMAs + slicing %6.0f | DMA wait + barrier %6.0f || per regressor: stream %8.0f | hand-over + conversion %7.0f | repair %6.0f | chol %7.0f | backsolve + out %7.0f\n",
             w, (int)nwg, z[w][0] / nk, z[w][2] / nk, z[w][4] / nwg, z[w][5] / nwg, z[w][3] / nwg, z[w][6] / nwg, z[w][7] / nwg);
    for (int w = 0; w < 4; ++w)
      printf("    wave %d back substitution in detail: reciprocal pivots + barrier %6.0f | substitution (wave 0) / T written (waves 1-3) %6.0f | logdet %6.0f | closing barrier %6.0f\n",
             w, z[w][12] / nwg, z[w][13] / nwg, z[w][14] / nwg, z[w][15] / nwg);
    for (int w = 0; w < 8; ++w)
      printf("    wave %d hand-over in detail: sums + tables + barriers %6.0f | conversion phase 0 %6.0f | (barrier +) phase 1 %6.0f | (barrier +) phase 2 %6.0f | barrier %6.0f | table pass %6.0f | rest (b, y'y, barrier) %6.0f\n",
             w, z[w][8] / nwg, z[w][9] / nwg, z[w][10] / nwg, z[w][11] / nwg, z[w][16] / nwg, z[w][17] / nwg, z[w][5] / nwg);
  }
#endif
  // compare
  std::vector<double> lp0(B), lp1(B); std::vector<int32_t> i0(B), i1(B);
  CK(hipMemcpy(lp0.data(), o[0].lp, (size_t)B * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(lp1.data(), o[1].lp, (size_t)B * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(i0.data(), o[0].info, (size_t)B * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(i1.data(), o[1].info, (size_t)B * 4, hipMemcpyDeviceToHost));
  std::vector<T> m0((size_t)BU * D), m1((size_t)BU * D), T0((size_t)BU * D * D), T1((size_t)BU * D * D), A0((size_t)BU * D * D), A1((size_t)BU * D * D);
  CK(hipMemcpy(m0.data(), o[0].mwp, m0.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(m1.data(), o[1].mwp, m1.size() * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(T0.data(), o[0].Tp, T0.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(T1.data(), o[1].Tp, T1.size() * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(A0.data(), o[0].Lp, A0.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(A1.data(), o[1].Lp, A1.size() * 8, hipMemcpyDeviceToHost));
  double e_lp = 0, e_m = 0, e_T = 0, e_A = 0; int bad = 0, exact_lp = 0;
  for (int b = 0; b < B; ++b) {
    if (i0[b] != i1[b] || i0[b] != 0) ++bad;
    e_lp = std::max(e_lp, std::fabs(lp0[b] - lp1[b]) / std::fabs(lp1[b]));
    if (lp0[b] == lp1[b]) ++exact_lp;
    if (lp0[b] != lp0[b % BU]) ++bad;  // replicas of one input must agree bit for bit
  }
  for (int b = 0; b < BU; ++b) {
    double mm = 0, tm = 0, am = 0;
    for (int i = 0; i < D; ++i) mm = std::max(mm, std::fabs(m1[(size_t)b * D + i]));
    for (size_t i = 0; i < (size_t)D * D; ++i) { tm = std::max(tm, std::fabs(T1[(size_t)b * D * D + i])); am = std::max(am, std::fabs(A1[(size_t)b * D * D + i])); }
    for (int i = 0; i < D; ++i) e_m = std::max(e_m, std::fabs(m0[(size_t)b * D + i] - m1[(size_t)b * D + i]) / mm);
    for (size_t i = 0; i < (size_t)D * D; ++i) {
      e_T = std::max(e_T, std::fabs(T0[(size_t)b * D * D + i] - T1[(size_t)b * D * D + i]) / tm);
      e_A = std::max(e_A, std::fabs(A0[(size_t)b * D * D + i] - A1[(size_t)b * D * D + i]) / am);
    }
  }
  unsigned long long hcnt[2]; CK(hipMemcpy(hcnt, dcnt, 16, hipMemcpyDeviceToHost));
  printf("  handed back to the fp64 kernel: %llu regressors over %d launches of %d\n", hcnt[0], reps + 2, B);
  printf("D=128 N=%d B=%d mode %d: int8 path %.3f ms/launch = %.3f M updates/s (%.2f TB/s of X) | fp64 kernel %.3f ms = %.3f M updates/s\n", N, B, mode,
         ms_i8, B / ms_i8 / 1e3, (double)B * N * D * 8 / ms_i8 / 1e9, ms_f64, B / ms_f64 / 1e3);
  printf("  int8 vs fp64 kernel: logpdf max rel diff %.2e (%d of %d bit-equal) | A = Lw' max |diff| / max|A| %.2e | mw' %.2e | T %.2e | status/replica mismatches %d\n",
         e_lp, exact_lp, B, e_A, e_m, e_T, bad);
  return (mode != 5 && bad != 0) || !(e_lp < 1e-11) || !(e_m < 1e-9) || !(e_T < 1e-9);
}
