// Development harness for the fused D <= 128 kernel: one launch of fused_small_kernel<T, NB, 4> over B synthetic regressors
// (ColVecs, 16-byte aligned: the LDS-DMA loader), timed with HIP events, log evidence of the first regressors checked
// against a plain host evaluation of the same formula.  Not part of the product (the library is csrc/libblr_mi355x.so).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../bayesianlinearregressors.jl_amd/csrc fused_bench.hip -o fused_bench
//   ./fused_bench [B] [N] [diag_noise 0/1] [reps]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "blr_fused_small.hpp"
#ifdef FB_WAVE
#include "blr_fused_wave.hpp"
#endif
#ifdef FB_D128
#include "blr_fused_d128_experiment.hpp"
#endif
using namespace blr;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#ifndef FB_D
#define FB_D 128
#endif
#ifndef FB_T
#define FB_T double
#endif

static double host_logpdf(const std::vector<FB_T>& X, const std::vector<FB_T>& y, const std::vector<FB_T>& s, bool diag, int D, int N) {
  std::vector<double> A((size_t)D * D, 0.0), b(D, 0.0);
  double q = 0, ld_s = 0;
  for (int n = 0; n < N; ++n) {
    const double sv = diag ? (double)s[n] : (double)s[0], w = 1.0 / sv;
    ld_s += std::log(sv);
    const double dl = (double)y[n];
    q += dl * dl * w;
    for (int i = 0; i < D; ++i) {
      const double xi = (double)X[(size_t)n * D + i];
      b[i] += xi * dl * w;
      for (int k = 0; k <= i; ++k) A[(size_t)i * D + k] += xi * w * (double)X[(size_t)n * D + k];
    }
  }
  for (int i = 0; i < D; ++i) A[(size_t)i * D + i] += 1.0;
  double ldA = 0, uu = 0;
  std::vector<double> u(D);
  for (int j = 0; j < D; ++j) {
    for (int i = j; i < D; ++i) {
      double v = A[(size_t)i * D + j];
      for (int k = 0; k < j; ++k) v -= A[(size_t)i * D + k] * A[(size_t)j * D + k];
      A[(size_t)i * D + j] = (i == j) ? std::sqrt(v) : v / A[(size_t)j * D + j];
    }
    double v = b[j];
    for (int k = 0; k < j; ++k) v -= A[(size_t)j * D + k] * u[k];
    u[j] = v / A[(size_t)j * D + j];
    uu += u[j] * u[j];
    ldA += 2.0 * std::log(A[(size_t)j * D + j]);
  }
  return -0.5 * (N * 1.8378770664093454835606594728112 + ld_s + q + ldA - uu);
}

int main(int argc, char** argv) {
  typedef FB_T T;
  constexpr int D = FB_D, NB = (D + 15) / 16;
  using C = SmallCfg<T, NB>;
  const int B = argc > 1 ? atoi(argv[1]) : 4096;
  const int N = argc > 2 ? atoi(argv[2]) : 4096;
  const bool diag = argc > 3 ? atoi(argv[3]) != 0 : false;
  const int reps = argc > 4 ? atoi(argv[4]) : 10;
  const bool shareX = argc > 5 ? atoi(argv[5]) != 0 : false;  // every regressor reads the SAME X (cache-resident): no HBM stream
  const int BU = std::min(B, 64);  // distinct regressors (the rest reuse them: strides wrap on the host side)
  std::vector<T> X((size_t)BU * N * D), y((size_t)BU * N), s(diag ? (size_t)BU * N : 1), mw(D, T(0)), dpr(D, T(1));
  unsigned long long st = 88172645463325252ULL;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0; };
  auto gauss = [&]() { return std::sqrt(-2.0 * std::log(rnd() + 1e-300)) * std::cos(6.283185307179586 * rnd()); };
  const bool zero_ops = getenv("FB_ZERO") != nullptr;  // all-zero operands: the clock the part holds when nothing toggles
  for (auto& v : X) v = zero_ops ? (T)0 : (T)gauss();
  for (auto& v : y) v = zero_ops ? (T)0 : (T)(3.0 * gauss());
  if (diag) for (auto& v : s) v = (T)std::exp(0.5 * gauss()); else s[0] = (T)0.1;
  T *dX, *dy, *ds, *dmw, *dpri, *dmwp, *dT; double* dlp; int32_t* dinfo;
  // FB_PAD: extra elements between consecutive regressors (a power-of-two stride makes every wave hit the same memory channel
  // bits at the same time -- partition camping -- when the waves run in lockstep)
  const size_t pad = getenv("FB_PAD") ? (size_t)atoll(getenv("FB_PAD")) : 0;
  const size_t strideXe = (size_t)N * D + pad;
  CK(hipMalloc((void**)&dX, (size_t)B * strideXe * sizeof(T)));
  CK(hipMalloc((void**)&dy, (size_t)B * N * sizeof(T)));
  CK(hipMalloc((void**)&ds, s.size() * (diag ? (size_t)B / BU + 1 : 1) * sizeof(T)));
  CK(hipMalloc((void**)&dmw, D * sizeof(T))); CK(hipMalloc((void**)&dpri, D * sizeof(T)));
  CK(hipMalloc((void**)&dmwp, (size_t)B * D * sizeof(T))); CK(hipMalloc((void**)&dT, (size_t)B * D * D * sizeof(T)));
  CK(hipMalloc((void**)&dlp, (size_t)B * 8)); CK(hipMalloc((void**)&dinfo, (size_t)B * 4));
  for (int b0 = 0; b0 < B; b0 += BU) {
    const int nb = std::min(BU, B - b0);
    for (int bb = 0; bb < nb; ++bb)
      CK(hipMemcpy(dX + (size_t)(b0 + bb) * strideXe, X.data() + (size_t)bb * N * D, (size_t)N * D * sizeof(T), hipMemcpyHostToDevice));
    CK(hipMemcpy(dy + (size_t)b0 * N, y.data(), (size_t)nb * N * sizeof(T), hipMemcpyHostToDevice));
  }
  CK(hipMemcpy(ds, s.data(), s.size() * sizeof(T), hipMemcpyHostToDevice));
  CK(hipMemcpy(dmw, mw.data(), D * sizeof(T), hipMemcpyHostToDevice));
  CK(hipMemcpy(dpri, dpr.data(), D * sizeof(T), hipMemcpyHostToDevice));
  PosteriorArgs<T> a{};
  a.X = dX; a.ldx = D; a.strideX = shareX ? 0 : (int64_t)strideXe; a.y = dy; a.stridey = N; a.s = ds; a.strides = 0;  // noise shared (diag: of regressor 0)
  a.mw = dmw; a.stridemw = 0; a.Lw = dpri; a.ldl = 1; a.strideLw = 0;
  a.mw_post = dmwp; a.stride_mwpost = D; a.T_post = dT; a.ldt = D; a.strideT = D * D; a.Lw_post = nullptr;
  a.logpdf = dlp; a.info = dinfo; a.layout = LAYOUT_COLVECS; a.noise_kind = diag ? NOISE_DIAGONAL : NOISE_ISOTROPIC;
  a.prior_kind = PRIOR_DIAGONAL; a.D = D; a.N = N; a.B = B; a.vec_ok = 1;
#if defined(FB_D128)
  const int kLds = D128Cfg<T>::LDS_BYTES, kBlock = kThreads, kGrid = std::min(B, getenv("FB_GRID") ? atoi(getenv("FB_GRID")) : 768);
  T* dscr; CK(hipMalloc((void**)&dscr, (size_t)kGrid * D128Cfg<T>::WG_SCRATCH * sizeof(T)));
  auto kern = [&](PosteriorArgs<T> aa) {};
  (void)kern;
#define FB_LAUNCH() hipLaunchKernelGGL(fused_d128_kernel<T>, dim3(kGrid), dim3(kBlock), kLds, 0, a, dscr)
#elif defined(FB_WAVE)
  auto kern = fused_wave_kernel<T, NB>;
  const int kLds = WaveCfg<T, NB>::LDS_BYTES, kBlock = 64, kGrid = std::min(B, getenv("FB_GRID") ? atoi(getenv("FB_GRID")) : 2048);
#else
  auto kern = fused_small_kernel<T, NB, 4>;
  const int kLds = C::LDS_BYTES, kBlock = kThreads, kGrid = getenv("FB_GRID") ? atoi(getenv("FB_GRID")) : B;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
#endif
#ifndef FB_LAUNCH
#define FB_LAUNCH() hipLaunchKernelGGL(kern, dim3(kGrid), dim3(kBlock), kLds, 0, a)
#endif
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) FB_LAUNCH();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) FB_LAUNCH();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
#ifdef BLR_GRAM_STAMPS
  {
    unsigned long long zero[4][8] = {};
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_gstamps), zero, sizeof(zero)));
    unsigned long long zero16[16] = {};
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_pstamps), zero16, sizeof(zero16)));
    FB_LAUNCH();
    CK(hipDeviceSynchronize());
    unsigned long long stp[4][8];
    CK(hipMemcpyFromSymbol(stp, HIP_SYMBOL(g_gstamps), sizeof(stp)));
    unsigned long long ps[16];
    CK(hipMemcpyFromSymbol(ps, HIP_SYMBOL(g_pstamps), sizeof(ps)));
    if (ps[15]) {
      const double nr = (double)ps[15];
      printf("  WG 0, cycles per regressor (%.0f regressors): prior %.0f | gram %.0f | A out %.0f | chol %.0f | T out %.0f | backsolve %.0f | mw/logpdf out %.0f\n",
             nr, ps[0] / nr, ps[1] / nr, ps[2] / nr, ps[3] / nr, ps[4] / nr, ps[5] / nr, ps[6] / nr);
    }
    const double nst = (double)((N + C::NSC - 1) / C::NSC) * ((B + 511) / 512 > 0 ? 1 : 1);
    {
      static unsigned long long wg[8192][2];
      CK(hipMemcpyFromSymbol(wg, HIP_SYMBOL(g_wgclk), sizeof(wg)));
      std::vector<double> clk;
      for (int i = 0; i < std::min(kGrid, 8192); ++i) if (wg[i][1]) clk.push_back((double)wg[i][0] / (double)wg[i][1] * 0.1);
      if (!clk.empty()) {
        std::sort(clk.begin(), clk.end());
        printf("  in-kernel clock of the ring loop, per workgroup (stamped launch, right behind %d timed launches = %.2f s back to back): median %.3f GHz, min %.3f, max %.3f over %zu workgroups\n",
               reps, reps * ms * 1e-3, clk[clk.size() / 2], clk.front(), clk.back(), clk.size());
      }
    }
    for (int w = 0; w < 4; ++w)
      if (stp[w][7]) printf("  wave %d: ring loop %.0f cycles per regressor = %.1f per k-step, in-kernel clock %.3f GHz (%llu samples)\n", w,
             (double)stp[w][5] / stp[w][7], (double)stp[w][5] / stp[w][7] / (N / 4), (double)stp[w][5] / stp[w][6] * 0.1, stp[w][7]);
    for (int w = 0; w < 4; ++w)
      if (stp[w][0] + stp[w][3]) printf("  wave %d of WG 0, cycles per stage: DMA wait %7.0f | barrier %7.0f | first frags + DMA issue %7.0f | 8 k-steps %7.0f | loop %5.0f\n", w,
             stp[w][0] / nst, stp[w][1] / nst, stp[w][2] / nst, stp[w][3] / nst, stp[w][4] / nst);
  }
#endif
  std::vector<double> lp(B); std::vector<int32_t> info(B);
  CK(hipMemcpy(lp.data(), dlp, (size_t)B * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(info.data(), dinfo, (size_t)B * 4, hipMemcpyDeviceToHost));
  int bad = 0; double worst = 0;
  for (int r = 0; r < std::min(BU, 3); ++r) {
    std::vector<T> Xr(X.begin() + (size_t)r * N * D, X.begin() + (size_t)(r + 1) * N * D), yr(y.begin() + (size_t)r * N, y.begin() + (size_t)(r + 1) * N);
    const double ref = host_logpdf(Xr, yr, s, diag, D, N);
    const double rel = std::fabs(lp[r] - ref) / std::fabs(ref);
    worst = std::max(worst, rel);
  }
  for (int r = 0; r < B; ++r) { if (info[r] != 0) ++bad; if (!shareX && lp[r] != lp[r % BU]) ++bad; }
  const double flops = (double)D * (D + 1) * N + 4.0 * D * N + (double)D * D * D / 3 + 3.0 * D * D + 5.0 * N;
#ifdef BLR_WAVE_CLK
  {
    unsigned long long wc[2];
    CK(hipMemcpyFromSymbol(wc, HIP_SYMBOL(blr::g_waveclk), sizeof(wc)));
    if (wc[1]) printf("  in-kernel clock: %.3f GHz (shader cycles / 100 MHz ticks, summed over every 64th workgroup and all launches)\n", (double)wc[0] / (double)wc[1] * 0.1);
  }
#endif
  printf("%s D=%d N=%d B=%d %s: %.3f ms/launch  %.3f M updates/s  %.1f TFLOP/s (algorithmic)  logpdf rel err vs host %.2e  bad=%d  [EXP=%d]\n",
         sizeof(T) == 8 ? "f64" : "f32", D, N, B, diag ? (shareX ? "diag sharedX" : "diag") : (shareX ? "iso sharedX" : "iso"), ms, B / ms / 1e3, flops * B / ms / 1e9, worst, bad, BLR_EXP);
  return bad != 0 || !(worst < (sizeof(T) == 8 ? 1e-10 : 1e-3) || zero_ops);
}
