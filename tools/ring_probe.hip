// Microbenchmark: gram_iso_ring (the fused kernel's NB == 8 isotropic Gram loop) alone -- real LDS-DMA stream from HBM, real
// barriers, no other phase.  Cycles per k-step per wave by s_memtime; 576 = nine back-to-back v_mfma_f64_16x16x4.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DRING_NOBAR] [-DRING_NORETIRE] -I../bayesianlinearregressors.jl_amd/csrc ring_probe.hip
//   ./ring_probe N random(0|1) [sustain_seconds]
// sustain_seconds > 0 (MI355X_MICROARCH.md, DVFS give-back item 6): each configuration is launched back to back for that
// long BEFORE anything is read; the clock is the MEDIAN over workgroups of delta s_memtime / delta s_memrealtime x 100 MHz of
// the LAST launch, the TFLOP/s come from HIP events around the last quarter of the launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "blr_fused_small.hpp"
using namespace blr;
#ifndef RING_MWZ
#define RING_MWZ false
#endif
#ifndef RING_NH
#define RING_NH 4
#endif
#ifndef RING_WPS
#define RING_WPS 2
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
struct Stamp { unsigned long long cyc, rt; };

__global__ __launch_bounds__(256, RING_WPS) void k_ring(const double* X, const double* y, int N, int64_t strideX, double* out, Stamp* st) {
  using T = double;
  using C = SmallCfg<T, 8>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* ring = reinterpret_cast<T*>(smem);
  T* ybuf = reinterpret_cast<T*>(smem + (RING_NH == 4 ? C::OFF_Y : 49152));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  typename Mfma<T>::acc4 acc[9];
  for (int i = 0; i < 9; ++i) acc[i] = typename Mfma<T>::acc4{0, 0, 0, 0};
  double bacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double qacc = 0;
  T mwf[8];
  for (int i = 0; i < 8; ++i) mwf[i] = 0.001 * (lane + i);
  const BLR_GLOBAL T* Xg = as_global(X + (int64_t)blockIdx.x * strideX);
  const BLR_GLOBAL T* yg = as_global(y + (int64_t)blockIdx.x * N);
  const unsigned voff = glds_lane_offset<T>(128, lane);
  __syncthreads();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  switch (wave) {
    case 0: gram_iso_ring<T, 0, RING_MWZ, RING_NH>(ring, ybuf, Xg, yg, 128, N, voff, lane, acc, bacc, qacc, mwf, T(10)); break;
    case 1: gram_iso_ring<T, 1, RING_MWZ, RING_NH>(ring, ybuf, Xg, yg, 128, N, voff, lane, acc, bacc, qacc, mwf, T(10)); break;
    case 2: gram_iso_ring<T, 2, RING_MWZ, RING_NH>(ring, ybuf, Xg, yg, 128, N, voff, lane, acc, bacc, qacc, mwf, T(10)); break;
    default: gram_iso_ring<T, 3, RING_MWZ, RING_NH>(ring, ybuf, Xg, yg, 128, N, voff, lane, acc, bacc, qacc, mwf, T(10)); break;
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = qacc;
  for (int i = 0; i < 9; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += bacc[i];
  out[blockIdx.x * 256 + tid] = s;
  if (lane == 0) { st[blockIdx.x * 4 + wave].cyc = c1 - c0; st[blockIdx.x * 4 + wave].rt = r1 - r0; }
}

__global__ void fill_random(double* p, size_t n, unsigned long long seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long z = (i + seed) * 0x9E3779B97F4A7C15ull;
    z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
    p[i] = (double)(long long)(z >> 11) * (1.0 / 4503599627370496.0) - 1.0;  // uniform in (-1, 1): every mantissa bit toggles
  }
}

int main(int argc, char** argv) {
  using C = SmallCfg<double, 8>;
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  const int N = argc > 1 ? atoi(argv[1]) : 16384;
  const double sustain = argc > 3 ? atof(argv[3]) : 0.0;
  const int maxB = cus * RING_WPS;
  double *X, *y, *out; Stamp* st;
  CK(hipMalloc((void**)&X, (size_t)maxB * N * 128 * 8)); CK(hipMemset(X, 0, (size_t)maxB * N * 128 * 8));
  CK(hipMalloc((void**)&y, (size_t)maxB * N * 8)); CK(hipMemset(y, 0, (size_t)maxB * N * 8));
  if (argc > 2 && atoi(argv[2]) != 0) {  // random operands: the clock the part holds depends on the data (zeros toggle nothing)
    fill_random<<<4096, 256>>>(X, (size_t)maxB * N * 128, 12345ull);
    fill_random<<<1024, 256>>>(y, (size_t)maxB * N, 999ull);
    CK(hipDeviceSynchronize());
    printf("operands: random\n");
  }
  CK(hipMalloc((void**)&out, (size_t)maxB * 256 * 8));
  CK(hipMalloc((void**)&st, (size_t)maxB * 4 * sizeof(Stamp)));
  const int kLds = RING_NH == 4 ? C::LDS_BYTES : 49152 + 1024;  // NH = 3: 48 KB ring + y
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ring), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
  for (int share = 0; share < 2; ++share)
    for (int wps = 1; wps <= RING_WPS; ++wps) {
      const int grid = cus * wps;
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      float ms = 0;
      int launches = 3;
      if (sustain > 0) {
        // calibrate, then queue `sustain` seconds of back-to-back launches (nothing synchronises in between); the events
        // bracket the last quarter
        CK(hipEventRecord(e0));
        k_ring<<<grid, 256, kLds>>>(X, y, N, share ? 0 : (int64_t)N * 128, out, st);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        launches = (int)(sustain * 1e3 / ms) + 4;
        const int tail = launches / 4;
        for (int rep = 0; rep < launches; ++rep) {
          if (rep == launches - tail) CK(hipEventRecord(e0));
          k_ring<<<grid, 256, kLds>>>(X, y, N, share ? 0 : (int64_t)N * 128, out, st);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= tail;
      } else {
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipEventRecord(e0));
          k_ring<<<grid, 256, kLds>>>(X, y, N, share ? 0 : (int64_t)N * 128, out, st);
          CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
      }
      std::vector<Stamp> h(grid * 4);
      CK(hipMemcpy(h.data(), st, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
      double c = 0, r = 0;
      std::vector<double> clk;
      for (auto& s : h) { c += (double)s.cyc; r += (double)s.rt; clk.push_back((double)s.cyc / (double)s.rt * 0.1); }
      std::sort(clk.begin(), clk.end());
      printf("ring loop, N=%d, %d WG/CU, X %s: %8.3f ms  %7.1f cycles/k-step/wave  clock %.2f GHz (median over %zu waves %.3f, min %.3f, max %.3f)  %6.1f TFLOP/s (matrix)  %5.2f TB/s  [%d launches back to back%s]\n", N, wps,
             share ? "shared (cache)" : "streamed (HBM)", ms, c / h.size() / (N / 4), c / r * 0.1, clk.size(), clk[clk.size() / 2], clk.front(), clk.back(),
             (double)grid * 4 * (N / 4) * 9 * 2048.0 / ms / 1e9, share ? 0.0 : (double)grid * N * 1024.0 / ms / 1e9, launches,
             sustain > 0 ? ", timed over the last quarter" : "");
    }
  return 0;
}
