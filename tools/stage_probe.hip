// Microbenchmark: cycles of ONE pinned stage body of the fused kernel (stage8_prologue + stage8_body: 8 k-steps = 72 MFMAs
// + the column-vector work of the 2 k-steps a wave owns), LDS image prefilled, no DMA, no barrier.  8 x 576 = 4608 cycles is
// the matrix-pipe floor.   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DBLR_EXP=3] -I../bayesianlinearregressors.jl_amd/csrc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "blr_fused_small.hpp"
using namespace blr;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
struct Stamp { unsigned long long cyc, rt; };

template <bool ISO, int WPS, int NPIECE, int NBAR = 0>
__global__ __launch_bounds__(256, 2) void k_stage(double* out, Stamp* st, int iters, const double* gsrc, size_t gmask) {
  using T = double;
  using C = SmallCfg<T, 8>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* slot = reinterpret_cast<T*>(smem);
  T* ybuf = reinterpret_cast<T*>(smem + C::OFF_Y);
  T* wbuf = reinterpret_cast<T*>(smem + C::OFF_W);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < C::SLOT; i += 256) slot[i] = 1.0 + 1e-6 * (i % 977);
  if (tid < 2 * C::NSC) { ybuf[tid] = 0.5 + tid; wbuf[tid] = 1.0 + 0.01 * tid; }
  __syncthreads();
  typename Mfma<T>::acc4 acc[9];
  for (int i = 0; i < 9; ++i) acc[i] = typename Mfma<T>::acc4{0, 0, 0, 0};
  double bacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double qacc = 0;
  T mwf[8];
  for (int i = 0; i < 8; ++i) mwf[i] = 0.001 * (lane + i);
  unsigned long long c0 = 0, r0 = 0, c1 = 0, r1 = 0;
  auto run = [&](auto ws) {
    constexpr int WS = decltype(ws)::value;
    c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
    unsigned slot2 = lds_addr_of(slot + C::SLOT);
    asm volatile("" : "+v"(slot2));
    for (int it = 0; it < iters; ++it) {
      KF8<T, WS> f0;
      stage8_prologue<T, WS, ISO>(f0, slot, ybuf, wbuf, lane);
#pragma unroll
      for (int p = 0; p < NPIECE; ++p) {
        // a 1 KiB piece shaped like the kernel's: 4 columns x 256 B, streaming through a buffer of gmask + 1 bytes
        const size_t off = (((size_t)blockIdx.x * 4 + WS) * 1048576 + ((size_t)it * NPIECE + p) * 4096) & gmask;
        glds_s<16>(uni((int64_t)(uintptr_t)gsrc + (int64_t)off), (unsigned)((lane >> 4) * 1024 + (lane & 15) * 16), slot2 + (unsigned)((p * 4 + WS) * 1024));
      }
      stage8_body<T, WS, ISO>(f0, slot, ybuf, wbuf, acc, bacc, qacc, mwf, T(10), lane, T(1));
      if (NPIECE > 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      if (NBAR >= 1) __syncthreads();
    }
    c1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
  };
  switch (wave) {
    case 0: run(std::integral_constant<int, 0>{}); break;
    case 1: run(std::integral_constant<int, 1>{}); break;
    case 2: run(std::integral_constant<int, 2>{}); break;
    default: run(std::integral_constant<int, 3>{}); break;
  }
  double s = qacc;
  for (int i = 0; i < 9; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += bacc[i];
  out[blockIdx.x * 256 + tid] = s;
  if (lane == 0) { st[blockIdx.x * 4 + wave].cyc = c1 - c0; st[blockIdx.x * 4 + wave].rt = r1 - r0; }
}

template <bool ISO, int WPS, int NPIECE, int NBAR = 0>
int run(const char* label, int cus, int iters, double* buf, Stamp* st, const double* gsrc, size_t gmask) {
  using C = SmallCfg<double, 8>;
  auto kern = k_stage<ISO, WPS, NPIECE, NBAR>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
  const int grid = cus * WPS;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    kern<<<grid, 256, C::LDS_BYTES>>>(buf, st, iters, gsrc, gmask);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  std::vector<Stamp> h(grid * 4);
  CK(hipMemcpy(h.data(), st, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
  double cw[4] = {0, 0, 0, 0};
  for (size_t i = 0; i < h.size(); ++i) cw[i & 3] += (double)h[i].cyc;
  printf("%-28s %d WG/CU: %7.3f ms  cycles/stage by wave: %7.0f %7.0f %7.0f %7.0f  (floor 4608)  %6.1f TFLOP/s  [EXP=%d]\n", label, WPS, ms,
         cw[0] / grid / iters, cw[1] / grid / iters, cw[2] / grid / iters, cw[3] / grid / iters,
         (double)grid * 4 * iters * 72 * 2048.0 / ms / 1e9, BLR_EXP);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return 0;
}

int main(int argc, char** argv) {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  double* buf; Stamp* st;
  CK(hipMalloc((void**)&buf, (size_t)cus * 2 * 256 * 8));
  CK(hipMalloc((void**)&st, (size_t)cus * 2 * 4 * sizeof(Stamp)));
  const int iters = argc > 1 ? atoi(argv[1]) : 500;
  double* gsrc;
  const size_t gbytes = (size_t)4 << 30;
  CK(hipMalloc((void**)&gsrc, gbytes));
  CK(hipMemset(gsrc, 0, gbytes));
  run<true, 1, 0>("iso, no DMA", cus, iters, buf, st, gsrc, gbytes - 1);
  run<true, 2, 0>("iso, no DMA", cus, iters, buf, st, gsrc, gbytes - 1);
  run<true, 1, 4>("iso, 4 pieces/stage (HBM)", cus, iters, buf, st, gsrc, gbytes - 1);
  run<true, 2, 4>("iso, 4 pieces/stage (HBM)", cus, iters, buf, st, gsrc, gbytes - 1);
  run<true, 1, 8>("iso, 8 pieces/stage (HBM)", cus, iters, buf, st, gsrc, gbytes - 1);
  run<true, 2, 8>("iso, 8 pieces/stage (HBM)", cus, iters, buf, st, gsrc, gbytes - 1);
  run<true, 1, 8, 1>("iso, 8 pieces (HBM) + barrier/stage", cus, iters, buf, st, gsrc, gbytes - 1);
  run<true, 2, 8, 1>("iso, 8 pieces (HBM) + barrier/stage", cus, iters, buf, st, gsrc, gbytes - 1);
  run<true, 1, 0, 1>("iso, no DMA + barrier/stage", cus, iters, buf, st, gsrc, gbytes - 1);
  run<true, 2, 0, 1>("iso, no DMA + barrier/stage", cus, iters, buf, st, gsrc, gbytes - 1);
  run<true, 1, 8>("iso, 8 pieces/stage (L2, 1 MiB)", cus, iters, buf, st, gsrc, ((size_t)1 << 20) - 1);
  run<true, 2, 8>("iso, 8 pieces/stage (L2, 1 MiB)", cus, iters, buf, st, gsrc, ((size_t)1 << 20) - 1);
  return 0;
}
