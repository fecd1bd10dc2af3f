// Diagnostic: where do the cycles of phase_chol (128 x 128 blocked Cholesky in LDS/registers) go?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DBLR_STAMPS -I. -I../bayesianlinearregressors.jl_amd/csrc chol_bench.hip -o chol_bench
//   ./chol_bench      phase_chol, f32 and f64: time, per-section cycles of wave 0, FNV fingerprint of L (bit-identity across versions)
//   ./chol_bench 2    fp64: phase_chol against the two experiments of blr_chol_dpp_experiment.hpp (time, sections, max |dL|, |du|, |W L - I|)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include <cstdlib>
#include "blr_fused_small.hpp"
#include "blr_chol_dpp_experiment.hpp"
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

constexpr int kOffW = (SmallCfg<double, 8>::LDS_BYTES + 15) & ~15, kOffU = kOffW + 16384, kLds2 = kOffU + 1024;
__global__ __launch_bounds__(256, 2) void k2(const double* A, double* out, double* uout, int* info, int reps, int which) {
  using C = SmallCfg<double, 8>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* P = reinterpret_cast<double*>(smem);
  double* bvec = reinterpret_cast<double*>(smem + C::OFF_B);
  const int tid = threadIdx.x;
  int rc = 0;
  for (int rep = 0; rep < reps; ++rep) {
    __syncthreads();
    for (int idx = tid; idx < 128 * 128; idx += 256) { int c = idx >> 7, r = idx & 127; if (r >= c) P[pidx(r, c)] = A[c * 128 + r]; }
    if (tid < 128) bvec[tid] = 1.0 + 0.01 * tid;
    __syncthreads();
    rc = which == 2 ? chol128_cw<kOffW, kOffU>(smem) : (which ? chol128_dpp<kOffW, kOffU>(smem) : phase_chol<double, 8>(smem, 128, 1));
  }
  if (tid == 0) info[blockIdx.x] = rc;
  for (int idx = tid; idx < 128 * 128; idx += 256) { int c = idx >> 7, r = idx & 127; if (r >= c) out[c * 128 + r] = P[pidx(r, c)]; }
  if (tid < 128) uout[tid] = bvec[tid];
  if (which) for (int idx = tid; idx < 2048; idx += 256) uout[128 + idx] = reinterpret_cast<double*>(smem + kOffW)[idx];
}

__global__ __launch_bounds__(512, 1) void k3(const double* A, double* out, double* uout, int* info, int reps) {
  using C = SmallCfg<double, 8>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* P = reinterpret_cast<double*>(smem);
  double* bvec = reinterpret_cast<double*>(smem + C::OFF_B);
  const int tid = threadIdx.x;
  int rc = 0;
  for (int rep = 0; rep < reps; ++rep) {
    __syncthreads();
    for (int idx = tid; idx < 128 * 128; idx += 512) { int c = idx >> 7, r = idx & 127; if (r >= c) P[pidx(r, c)] = A[c * 128 + r]; }
    if (tid < 128) bvec[tid] = 1.0 + 0.01 * tid;
    __syncthreads();
    rc = chol128_cw<kOffW, kOffU, 8>(smem);
  }
  if (tid == 0) info[blockIdx.x] = rc;
  for (int idx = tid; idx < 128 * 128; idx += 512) { int c = idx >> 7, r = idx & 127; if (r >= c) out[c * 128 + r] = P[pidx(r, c)]; }
  if (tid < 128) uout[tid] = bvec[tid];
  for (int idx = tid; idx < 2048; idx += 512) uout[128 + idx] = reinterpret_cast<double*>(smem + kOffW)[idx];
}

int run2() {
  std::vector<double> A(128 * 128);
  for (int c = 0; c < 128; ++c) for (int r = 0; r < 128; ++r) A[c * 128 + r] = ((r == c ? 200.0 : 0.0) + std::cos(0.37 * (r + 1) * (c + 1)));
  for (int c = 0; c < 128; ++c) for (int r = 0; r < c; ++r) A[c * 128 + r] = A[r * 128 + c];
  double *dA, *dO, *dU; int* dI;
  CK(hipMalloc((void**)&dA, A.size() * 8)); CK(hipMalloc((void**)&dO, A.size() * 8)); CK(hipMalloc((void**)&dU, (128 + 2048) * 8)); CK(hipMalloc((void**)&dI, 4096));
  CK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, kLds2));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 50;
  unsigned long long zero[8] = {0};
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k3), hipFuncAttributeMaxDynamicSharedMemorySize, kLds2));
  std::vector<double> L[4], U[4];
  for (int which = 0; which < 4; ++which) {
    for (int grid : {1, 256}) {
      float ms;
      if (which == 3) k3<<<grid, 512, kLds2>>>(dA, dO, dU, dI, 2); else k2<<<grid, 256, kLds2>>>(dA, dO, dU, dI, 2, which);
      CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), zero, sizeof(zero)));
      CK(hipEventRecord(e0));
      if (which == 3) k3<<<grid, 512, kLds2>>>(dA, dO, dU, dI, reps); else k2<<<grid, 256, kLds2>>>(dA, dO, dU, dI, reps, which);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long st[8];
      CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st)));
      int info; CK(hipMemcpy(&info, dI, 4, hipMemcpyDeviceToHost));
      printf("%s grid=%d: %.2f us per factorisation (info=%d)", which == 3 ? "chol128_cw8" : (which == 2 ? "chol128_cw " : (which ? "chol128_dpp" : "phase_chol ")), grid, ms * 1e3 / reps, info);
      if (grid == 1) {
        if (which >= 2) printf("  chain wave, cycles/fact: first tile %llu | wait B2 %llu | solve row J+1 + update its diagonal tile %llu | wait B3 %llu | factor + invert + stores %llu",
               st[1] / reps, st[2] / reps, st[3] / reps, st[4] / reps, st[5] / reps);
        else if (which) printf("  cycles/fact: load-tiles %llu | barrier %llu | diag tile factor + inverse %llu | u_J, stores, solves %llu | barrier %llu | r update + trailing %llu",
               st[0] / reps, st[1] / reps, st[2] / reps, st[3] / reps, st[4] / reps, st[5] / reps);
        else printf("  cycles/fact: load-tiles %llu | barrier %llu | (a) store panel %llu | (b) eliminate %llu | writeback+barrier %llu | (c) trailing %llu",
               st[0] / reps, st[1] / reps, st[2] / reps, st[3] / reps, st[4] / reps, st[5] / reps);
      }
      printf("\n");
    }
    L[which].resize(128 * 128); U[which].resize(128 + 2048);
    CK(hipMemcpy(L[which].data(), dO, 128 * 128 * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(U[which].data(), dU, (128 + 2048) * 8, hipMemcpyDeviceToHost));
  }
  for (int w = 1; w < 4; ++w) {
  double dl = 0, du = 0, ml = 0, mu = 0;
  for (int c = 0; c < 128; ++c) for (int r = c; r < 128; ++r) { dl = fmax(dl, fabs(L[0][c * 128 + r] - L[w][c * 128 + r])); ml = fmax(ml, fabs(L[0][c * 128 + r])); }
  for (int i = 0; i < 128; ++i) { du = fmax(du, fabs(U[0][i] - U[w][i])); mu = fmax(mu, fabs(U[0][i])); }
  double dw = 0;  // W_J L_JJ = I ?
  for (int J = 0; J < 8; ++J) for (int i = 0; i < 16; ++i) for (int c = 0; c < 16; ++c) {
    double s = 0;
    for (int k = 0; k < 16; ++k) { const int rr = 16 * J + k, cc = 16 * J + c; s += U[w][128 + (16 * J + i) * 16 + k] * (rr >= cc ? L[w][cc * 128 + rr] : 0.0); }
    dw = fmax(dw, fabs(s - (i == c ? 1.0 : 0.0)));
  }
  printf("%s vs phase_chol: max |dL| %.3e (max |L| %.3e) | max |du| %.3e (max |u| %.3e) | max |W_J L_JJ - I| %.3e\n", w == 3 ? "chol128_cw8" : (w == 2 ? "chol128_cw " : "chol128_dpp"), dl, ml, du, mu, dw);
  }
  return 0;
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
    std::vector<T> O(128 * 128);
    CK(hipMemcpy(O.data(), dO, O.size() * sizeof(T), hipMemcpyDeviceToHost));
    unsigned long long hsh = 1469598103934665603ull;  // FNV-1a over the lower triangle: a bit-for-bit fingerprint of L
    for (int c = 0; c < 128; ++c) for (int r = c; r < 128; ++r) { const unsigned char* b = reinterpret_cast<const unsigned char*>(&O[c * 128 + r]); for (size_t q = 0; q < sizeof(T); ++q) { hsh ^= b[q]; hsh *= 1099511628211ull; } }
    if (const char* dp = getenv("CHOL_DUMP")) { if (grid == 1) { char fn[256]; snprintf(fn, sizeof fn, "%s_%s.bin", dp, name); FILE* f = fopen(fn, "wb"); fwrite(O.data(), sizeof(T), O.size(), f); fclose(f); } }
    printf("%s grid=%d: %.2f us per factorisation (info=%d, L fingerprint %016llx)", name, grid, ms * 1e3 / reps, info, hsh);
    if (grid == 1) {
      printf("  cycles/fact: load-tiles %llu | barrier %llu | (a) store panel %llu | (b) eliminate %llu | writeback+barrier %llu | (c) trailing %llu",
             st[0] / reps, st[1] / reps, st[2] / reps, st[3] / reps, st[4] / reps, st[5] / reps);
    }
    printf("\n");
  }
  return 0;
}
int main(int argc, char** argv) { if (argc > 1) return run2(); run<float>("f32"); run<double>("f64"); return 0; }
