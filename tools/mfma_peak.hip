// Microbenchmark: sustained v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32 rate on this device
// (operands in registers, NACC independent accumulators per wave, W waves per SIMD), with the in-kernel
// shader clock (s_memtime / s_memrealtime).  Calibrates the roofline peak bench.py prices against.
//   hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Stamp { unsigned long long cyc, rt; };

template <int NACC>
__global__ __launch_bounds__(256) void k_f64(double* out, Stamp* st, int iters, double a0, double b0) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { st[blockIdx.x].cyc = c1 - c0; st[blockIdx.x].rt = r1 - r0; }
}
template <int NACC>
__global__ __launch_bounds__(256) void k_f32(float* out, Stamp* st, int iters, float a0, float b0) {
  f4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f4{0, 0, 0, 0};
  float a = a0 + threadIdx.x * 1e-6f, b = b0 - threadIdx.x * 1e-6f;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { st[blockIdx.x].cyc = c1 - c0; st[blockIdx.x].rt = r1 - r0; }
}
// the same f32 loop on RANDOM operands (a different pair per accumulator and lane): the clock the part holds under a matrix
// load depends on how many datapath bits toggle (DVFS); near-constant operands flatter it
template <int NACC>
__global__ __launch_bounds__(256) void k_f32_random(float* out, Stamp* st, int iters) {
  f4 acc[NACC];
  float ra[NACC], rb[NACC];
  for (int i = 0; i < NACC; ++i) {
    acc[i] = f4{0, 0, 0, 0};
    unsigned long long z = ((unsigned long long)(blockIdx.x * blockDim.x + threadIdx.x) * NACC + i + 1) * 0x9E3779B97F4A7C15ull;
    z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
    ra[i] = (float)((double)(long long)(z >> 11) * (1.0 / 4503599627370496.0) - 1.0);
    z *= 0x94D049BB133111EBull; z ^= z >> 31;
    rb[i] = (float)((double)(long long)(z >> 11) * (1.0 / 4503599627370496.0) - 1.0);
  }
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(ra[i]), "v"(rb[i]));
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { st[blockIdx.x].cyc = c1 - c0; st[blockIdx.x].rt = r1 - r0; }
}
__global__ __launch_bounds__(256) void k_fma64(double* out, Stamp* st, int iters, double a0) {
  double x[16];
  for (int i = 0; i < 16; ++i) x[i] = a0 + i + threadIdx.x;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = __builtin_fma(x[i], 1.0000001, 0.5);
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { st[blockIdx.x].cyc = c1 - c0; st[blockIdx.x].rt = r1 - r0; }
}

static double clock_ghz(Stamp* dst, int grid) {
  std::vector<Stamp> h(grid);
  hipMemcpy(h.data(), dst, grid * sizeof(Stamp), hipMemcpyDeviceToHost);
  double c = 0, r = 0;
  for (auto& s : h) { c += (double)s.cyc; r += (double)s.rt; }
  return c / r * 0.1;  // s_memrealtime ticks at 100 MHz
}

int main(int argc, char** argv) {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount;
  printf("device %s, %d CUs, nominal clock %d kHz\n", p.gcnArchName, cus, p.clockRate);
  void* buf; Stamp* st;
  CK(hipMalloc(&buf, (size_t)cus * 8 * 256 * 8));
  CK(hipMalloc((void**)&st, (size_t)cus * 8 * sizeof(Stamp)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  float ms = 0;
#define RUN(label, flops_per_thread_block_iter, launch)                                         \
  for (int rep = 0; rep < 3; ++rep) { CK(hipEventRecord(e0)); launch; CK(hipEventRecord(e1));   \
    CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); }                        \
  { double ghz = clock_ghz(st, grid);                                                          \
    printf("%-44s %8.3f ms %7.1f TFLOP/s  clock %.2f GHz\n", label, ms, (flops_per_thread_block_iter) / ms / 1e9, ghz); }
  for (int w = 1; w <= 4; ++w) {
    int grid = cus * w;
    char lab[128];
    snprintf(lab, sizeof lab, "f64 mfma 16x16x4, %d wave/SIMD, 9 acc", w);
    RUN(lab, (double)grid * 4 * iters * 9 * 2048.0, (k_f64<9><<<grid, 256>>>((double*)buf, st, iters, 1.0, 2.0)));
    snprintf(lab, sizeof lab, "f64 mfma 16x16x4, %d wave/SIMD, 4 acc", w);
    RUN(lab, (double)grid * 4 * iters * 4 * 2048.0, (k_f64<4><<<grid, 256>>>((double*)buf, st, iters, 1.0, 2.0)));
  }
  for (int w = 1; w <= 2; ++w) {
    int grid = cus * w;
    char lab[128];
    snprintf(lab, sizeof lab, "f32 mfma 16x16x4, %d wave/SIMD, 9 acc", w);
    RUN(lab, (double)grid * 4 * iters * 9 * 2048.0, (k_f32<9><<<grid, 256>>>((float*)buf, st, iters, 1.0f, 2.0f)));
    snprintf(lab, sizeof lab, "f32 mfma 16x16x4 RANDOM, %d wave/SIMD, 9 acc", w);
    RUN(lab, (double)grid * 4 * iters * 9 * 2048.0, (k_f32_random<9><<<grid, 256>>>((float*)buf, st, iters)));
  }
  for (int w = 1; w <= 8; w *= 2) {
    int grid = cus * w;
    char lab[128];
    snprintf(lab, sizeof lab, "f64 valu fma, %d wave/SIMD, 16 chains", w);
    RUN(lab, (double)grid * 256 * iters * 16 * 2.0, (k_fma64<<<grid, 256>>>((double*)buf, st, iters, 1.0)));
  }
  return 0;
}
