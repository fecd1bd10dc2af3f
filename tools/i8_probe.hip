// Feasibility probe for an int8-sliced (Ozaki-style) f64 Gram on gfx950:
//   (1) operand / result lane maps of v_mfma_i32_16x16x64_i8, checked with exact random integer data (asymmetric B);
//   (2) sustained rate of that instruction (independent accumulators, 1 and 2 waves per SIMD) with the in-kernel clock;
//   (3) cost of cutting an f64 into 7 signed 7-bit digits on the vector ALU (the slicing pass of such a kernel).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 i8_probe.hip -o i8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// ---- (1) layout: C[16][16] = A[16][64] * B[64][16], int8 in, int32 out ------------------------------------------------
__global__ void k_layout(const int8_t* A, const int8_t* B, int* C) {
  const int l = threadIdx.x;
  // hypothesis (bf16 map scaled to 16 bytes per lane): lane l holds A[l & 15][16 (l >> 4) + j], B[16 (l >> 4) + j][l & 15]
  i32x4 a, b;
  int8_t av[16], bv[16];
  for (int j = 0; j < 16; ++j) {
    av[j] = A[(l & 15) * 64 + 16 * (l >> 4) + j];
    bv[j] = B[(16 * (l >> 4) + j) * 16 + (l & 15)];
  }
  for (int w = 0; w < 4; ++w) {
    a[w] = (uint8_t)av[4 * w] | ((uint8_t)av[4 * w + 1] << 8) | ((uint8_t)av[4 * w + 2] << 16) | ((uint32_t)(uint8_t)av[4 * w + 3] << 24);
    b[w] = (uint8_t)bv[4 * w] | ((uint8_t)bv[4 * w + 1] << 8) | ((uint8_t)bv[4 * w + 2] << 16) | ((uint32_t)(uint8_t)bv[4 * w + 3] << 24);
  }
  i32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
  // C/D map (dtype independent, f64 excepted): col = l & 15, row = 4 (l >> 4) + v
  for (int v = 0; v < 4; ++v) C[(4 * (l >> 4) + v) * 16 + (l & 15)] = c[v];
}

// ---- (2) rate ---------------------------------------------------------------------------------------------------------
struct Stamp { unsigned long long cyc, rt; };
template <int NACC>
__global__ __launch_bounds__(256) void k_rate(int* out, Stamp* st, int iters, int seed) {
  i32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = i32x4{0, 0, 0, 0};
  i32x4 a, b;
  unsigned x = seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x;
  for (int w = 0; w < 4; ++w) { x = x * 1664525u + 1013904223u; a[w] = (int)(x & 0x3f3f3f3f); x = x * 1664525u + 1013904223u; b[w] = (int)(x & 0x3f3f3f3f); }
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  int s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { st[blockIdx.x].cyc = c1 - c0; st[blockIdx.x].rt = r1 - r0; }
}

typedef int i32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k_rate32(int* out, Stamp* st, int iters, int seed) {
  i32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int v = 0; v < 16; ++v) acc[i][v] = 0;
  i32x4 a, b;
  unsigned x = seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x;
  for (int w = 0; w < 4; ++w) { x = x * 1664525u + 1013904223u; a[w] = (int)(x & 0x3f3f3f3f); x = x * 1664525u + 1013904223u; b[w] = (int)(x & 0x3f3f3f3f); }
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  int s = 0;
  for (int i = 0; i < NACC; ++i) for (int v = 0; v < 16; ++v) s += acc[i][v];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { st[blockIdx.x].cyc = c1 - c0; st[blockIdx.x].rt = r1 - r0; }
}

// ---- (3) slicing: t in [-0.5, 0.5] -> 7 digits in [-64, 64], packed bytes --------------------------------------------
__global__ __launch_bounds__(256) void k_slice(const double* X, unsigned* out, Stamp* st, int per_thread) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned acc0 = 0, acc1 = 0;
  const double x0 = X[tid];
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int e = 0; e < per_thread; ++e) {
    double r = (x0 + e * 1e-3) * 0.0625;  // stand-in for the row scale 2^-e (operand in registers: pure ALU cost)
    unsigned lo = 0, hi = 0;
#pragma unroll
    for (int s = 0; s < 7; ++s) {
      const double v = r * 128.0;
      const double q = __builtin_rint(v);
      r = v - q;
      const unsigned qb = (unsigned)(__double2int_rn(q)) & 0xffu;
      if (s < 4) lo |= qb << (8 * s); else hi |= qb << (8 * (s - 4));
    }
    acc0 ^= lo; acc1 += hi;
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[tid] = acc0 ^ acc1;
  if (threadIdx.x == 0) { st[blockIdx.x].cyc = c1 - c0; st[blockIdx.x].rt = r1 - r0; }
}

static double clock_ghz(Stamp* dst, int grid, double* cyc_avg) {
  std::vector<Stamp> h(grid);
  hipMemcpy(h.data(), dst, grid * sizeof(Stamp), hipMemcpyDeviceToHost);
  double c = 0, r = 0;
  for (auto& s : h) { c += (double)s.cyc; r += (double)s.rt; }
  if (cyc_avg) *cyc_avg = c / grid;
  return c / r * 0.1;
}

int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  printf("device %s, %d CUs\n", p.gcnArchName, cus);
  {  // (1)
    std::vector<int8_t> A(16 * 64), B(64 * 16);
    srand(1);
    for (auto& v : A) v = (int8_t)(rand() % 255 - 127);
    for (auto& v : B) v = (int8_t)(rand() % 255 - 127);
    int8_t *dA, *dB; int* dC;
    CK(hipMalloc((void**)&dA, A.size())); CK(hipMalloc((void**)&dB, B.size())); CK(hipMalloc((void**)&dC, 256 * 4));
    CK(hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    std::vector<int> C(256);
    CK(hipMemcpy(C.data(), dC, 256 * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
      int ref = 0;
      for (int k = 0; k < 64; ++k) ref += (int)A[i * 64 + k] * (int)B[k * 16 + j];
      if (ref != C[i * 16 + j]) ++bad;
    }
    printf("(1) layout hypothesis A[l&15][16(l>>4)+j], B[16(l>>4)+j][l&15], C row 4(l>>4)+v col l&15: %s (%d / 256 wrong)\n", bad ? "WRONG" : "exact", bad);
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  {  // (2)
    int* out; Stamp* st;
    CK(hipMalloc((void**)&out, (size_t)cus * 8 * 256 * 4)); CK(hipMalloc((void**)&st, (size_t)cus * 8 * sizeof(Stamp)));
    const int iters = 20000;
    for (int wps = 1; wps <= 2; ++wps) {
      const int grid = cus * wps;
      hipLaunchKernelGGL(k_rate<8>, dim3(grid), dim3(256), 0, 0, out, st, 2000, 1);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_rate<8>, dim3(grid), dim3(256), 0, 0, out, st, iters, 2);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      double cyc; const double ghz = clock_ghz(st, grid, &cyc);
      const double ops = (double)grid * 4 * iters * 8 * (2.0 * 16 * 16 * 64);
      printf("(2) i8 16x16x64, %d wave(s)/SIMD: %.1f TOP/s, clock %.2f GHz, %.1f cycles per MFMA per SIMD\n", wps, ops / ms * 1e-9, ghz,
             cyc / ((double)iters * 8) / 1.0 * (1.0 / wps) * wps);
    }
  }
  {  // (2b)
    int* out; Stamp* st;
    CK(hipMalloc((void**)&out, (size_t)cus * 8 * 256 * 4)); CK(hipMalloc((void**)&st, (size_t)cus * 8 * sizeof(Stamp)));
    const int iters = 20000;
    for (int wps = 1; wps <= 2; ++wps) {
      const int grid = cus * wps;
      hipLaunchKernelGGL(k_rate32<4>, dim3(grid), dim3(256), 0, 0, out, st, 2000, 1);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_rate32<4>, dim3(grid), dim3(256), 0, 0, out, st, iters, 2);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      double cyc; const double ghz = clock_ghz(st, grid, &cyc);
      const double ops = (double)grid * 4 * iters * 4 * (2.0 * 32 * 32 * 32);
      printf("(2b) i8 32x32x32, %d wave(s)/SIMD: %.1f TOP/s, clock %.2f GHz\n", wps, ops / ms * 1e-9, ghz);
    }
  }
  {  // (3)
    const int grid = cus * 8, per = 4096;
    const size_t n = (size_t)grid * 256;
    std::vector<double> X(n);
    for (size_t i = 0; i < n; ++i) X[i] = (double)rand() / RAND_MAX * 7.9 - 3.95;
    double* dX; unsigned* out; Stamp* st;
    CK(hipMalloc((void**)&dX, n * 8)); CK(hipMalloc((void**)&out, (size_t)grid * 256 * 4)); CK(hipMalloc((void**)&st, grid * sizeof(Stamp)));
    CK(hipMemcpy(dX, X.data(), n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_slice, dim3(grid), dim3(256), 0, 0, dX, out, st, per);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_slice, dim3(grid), dim3(256), 0, 0, dX, out, st, per);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double cyc; const double ghz = clock_ghz(st, grid, &cyc);
    printf("(3) slicing f64 -> 7 digits (8 waves/SIMD): %.1f cycles per element per wave, %.2f G elements/s chip-wide, clock %.2f GHz\n",
           cyc / per, (double)n * per / ms * 1e-6, ghz);
  }
  return 0;
}
