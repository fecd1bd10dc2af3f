// Microbenchmark: bf16 matrix instructions of gfx950 next to the f32 one the large-D Gram uses (rates per CU; operands resident).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 bf16_probe.hip -o bf16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f4 acc[8];
  f16v big[2];
  for (int i = 0; i < 8; ++i) acc[i] = f4{0, 0, 0, 0};
  for (int i = 0; i < 2; ++i) for (int v = 0; v < 16; ++v) big[i][v] = 0;
  const float x = 1.0f + threadIdx.x * 1e-3f;
  s4 a4 = {(short)threadIdx.x, 1, 2, 3}, b4 = {3, 2, 1, (short)threadIdx.x};
  bf8 a8, b8;
  for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(x + i); b8[i] = (__bf16)(x - i); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, acc[i], 0, 0, 0);
      if constexpr (KIND == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i], 0, 0, 0);
      if constexpr (KIND == 2) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i], 0, 0, 0);
      if constexpr (KIND == 3) big[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, big[i & 1], 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0];
  s += big[0][0] + big[1][0];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
int run(const char* name, double flops_per_instr) {
  int cus = 256;
  float* buf;
  CK(hipMalloc((void**)&buf, cus * 8 * 256 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 20000;
  for (int w = 1; w <= 2; ++w) {
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      k<KIND><<<cus * w, 256>>>(buf, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double instr = (double)cus * w * 4 * iters * 8;
    printf("%-28s %d wave/SIMD: %.3f ms  %.1f TFLOP/s  (%.1f cycles per instruction per SIMD at 2.4 GHz)\n", name, w, ms, instr * flops_per_instr / ms / 1e9,
           ms * 1e-3 * 2.4e9 / (iters * 8.0 * w));
  }
  return 0;
}
int main() {
  run<0>("f32 16x16x4", 2048);
  run<1>("bf16 16x16x16 (_1k)", 8192);
  run<2>("bf16 16x16x32", 16384);
  run<3>("bf16 32x32x16", 32768);
  return 0;
}
