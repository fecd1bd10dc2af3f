// Microbenchmark: v_fmac_f64 with DPP row_newbcast (lane k of each 16-lane row broadcast as src0) -- the
// register-only operand-sharing idiom for an f64 SYRK on the vector ALU.  Checks correctness and rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#define FMAC_BCAST(ACC, A, B, K) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf" : "+v"(ACC) : "v"(A), "v"(B))

__global__ __launch_bounds__(256) void k_check(double* out) {
  double a = (double)(threadIdx.x & 63) + 1.0, b = 2.0, acc = 0.0;
  FMAC_BCAST(acc, a, b, 5);  // expect (16*(lane/16) + 5 + 1) * 2
  out[threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_rate(double* out, int iters, double a0) {
  double acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = 0.0;
  double a = a0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
    FMAC_BCAST(acc[0], a, b, 0); FMAC_BCAST(acc[1], a, b, 1); FMAC_BCAST(acc[2], a, b, 2); FMAC_BCAST(acc[3], a, b, 3);
    FMAC_BCAST(acc[4], a, b, 4); FMAC_BCAST(acc[5], a, b, 5); FMAC_BCAST(acc[6], a, b, 6); FMAC_BCAST(acc[7], a, b, 7);
    FMAC_BCAST(acc[8], a, b, 8); FMAC_BCAST(acc[9], a, b, 9); FMAC_BCAST(acc[10], a, b, 10); FMAC_BCAST(acc[11], a, b, 11);
    FMAC_BCAST(acc[12], a, b, 12); FMAC_BCAST(acc[13], a, b, 13); FMAC_BCAST(acc[14], a, b, 14); FMAC_BCAST(acc[15], a, b, 15);
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount;
  double* buf;
  CK(hipMalloc((void**)&buf, (size_t)cus * 8 * 256 * 8));
  k_check<<<1, 256>>>(buf);
  std::vector<double> h(64);
  CK(hipMemcpy(h.data(), buf, 64 * 8, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int l = 0; l < 64; ++l) if (h[l] != (16 * (l / 16) + 5 + 1) * 2.0) ++bad;
  printf("row_newbcast check: %s (lane0=%g lane17=%g lane63=%g)\n", bad ? "WRONG" : "ok", h[0], h[17], h[63]);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 20000;
  for (int w = 1; w <= 4; w *= 2) {
    int grid = cus * w;
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      k_rate<<<grid, 256>>>(buf, iters, 1.0);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("v_fmac_f64_dpp row_newbcast, %d wave/SIMD: %.3f ms  %.1f TFLOP/s\n", w, ms, (double)grid * 256 * iters * 16 * 2.0 / ms / 1e9);
  }
  return 0;
}
