// Microbenchmark: issue rate of v_fmac_f32 with a DPP row_newbcast source against the plain instruction, one wave per SIMD
// (the chain wave of panel_chain_kernel is alone on its pipeline most of the time).
//   hipcc --offload-arch=gfx950 -O3 dpp_fmac32.hip -o dpp_fmac32
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define F_DPP(ACC, A, B, K) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf" : "+v"(ACC) : "v"(A), "v"(B))
#define F_PLAIN(ACC, A, B) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(ACC) : "v"(A), "v"(B))

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, unsigned long long* cyc) {
  float acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float a = a0 + threadIdx.x * 1e-6f, b = 1.f - threadIdx.x * 1e-6f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {  // 16 independent DPP multiply-adds
      F_DPP(acc[0], a, b, 0); F_DPP(acc[1], a, b, 1); F_DPP(acc[2], a, b, 2); F_DPP(acc[3], a, b, 3);
      F_DPP(acc[4], a, b, 4); F_DPP(acc[5], a, b, 5); F_DPP(acc[6], a, b, 6); F_DPP(acc[7], a, b, 7);
      F_DPP(acc[8], a, b, 8); F_DPP(acc[9], a, b, 9); F_DPP(acc[10], a, b, 10); F_DPP(acc[11], a, b, 11);
      F_DPP(acc[12], a, b, 12); F_DPP(acc[13], a, b, 13); F_DPP(acc[14], a, b, 14); F_DPP(acc[15], a, b, 15);
    } else if (MODE == 1) {  // 16 independent plain multiply-adds
      for (int i = 0; i < 16; ++i) F_PLAIN(acc[i], a, b);
    } else if (MODE == 2) {  // a dependent chain of plain multiply-adds
      for (int i = 0; i < 16; ++i) F_PLAIN(acc[0], acc[0], b);
    } else {  // a dependent chain through DPP (with its two wait states)
      for (int i = 0; i < 16; ++i) asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc[0]) : "v"(b));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
  float* buf; unsigned long long* cyc;
  CK(hipMalloc((void**)&buf, 1 << 20)); CK(hipMalloc((void**)&cyc, 8));
  const int iters = 2000;
  const char* names[4] = {"16 independent v_fmac_f32_dpp", "16 independent v_fmac_f32", "dependent v_fmac_f32 chain", "dependent s_nop 1 + v_fmac_f32_dpp chain"};
  for (int w = 1; w <= 2; ++w)
    for (int m = 0; m < 4; ++m) {
      for (int rep = 0; rep < 2; ++rep) {
        if (m == 0) k<0><<<1, 256 * w>>>(buf, iters, 1.f, cyc);
        if (m == 1) k<1><<<1, 256 * w>>>(buf, iters, 1.f, cyc);
        if (m == 2) k<2><<<1, 256 * w>>>(buf, iters, 1.f, cyc);
        if (m == 3) k<3><<<1, 256 * w>>>(buf, iters, 1.f, cyc);
        CK(hipDeviceSynchronize());
      }
      unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
      printf("%d wave(s)/SIMD  %-42s %.2f cycles per instruction\n", w, names[m], (double)c / (iters * 16.0));
    }
  return 0;
}
