// Microbenchmark: what does one k-step of the fused kernel's Gram loop cost a wave, piece by piece?
// A k-step = fragment reads from LDS for the NEXT k-step (ds_read) + 9 v_mfma_f64_16x16x4 on the CURRENT fragments.
// Variants isolate: operand diversity (distinct A/B registers per MFMA), the LDS reads (none / all in front / spread
// between the MFMAs; b64 vs read2st64), waves per SIMD.  Cycles per k-step per wave from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 tools/kstep_probe.hip -o tools/kstep_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
struct Stamp { unsigned long long cyc, rt; };

#define MFMA(ACC, A, B) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))

// MODE 0: same A/B registers, no LDS        MODE 1: distinct registers (8 fragments), no LDS
// MODE 2: + 8 ds_read_b64 in front of the MFMAs (pinned)   MODE 3: the 8 reads spread, one after each MFMA
// MODE 4: reads in front, but only 4 (ds_read2st64_b64 x 4 = 8 fragments)  MODE 5: 8 reads AFTER the MFMAs of the group
template <int MODE, int WPS>
__global__ __launch_bounds__(256, WPS) void k_kstep(double* out, Stamp* st, int iters) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8 * 8 * 64 * 2; i += 256) lds[i] = 1.0 + 1e-9 * i;
  __syncthreads();
  d4 acc[9];
  for (int i = 0; i < 9; ++i) acc[i] = d4{0, 0, 0, 0};
  const double* base = lds + lane;
  double f0[8], f1[8];
  for (int i = 0; i < 8; ++i) { f0[i] = base[i * 64]; f1[i] = base[512 + i * 64]; }
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      // ---- k-step j on f0, reading f1 for j+1
      if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) f1[i] = base[(j + 1) * 512 + i * 64];
      }
      if (MODE == 4) {
#pragma unroll
        for (int i = 0; i < 8; ++i) f1[i] = base[(j + 1) * 512 + i * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        if (MODE == 0) MFMA(acc[i], f0[0], f0[1]);
        else MFMA(acc[i], f0[i < 3 ? 2 : 7], f0[i < 3 ? i : i - 3]);
        if (MODE == 3 && i < 8) { f1[i] = base[(j + 1) * 512 + i * 64]; __builtin_amdgcn_sched_barrier(0); }
      }
      if (MODE == 5) {
#pragma unroll
        for (int i = 0; i < 8; ++i) f1[i] = base[(j + 1) * 512 + i * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- k-step j+1 on f1, reading f0 for j+2
      if (MODE == 2 || MODE == 4) {
#pragma unroll
        for (int i = 0; i < 8; ++i) f0[i] = base[((j + 2) & 7) * 512 + i * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        if (MODE == 0) MFMA(acc[i], f1[0], f1[1]);
        else MFMA(acc[i], f1[i < 3 ? 2 : 7], f1[i < 3 ? i : i - 3]);
        if (MODE == 3 && i < 8) { f0[i] = base[((j + 2) & 7) * 512 + i * 64]; __builtin_amdgcn_sched_barrier(0); }
      }
      if (MODE == 5) {
#pragma unroll
        for (int i = 0; i < 8; ++i) f0[i] = base[((j + 2) & 7) * 512 + i * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < 9; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + f0[0] + f1[0];
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    st[w].cyc = c1 - c0; st[w].rt = r1 - r0;
  }
}

template <int MODE, int WPS>
int run(const char* label, int cus, int iters, double* buf, Stamp* st) {
  const int grid = cus * WPS;
  const size_t lds = 8 * 8 * 64 * 2 * sizeof(double);  // 64 KB: two slots of 8 k-steps
  auto kern = k_kstep<MODE, WPS>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    kern<<<grid, 256, lds>>>(buf, st, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  const int nw = grid * 4;
  std::vector<Stamp> h(nw);
  CK(hipMemcpy(h.data(), st, nw * sizeof(Stamp), hipMemcpyDeviceToHost));
  double c = 0, r = 0;
  for (auto& s : h) { c += (double)s.cyc; r += (double)s.rt; }
  printf("%-46s %d wave/SIMD: %8.3f ms  %7.1f cycles/k-step/wave (9 MFMA = 576)  clock %.2f GHz  %6.1f TFLOP/s\n", label, WPS, ms,
         c / nw / ((double)iters * 8), c / r * 0.1, (double)nw * iters * 8 * 9 * 2048.0 / ms / 1e9);
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
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
#define BOTH(M, L) run<M, 1>(L, cus, iters, buf, st); run<M, 2>(L, cus, iters, buf, st);
  BOTH(0, "same A/B registers, no LDS reads");
  BOTH(1, "distinct A/B registers, no LDS reads");
  BOTH(2, "+ 8 ds_read_b64 in front of the 9 MFMAs");
  BOTH(3, "+ 8 ds_read_b64 spread, one behind each MFMA");
  BOTH(4, "+ reads in front, compiler may pair (read2)");
  BOTH(5, "+ 8 ds_read_b64 behind the 9 MFMAs");
  return 0;
}
