// Microbenchmark: what paces v_mfma_f64_16x16x4_f64 on gfx950?  tools/mfma_peak.hip measured 105 cycles per MFMA per SIMD at
// >= 2 waves/SIMD (141 at one wave) against the 64 the 78.6 TF spec implies.  This probe separates the candidates:
//   - chip-level throttling (power / di-dt): one wave alone on the chip vs one CU vs every CU;
//   - register-file placement: accumulators in VGPRs vs AGPRs, A/B operands in AGPRs;
//   - issue pacing: s_nop padding between MFMAs (an inherent issue interval absorbs the padding);
//   - operand data (zeros vs random-ish), dependent chain latency, the 4x4x4 (4-block) shape;
//   - co-issue with f32 / f64 vector FMAs.
// Per configuration: shader cycles per MFMA per wave (s_memtime), in-kernel clock (s_memtime / s_memrealtime), TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_probe.hip -o tools/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Stamp { unsigned long long cyc, rt; };
constexpr int NACC = 8;

enum { M_VGPR = 0, M_AGPR_ACC = 1, M_NOP = 2, M_ZERO = 3, M_4X4 = 4, M_CHAIN = 5, M_AGPR_AB = 6, M_MIX_F32 = 7, M_MIX_F64 = 8,
       M_NOP_LONG = 9, M_MIX_F64x4 = 10, M_RANDOM = 11 };

template <int MODE>
__global__ __launch_bounds__(256) void k_probe(double* out, Stamp* st, int iters, double a0, double b0) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double sc[NACC];
  for (int i = 0; i < NACC; ++i) sc[i] = 0.0;
  double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
  if (MODE == M_ZERO) { a = 0.0; b = 0.0; }
  // M_RANDOM: a different uniformly random operand pair per accumulator and lane -- every mantissa bit of the multiplier inputs
  // toggles from one MFMA to the next (the clock the part holds depends on it: DVFS)
  double ra[NACC], rb[NACC];
  for (int i = 0; i < NACC; ++i) {
    unsigned long long z = ((unsigned long long)(blockIdx.x * blockDim.x + threadIdx.x) * NACC + i + 1) * 0x9E3779B97F4A7C15ull;
    z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
    ra[i] = (double)(long long)(z >> 11) * (1.0 / 4503599627370496.0) - 1.0;
    z *= 0x94D049BB133111EBull; z ^= z >> 31;
    rb[i] = (double)(long long)(z >> 11) * (1.0 / 4503599627370496.0) - 1.0;
  }
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = (float)threadIdx.x + i;
  double g[8];
  for (int i = 0; i < 8; ++i) g[i] = (double)threadIdx.x + i;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if constexpr (MODE == M_AGPR_ACC) {
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
      } else if constexpr (MODE == M_AGPR_AB) {
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "a"(a), "a"(b));
      } else if constexpr (MODE == M_RANDOM) {
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(ra[i]), "v"(rb[i]));
      } else if constexpr (MODE == M_4X4) {
        sc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, sc[i], 0, 0, 0);
      } else if constexpr (MODE == M_CHAIN) {
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[0], 0, 0, 0);
      } else {  // accumulators pinned to VGPRs (left alone, hipcc moves them to AGPRs in a 512-register kernel)
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
      }
      if constexpr (MODE == M_NOP) asm volatile("s_nop 15");
      if constexpr (MODE == M_NOP_LONG) asm volatile("s_nop 15\n s_nop 15\n s_nop 15");
      if constexpr (MODE == M_MIX_F32) {
        f[i] = __builtin_fmaf(f[i], 1.0000001f, 0.5f);
        f[(i + 4) & 7] = __builtin_fmaf(f[(i + 4) & 7], 1.0000001f, 0.25f);
      }
      if constexpr (MODE == M_MIX_F64) { g[i] = __builtin_fma(g[i], 1.0000001, a); }
      if constexpr (MODE == M_MIX_F64x4) {
        g[i] = __builtin_fma(g[i], 1.0000001, a);
        g[(i + 2) & 7] = __builtin_fma(g[(i + 2) & 7], 1.0000001, b);
        g[(i + 4) & 7] = __builtin_fma(g[(i + 4) & 7], 1.0000002, a);
        g[(i + 6) & 7] = __builtin_fma(g[(i + 6) & 7], 1.0000003, b);
      }
    }
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + sc[i];
  for (int i = 0; i < 8; ++i) s += (double)f[i] + g[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    st[w].cyc = c1 - c0;
    st[w].rt = r1 - r0;
  }
}

template <int MODE>
int run(const char* label, int grid, int block, int iters, double* buf, Stamp* st, double flops_per_mfma) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    k_probe<MODE><<<grid, block>>>(buf, st, iters, 1.0, 2.0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  const int nw = grid * (block / 64);
  std::vector<Stamp> h(nw);
  CK(hipMemcpy(h.data(), st, nw * sizeof(Stamp), hipMemcpyDeviceToHost));
  double c = 0, r = 0;
  for (auto& s : h) { c += (double)s.cyc; r += (double)s.rt; }
  const double cyc_per_mfma = c / nw / ((double)iters * NACC);
  const double ghz = c / r * 0.1;
  const double tf = (double)nw * iters * NACC * flops_per_mfma / ms / 1e9;
  printf("%-34s grid %5d x %3d : %8.3f ms  %7.1f cycles/MFMA/wave  clock %.2f GHz  %7.2f TFLOP/s (matrix)\n", label, grid, block, ms,
         cyc_per_mfma, ghz, tf);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return 0;
}

int main(int argc, char** argv) {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  printf("device %s, %d CUs\n", p.gcnArchName, cus);
  double* buf; Stamp* st;
  CK(hipMalloc((void**)&buf, (size_t)cus * 8 * 256 * 8));
  CK(hipMalloc((void**)&st, (size_t)cus * 8 * 4 * sizeof(Stamp)));
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  const double F16 = 2048.0, F4 = 512.0;
  struct G { int grid, block; const char* what; };
  const G geos[] = {{1, 64, "1 wave on the chip"}, {1, 256, "1 CU, 1 wave/SIMD"}, {1, 512, "1 CU, 2 waves/SIMD"}, {cus, 256, "all CUs, 1 wave/SIMD"},
                    {2 * cus, 256, "all CUs, 2 waves/SIMD"}, {4 * cus, 256, "all CUs, 4 waves/SIMD"}};
  for (const G& g : geos) {
    printf("---- %s\n", g.what);
    run<M_VGPR>("acc VGPR", g.grid, g.block, iters, buf, st, F16);
    run<M_AGPR_ACC>("acc AGPR", g.grid, g.block, iters, buf, st, F16);
    run<M_AGPR_AB>("A/B AGPR, acc VGPR", g.grid, g.block, iters, buf, st, F16);
    run<M_ZERO>("zero operands", g.grid, g.block, iters, buf, st, F16);
    run<M_RANDOM>("random operands", g.grid, g.block, iters, buf, st, F16);
    run<M_NOP>("+ s_nop 15 per MFMA", g.grid, g.block, iters, buf, st, F16);
    run<M_NOP_LONG>("+ 3 x s_nop 15 per MFMA", g.grid, g.block, iters, buf, st, F16);
    run<M_CHAIN>("dependent chain (1 acc)", g.grid, g.block, iters, buf, st, F16);
    run<M_4X4>("4x4x4 4-block shape", g.grid, g.block, iters, buf, st, F4);
    run<M_MIX_F32>("+ 2 v_fma_f32 per MFMA", g.grid, g.block, iters, buf, st, F16);
    run<M_MIX_F64>("+ 1 v_fma_f64 per MFMA", g.grid, g.block, iters, buf, st, F16);
    run<M_MIX_F64x4>("+ 4 v_fma_f64 per MFMA", g.grid, g.block, iters, buf, st, F16);
  }
  return 0;
}
