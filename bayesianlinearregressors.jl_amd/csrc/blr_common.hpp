// Shared device-side helpers for the gfx950 kernels (wave64, MFMA 16x16x4 in f64 / f32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace blr {

constexpr int kWave = 64;
constexpr int kThreads = 256;  // 4 waves: one per SIMD of a CU
constexpr int kWaves = kThreads / kWave;

enum : int { LAYOUT_COLVECS = 0, LAYOUT_ROWVECS = 1 };
enum : int { NOISE_ISOTROPIC = 0, NOISE_DIAGONAL = 1 };
enum : int { PRIOR_DENSE = 0, PRIOR_UPPER_FACTOR = 1, PRIOR_DIAGONAL = 2 };

// ---- MFMA traits ---------------------------------------------------------------------------
// v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32: lane l supplies A[i = l&15][k = l>>4] and
// B[k = l>>4][j = l&15].  For the Gram G = X S X' both operands are the SAME fragment shape
// frag(I)[l] = X[16I + (l&15), n0 + (l>>4)], so one LDS image feeds both sides.
// C/D maps differ: f64 row = (l>>4) + 4v, f32 row = 4(l>>4) + v; col = l&15 in both.
template <typename T>
struct Mfma;

template <>
struct Mfma<double> {
  typedef double acc4 __attribute__((ext_vector_type(4)));
  static constexpr int VEC = 2;  // elements per 16-byte vector
  static __device__ __forceinline__ acc4 mma(double a, double b, acc4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int crow(int lane, int v) { return (lane >> 4) + 4 * v; }
};

template <>
struct Mfma<float> {
  typedef float acc4 __attribute__((ext_vector_type(4)));
  static constexpr int VEC = 4;
  static __device__ __forceinline__ acc4 mma(float a, float b, acc4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int crow(int lane, int v) { return 4 * (lane >> 4) + v; }
};

// lower-triangular tile enumeration t = I(I+1)/2 + J, J <= I
__host__ __device__ constexpr int tile_I(int t) {
  int i = 0;
  while ((i + 1) * (i + 2) / 2 <= t) ++i;
  return i;
}
__host__ __device__ constexpr int tile_J(int t) { return t - tile_I(t) * (tile_I(t) + 1) / 2; }

// ---- global address space ------------------------------------------------------------------------
// A pointer that reaches a function through LDS or a noinline call has lost its address space: hipcc then emits
// flat_load, and while ONE flat access is pending every s_waitcnt it inserts is vmcnt(0) + lgkmcnt(0) (flat returns
// out of order) -- which serialises a software-pipelined LDS loop.  Cast such pointers back at the receiving end.
#define BLR_GLOBAL __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ const BLR_GLOBAL T* as_global(const T* p) { return (const BLR_GLOBAL T*)p; }
template <typename T>
__device__ __forceinline__ BLR_GLOBAL T* as_global(T* p) { return (BLR_GLOBAL T*)p; }

// ---- LDS-DMA piece: 64 lanes x 16 bytes, global -> LDS without passing through registers ----------------------------
// `dst` is the wave-uniform LDS address of the piece (lane l lands at dst + 16 l), `src` the per-lane global address.
// Issued through inline asm on purpose: for the BUILTIN (__builtin_amdgcn_global_load_lds) hipcc's wait-count insertion
// marks a "flat access pending" and from then on turns every LDS wait into lgkmcnt(0) -- in a software-pipelined loop
// that is a wait for the reads issued one instruction earlier (measured in the fused kernel: 59 % MFMA-busy).  Hidden
// from the compiler, its own LDS waits stay counted; the DMA itself is ordered by the explicit `s_waitcnt vmcnt(N)` +
// barrier the stage loops already carry.  M0 is saved and restored (the compiler treats it as reserved).
__device__ __forceinline__ void glds16(const BLR_GLOBAL void* src, const void* dst_wave_uniform) {
  const unsigned lds_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)dst_wave_uniform;
  unsigned keep, base;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "v_readfirstlane_b32 %1, %2\n\t"
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"  // SALU write of M0 -> LDS-DMA: one wait state
      "global_load_lds_dwordx4 %3, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep), "=&s"(base)
      : "v"(lds_addr), "v"(src)
      : "memory");
}

// The same piece with a SCALAR base: global address = saddr (SGPR pair) + voff (per-lane byte offset, 32 bits).  For a full
// tile the per-lane part of a stage piece's address never changes, so a whole stage costs scalar adds and ONE vector
// register instead of a 64-bit address computation per piece and lane.  SIZE = 16 (dwordx4) or 4 (dword) bytes per lane.
// NT: the piece is marked non-temporal -- for inputs that are read ONCE by one workgroup (the per-regressor kernels' X stream).  Measured on
// fused_i8_kernel: 4.19 against 4.23 ms per 4096 updates on one box and a 20 MHz higher sustained clock (less energy in the cache fills);
// never for tiles that other workgroups read again from the L2 (gram_tile_kernel).
template <int SIZE, int LANES = 64, bool NT = false>
__device__ __forceinline__ void glds_s(uint64_t saddr_uniform, unsigned voff, unsigned lds_addr_uniform) {
  static_assert((SIZE == 16 && LANES == 64) || SIZE == 4, "LDS-DMA piece width");
  static_assert(LANES == 64 || LANES == 32 || LANES == 16 || LANES == 8, "active lanes of the piece");
  // Call from wave-uniform control flow.  The LDS address goes through v_readfirstlane INSIDE the asm: hipcc hands an
  // "s" operand over in a VGPR when its own analysis calls the value divergent (the assembler then rejects the
  // s_mov -- a build failure, never a silent one; the 64-bit base must really be scalar for the same reason).
  // A 32-lane piece narrows EXEC inside the asm.
  unsigned keep, m0v;
  if constexpr (SIZE == 16 && NT) {
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "v_readfirstlane_b32 %1, %2\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %3, %4 nt\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep), "=&s"(m0v)
        : "v"(lds_addr_uniform), "v"(voff), "s"(saddr_uniform)
        : "memory");
  } else if constexpr (SIZE == 16) {
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "v_readfirstlane_b32 %1, %2\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %3, %4\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep), "=&s"(m0v)
        : "v"(lds_addr_uniform), "v"(voff), "s"(saddr_uniform)
        : "memory");
  } else if constexpr (LANES == 64) {
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "v_readfirstlane_b32 %1, %2\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dword %3, %4\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep), "=&s"(m0v)
        : "v"(lds_addr_uniform), "v"(voff), "s"(saddr_uniform)
        : "memory");
  } else {
    uint64_t keep_exec;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "v_readfirstlane_b32 %1, %3\n\t"
        "s_mov_b64 %2, exec\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_mov_b64 exec, %6\n\t"
        "global_load_lds_dword %4, %5\n\t"
        "s_mov_b64 exec, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep), "=&s"(m0v), "=&s"(keep_exec)
        : "v"(lds_addr_uniform), "v"(voff), "s"(saddr_uniform), "s"((uint64_t)(LANES == 32 ? 0xffffffffull : (LANES == 16 ? 0xffffull : 0xffull)))
        : "memory");
  }
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)p;
}

// ---- cross-lane helpers --------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  int v = __builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false);
  return __int_as_float(v);
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

// All-reduce (sum) over each 16-lane DPP row.  Butterfly with commutative adds: every lane of a row
// ends with the bitwise-identical value.  quad_perm[1,0,3,2], quad_perm[2,3,0,1], row_half_mirror,
// row_mirror.
template <typename T>
__device__ __forceinline__ T row16_allreduce(T x) {
  x += dpp_mov<0xB1>(x);
  x += dpp_mov<0x4E>(x);
  x += dpp_mov<0x141>(x);
  x += dpp_mov<0x140>(x);
  return x;
}

__device__ __forceinline__ double wave_allreduce(double x) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
  return x;
}

__device__ __forceinline__ double readlane_f64(double x, int srclane /*wave-uniform*/) {
  int lo = __builtin_amdgcn_readlane(__double2loint(x), srclane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(x), srclane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float readlane_f32(float x, int srclane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), srclane));
}
__device__ __forceinline__ double readlane(double x, int l) { return readlane_f64(x, l); }
__device__ __forceinline__ float readlane(float x, int l) { return readlane_f32(x, l); }

// Fixed-order block reduction: wave butterflies, then the 4 wave partials summed 0..3 by everyone.
// `scratch` needs kWaves doubles; contains two barriers.
__device__ __forceinline__ double block_allreduce(double v, double* scratch, int tid) {
  v = wave_allreduce(v);
  __syncthreads();
  if ((tid & 63) == 0) scratch[tid >> 6] = v;
  __syncthreads();
  return ((scratch[0] + scratch[1]) + scratch[2]) + scratch[3];
}
__device__ __forceinline__ int block_min_int(int v, int* scratch, int tid) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    int o = __shfl_xor(v, m, 64);
    v = o < v ? o : v;
  }
  __syncthreads();
  if ((tid & 63) == 0) scratch[tid >> 6] = v;
  __syncthreads();
  int a = scratch[0] < scratch[1] ? scratch[0] : scratch[1];
  int b = scratch[2] < scratch[3] ? scratch[2] : scratch[3];
  return a < b ? a : b;
}

__host__ __device__ __forceinline__ int pidx(int i, int k) { return i * (i + 1) / 2 + k; }  // packed lower, k <= i

// Reciprocal / reciprocal square root for the serial pivot chain of the blocked Cholesky: hardware seed plus
// Newton steps (f32: 1, f64: 2) instead of the ~15-20-instruction IEEE division / square-root expansions.
// Result within 1-2 ulp for normal positive inputs, which is all the pivot of an SPD matrix can be.
// a * b + c in ONE rounding, spelled out: where the bits matter (the elimination of phase_chol) the contraction is not left to the optimiser
// -- the SLP vectoriser otherwise turns some of the products into v_pk_mul_f32 + v_sub_f32
__device__ __forceinline__ float fused_madd(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fused_madd(double a, double b, double c) { return __builtin_fma(a, b, c); }

__device__ __forceinline__ float fast_rcp(float x) {
  float r = __builtin_amdgcn_rcpf(x);
  return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
}
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  return __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
}
__device__ __forceinline__ float fast_rsqrt(float x) {
  float y = __builtin_amdgcn_rsqf(x);
  float h = 0.5f * x * y;
  return __builtin_fmaf(__builtin_fmaf(-h, y, 0.5f), y, y);
}
__device__ __forceinline__ double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double h = 0.5 * x * y;
  y = __builtin_fma(__builtin_fma(-h, y, 0.5), y, y);
  h = 0.5 * x * y;
  return __builtin_fma(__builtin_fma(-h, y, 0.5), y, y);
}

}  // namespace blr
