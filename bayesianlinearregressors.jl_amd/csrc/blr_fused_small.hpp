// Fused per-regressor inference kernel for D <= 128 (one 256-thread workgroup per regressor).
//
// Replaces reference src/bayesian_linear_regression.jl:72-89 (__compute_inference_quantities),
// :55-58 (logpdf), :60-69 (posterior) with the direct Gram form of SURVEY.md 0.1:
//   phase 0  prior: SPD check + logdet Lw (dense prior: blocked Cholesky of Lw)              (:78)
//   phase 1  one streaming pass over X: MFMA SYRK  A = Lw + sum_n x_n w_n x_n'  (lower 16x16 tiles
//            only), per column mu_n = x_n'mw, delta_n, b += x_n delta_n w_n, q += delta_n^2 w_n,
//            l += log s_n                                                  (:79-84, :86, :57)
//   phase 2  blocked Cholesky A = L L' (T = L') with the trailing matrix in MFMA accumulators,
//            forward substitution u = L^-1 b fused into the panel step                   (:86, :67, :57)
//   phase 3  m = L^-T u, mw' = mw + m, evidence                                          (:64, :68)
// X is read from HBM exactly once; nothing intermediate touches HBM.
//
// LDS image of a stage (4*KS columns): [k-step j][row block I][lane l] holds
//   X[16I + (l&15), n0 + 4j + (l>>4)]  -- i.e. already in MFMA operand order, so every fragment
// read is one conflict-free ds_read of 64 consecutive elements.
//
// Code structure: the phases are separate NOINLINE device functions that talk through LDS.  Each gets
// its own register allocation; inlined into one body, hipcc's LICM hoists hundreds of loop-invariant
// address/mask/tile-coordinate values across the phases and the hot MFMA loop spills (measured: 1.5 KB
// of scratch per lane and 46 % MFMA utilisation).
#pragma once
#ifndef BLR_SMALL_NT
#define BLR_SMALL_NT true  /* the X stream of the per-regressor kernels is read once: non-temporal LDS-DMA pieces (blr_common.hpp, glds_s) */
#endif
#include <utility>

#include "blr_common.hpp"

// BLR_EXP: timing experiments only (never defined in the shipped build).
//   1 = stop after the Gram loop   2 = Gram loop without MFMA   3 = Gram loop without the column-vector work
//   4 = no global loads in the loop   5 = skip the back substitution
#ifndef BLR_EXP
#define BLR_EXP 0
#endif
// Phase functions: noinline (own register allocation, but the AMDGPU call ABI saves ~108 callee-saved
// VGPRs per call to scratch) or always-inline with an opaque thread id at phase entry.
// waves per SIMD requested for the fp32 D > 64 instantiations (fp64 needs the registers: 2).  Measured at D=128, N=4096:
// 2 -> 0.96 M, 3 -> 1.09 M, 4 -> 1.07 M updates/s (B = 4096); 4 slows the serial panel chain of the large-D path.
#ifndef BLR_F32_WAVES_PER_SIMD
#define BLR_F32_WAVES_PER_SIMD 3
#endif
#ifndef BLR_PHASE_INLINE
#define BLR_PHASE_INLINE 0
#endif
#if BLR_PHASE_INLINE
#define BLR_PHASE __device__ __forceinline__
#else
// `static` + `not_tail_called`: with internal linkage and no call site marked `tail`, LLVM's interprocedural register allocation
// (on by default for AMDGPU) applies its no-callee-saved-registers optimisation -- TargetFrameLowering::isSafeForNoCSROpt wants a local,
// non-recursive function none of whose calls is a tail call, and the optimiser marks every call of a function that takes no stack
// pointer `tail`.  Without the two attributes a phase that uses all 256 registers saves the ABI's ~125 callee-saved ones at its entry
// and restores them at its exit for a caller that keeps nothing in them: 612 B per lane of scratch in fused_small_kernel<double, 8, 4>
// (500 of them phase_gram's), ~0.9 MB of scratch traffic per regressor next to 4.2 MB of X.  With them: 252 B, of which the phases 60.
#define BLR_PHASE static __device__ __attribute__((noinline, not_tail_called))
#endif

namespace blr {

// BLR_STAMPS: diagnostic builds only (tools/chol_bench.hip) -- per-section cycle sums of phase_chol, wave 0
#ifdef BLR_STAMPS
__device__ unsigned long long g_stamps[8];
// section sums are kept in registers and flushed once (BLR_STAMP_FLUSH): a global read-modify-write per stamp would put a
// memory round trip in front of the next barrier and show up as "barrier time"
#define BLR_STAMP(slot)                                       \
  do {                                                        \
    unsigned long long t__ = __builtin_amdgcn_s_memtime();    \
    stamp_acc[slot] += t__ - stamp_prev;                      \
    stamp_prev = t__;                                         \
  } while (0)
#define BLR_STAMP_INIT                                                  \
  unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};           \
  unsigned long long stamp_prev = __builtin_amdgcn_s_memtime()
#define BLR_STAMP_FLUSH                                                                   \
  do {                                                                                    \
    if (threadIdx.x == 0 && blockIdx.x == 0)                                              \
      for (int q__ = 0; q__ < 8; ++q__) g_stamps[q__] += stamp_acc[q__];                  \
  } while (0)
#else
#define BLR_STAMP(slot) do {} while (0)
#define BLR_STAMP_INIT do {} while (0)
#define BLR_STAMP_FLUSH do {} while (0)
#endif

// BLR_GRAM_STAMPS: diagnostic builds only (tools/fused_bench.hip) -- per-section cycle sums of the stage loop of phase_gram,
// one row per wave of workgroup 0: [0] DMA wait, [1] barrier, [2] first fragments + DMA issue, [3] the eight k-steps
#ifdef BLR_GRAM_STAMPS
__device__ unsigned long long g_gstamps[4][8];
#define BLR_GSTAMP_INIT unsigned long long gstamp_prev = __builtin_amdgcn_s_memtime()
#define BLR_GSTAMP(slot)                                                                        \
  do {                                                                                          \
    unsigned long long t__ = __builtin_amdgcn_s_memtime();                                      \
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) g_gstamps[threadIdx.x >> 6][slot] += t__ - gstamp_prev; \
    gstamp_prev = t__;                                                                          \
  } while (0)
__device__ unsigned long long g_pstamps[16];  // per-phase cycle sums of workgroup 0 (fused_small_kernel), [15] = regressors
__device__ unsigned long long g_wgclk[8192][2];  // per workgroup (wave 0): shader cycles and 100 MHz ticks of its last ring loop
#define BLR_PSTAMP_INIT unsigned long long pstamp_prev = __builtin_amdgcn_s_memtime()
#ifdef BLR_PSTAMP_ALL  /* every workgroup, atomically */
#define BLR_PSTAMP(slot)                                                                   \
  do {                                                                                     \
    unsigned long long t__ = __builtin_amdgcn_s_memtime();                                 \
    if (threadIdx.x == 0) atomicAdd(&g_pstamps[slot], t__ - pstamp_prev);                  \
    pstamp_prev = t__;                                                                     \
  } while (0)
#else
#define BLR_PSTAMP(slot)                                                                   \
  do {                                                                                     \
    unsigned long long t__ = __builtin_amdgcn_s_memtime();                                 \
    if (blockIdx.x == 0 && threadIdx.x == 0) g_pstamps[slot] += t__ - pstamp_prev;         \
    pstamp_prev = t__;                                                                     \
  } while (0)
#endif
#else
#define BLR_GSTAMP_INIT do {} while (0)
#define BLR_GSTAMP(slot) do {} while (0)
#define BLR_PSTAMP_INIT do {} while (0)
#define BLR_PSTAMP(slot) do {} while (0)
#endif

template <typename T>
struct PosteriorArgs {
  const T* X; int64_t ldx, strideX;
  const T* y; int64_t stridey;
  const T* s; int64_t strides;
  const T* mw; int64_t stridemw;
  const T* Lw; int64_t ldl, strideLw;
  T* mw_post; int64_t stride_mwpost;
  T* T_post; int64_t ldt, strideT;
  T* Lw_post; int64_t ldlp, strideLp;
  double* logpdf;
  int32_t* info;
  int layout, noise_kind, prior_kind;
  int D, N, B;
  int vec_ok;  // ColVecs, 16-byte aligned columns: vector loads allowed
  int retry_only;  // fused_small_kernel: take only the regressors whose info word says kI8Retry (follow-up of fused_i8_kernel)
  // fused_i8_kernel<true> (diagonal noise), written by i8_noise_prep_kernel: y / sqrt(s) and 1 / sqrt(s), [B][N] at i8_stride each;
  // sum_n log s_n and a flag (some s_n not positive / not finite: the fp64 kernel reports it) per regressor
  const T* i8_yt; const T* i8_rw; int64_t i8_stride; const double* i8_logdet; const int32_t* i8_bad;
  const double* i8_rwmax;  // max_n 1 / sqrt(s_n): the row bounds of the sliced values x / sqrt(s_n) are bounds of x times this
  // hand-back accounting of the int8 route (launch_fused_i8): the retry launch counts the regressors it redoes into *i8_handed_slice
  // (zeroed by the slice's int8 launch) and *i8_handed_tot (blr_get_stat); the int8 launch of the NEXT slice reads the previous
  // slice's count -- *i8_prev_handed of i8_prev_n regressors -- and leaves its regressors to the fp64 kernel when that was > 1/4
  unsigned long long* i8_handed_tot; unsigned long long* i8_handed_slice; const unsigned long long* i8_prev_handed; int i8_prev_n;
  unsigned long long* i8_call_base;  // first slice of a call only: receives *i8_handed_tot as it stands when the call starts (blr_last_route: hand-backs of THIS call)
  // dense prior on the int8 route: logdet Lw and the status of its Cholesky per prior (i8_prior_logdet_kernel); stride 0 = one shared prior
  const double* i8_prior_logdet; const int32_t* i8_prior_info; int64_t i8_prior_stride;
  // D > 128, fp32: the design matrix is a random-Fourier basis phi_f(x_n) = rff_scale cos(Omega_f' x_n + phase_f) that is never
  // materialised (blr_posterior_rff_f32; rff_Omega != NULL: X is not read) -- planes_kernel evaluates it once per element
  const T* rff_Xin; int64_t rff_ldxin; const T* rff_Omega; int64_t rff_ldo; const T* rff_phase; T rff_scale; int rff_Din;
};
constexpr int kI8RetryCode = (int)0x80000007u;  // == kI8Retry (blr_fused_i8.hpp)

// per-regressor context handed to the phase functions through LDS (uniform values)
template <typename T>
struct RegCtx {
  const T* X; const T* y; const T* s; const T* mw; const T* Lw;
  int64_t ldx, ldl;
  int D, N, noise_kind, prior_kind;
  double logdet_Lw;  // (kept here between the glue phases of fused_small_kernel)
};

template <typename T, int NB>
struct SmallCfg {
  static constexpr int DP = 16 * NB;
  // k-steps (of 4 columns) per stage; chosen so PER is a multiple of the 16-byte vector width
  // (f32, NB <= 4: 16 k-steps = 64 columns = 16 KB per stage -- these shapes are bound by bytes in flight per CU, not LDS)
  static constexpr int KS = (NB < 3 || ((NB & 1) && sizeof(T) == 4) || (NB <= 4 && sizeof(T) == 4)) ? 16 : 8;
  static constexpr int NSC = 4 * KS;                    // columns per stage
  static constexpr int SLOT = KS * NB * 64;             // elements per LDS slot
  static constexpr int PER = SLOT / kThreads;           // elements per thread per stage
  static constexpr int NT = NB * (NB + 1) / 2;          // lower-triangular 16x16 tiles
  static constexpr int TPW = (NB == 8) ? 9 : (NT + kWaves - 1) / kWaves;  // accumulator tiles per wave
  static constexpr int PACKED = DP * (DP + 1) / 2;
  static constexpr int RED_BYTES = 16 * DP * 8;          // b partials: 4 waves x 4 lane-rows x DP doubles
  // staging area: two stage slots, or (NB == 8) the ring of gram_iso_ring: RING_NH halves of RING_HK k-steps.  f64: 4 x 4.
  // f32, measured back to back on one box (c2 shape, M updates/s): 4 halves of 4 k-steps 1.220, 3 halves of 8 (twice the MFMA
  // time between two barriers) 1.241, 4 halves of 8 (64 KB: two workgroups per CU instead of four) 0.981, **3 halves of 4: 1.354**
#ifndef BLR_RING_F32_HK
#define BLR_RING_F32_HK 4
#endif
#ifndef BLR_RING_F32_NH
#define BLR_RING_F32_NH 3
#endif
  static constexpr int RING_HK = sizeof(T) == 4 ? BLR_RING_F32_HK : 4;
#ifndef BLR_RING_F64_NH
#define BLR_RING_F64_NH 4
#endif
  static constexpr int RING_NH = sizeof(T) == 4 ? BLR_RING_F32_NH : BLR_RING_F64_NH;
  static constexpr int RING_BYTES = (NB == 8) ? RING_NH * RING_HK * NB * 64 * (int)sizeof(T) : 0;
  static constexpr int REGION0_A = (2 * SLOT * (int)sizeof(T) > RING_BYTES) ? 2 * SLOT * (int)sizeof(T) : RING_BYTES;
  static constexpr int REGION0_B = PACKED * (int)sizeof(T);
  static constexpr int REGION0_C = RED_BYTES;
  static constexpr int REGION0 =
      ((REGION0_A > REGION0_B ? (REGION0_A > REGION0_C ? REGION0_A : REGION0_C)
                              : (REGION0_B > REGION0_C ? REGION0_B : REGION0_C)) + 15) & ~15;
  // after region 0: ybuf[2][NSC], wbuf[2][NSC], bvec[DP], dinv[DP], mw[DP] (T); scratch; context
  static constexpr int OFF_Y = REGION0;
  static constexpr int OFF_W = OFF_Y + (2 * NSC > 128 ? 2 * NSC : 128) * (int)sizeof(T);  // (the ring keeps y for four halves of 32 columns)
  static constexpr int OFF_B = OFF_W + 2 * NSC * (int)sizeof(T);
  static constexpr int OFF_DINV = OFF_B + DP * (int)sizeof(T);
  static constexpr int OFF_MW = OFF_DINV + DP * (int)sizeof(T);
  static constexpr int OFF_SCR = (OFF_MW + DP * (int)sizeof(T) + 15) & ~15;  // 8 doubles + 8 ints
  static constexpr int OFF_CTX = OFF_SCR + 96;
  static constexpr int LDS_BYTES = (OFF_CTX + (int)sizeof(RegCtx<T>) + 15) & ~15;
};

// scalarise a value that is uniform by construction but lives in a VGPR (LDS broadcast loads, arguments
// of noinline functions): v_readfirstlane -> SGPR, so branches and address bases stay scalar.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int64_t uni(int64_t v) {
  int lo = __builtin_amdgcn_readfirstlane((int)(v & 0xffffffff));
  int hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
  return ((int64_t)hi << 32) | (uint32_t)lo;
}
template <typename P>
__device__ __forceinline__ P* uni(P* p) { return reinterpret_cast<P*>(uni((int64_t)(uintptr_t)p)); }

// ---- tile -> wave assignment -------------------------------------------------------------------------
// NB == 8 (D in 113..128, the headline shape): wave W owns block rows W and 7-W (9 tiles each), so only two
// A-side fragments per k-step need the Sigma^-1 scaling.  Otherwise round-robin over t = I(I+1)/2 + K.
// Tiles are selected by LDS ADDRESS (SGPR offsets), never by register index: all four waves run the same
// instruction stream.
__device__ __forceinline__ bool wave_tile(int NB, int wave, int i, int& I, int& K) {
  if (NB == 8) {
    I = (i <= wave) ? wave : 7 - wave;
    K = (i <= wave) ? i : i - (wave + 1);
    return true;
  }
  const int t = wave + kWaves * i;
  I = 0;
  while ((I + 1) * (I + 2) / 2 <= t) ++I;
  K = t - I * (I + 1) / 2;
  return t < NB * (NB + 1) / 2;
}

template <typename T, int NB>
using AccArr = typename Mfma<T>::acc4[SmallCfg<T, NB>::TPW];

// ---- stage loader ------------------------------------------------------------------------------
// MODE 0: ColVecs data, generic (any D, ldx, alignment)   MODE 1: RowVecs data
// MODE 2: prior factor U as pseudo-observations: element (d, j) = U[j + d*ld] for j <= d, else 0
// MODE 3: ColVecs data, 16-byte vectors (D % VEC == 0, 16-byte aligned columns)
template <typename T, int NB>
struct StageRegs {
  T x[SmallCfg<T, NB>::PER];
  T yv, wv;
};
template <typename T>
struct StageRegs<T, 0> {  // MODE 4: X goes straight to LDS, only the per-column scalars pass through registers
  T yv, wv;
};

__device__ __forceinline__ int frag_off(int NB, int d, int nl) {
  return ((((nl >> 2) * NB + (d >> 4)) * 4 + (nl & 3)) << 4) + (d & 15);
}

template <typename T, int NB, int MODE>
__device__ __forceinline__ void stage_load(StageRegs<T, NB>& r, const BLR_GLOBAL T* __restrict__ base, int64_t ld, int D, int ncols,
                                           int n0, int tid) {
  using C = SmallCfg<T, NB>;
  constexpr int VEC = Mfma<T>::VEC;
  // Opaque copy of tid: keeps the per-element index arithmetic inside the stage loop.  Hoisted out
  // (LICM) it costs > 100 VGPRs of loop-invariant addresses and masks and forces spills.
  asm volatile("" : "+v"(tid));
  if constexpr (MODE == 3) {
    typedef T vecT __attribute__((ext_vector_type(VEC)));
    constexpr int VPC = C::DP / VEC;  // vectors per (padded) column
    static_assert(C::PER % VEC == 0, "PER must be a multiple of the vector width");
#pragma unroll
    for (int e = 0; e < C::PER / VEC; ++e) {
      int vidx = tid + kThreads * e;
      int dv = vidx % VPC, nl = vidx / VPC;
      int n = n0 + nl;
      bool ok = dv * VEC < D && n < ncols;
      int64_t addr = ok ? (int64_t)n * ld + dv * VEC : 0;
      vecT v = *(const BLR_GLOBAL vecT*)(base + addr);
#pragma unroll
      for (int c = 0; c < VEC; ++c) r.x[e * VEC + c] = ok ? v[c] : T(0);
    }
  } else {
#pragma unroll
    for (int e = 0; e < C::PER; ++e) {
      int idx = tid + kThreads * e;
      int d, nl;
      if constexpr (MODE == 0) { d = idx % C::DP; nl = idx / C::DP; }
      else                     { nl = idx % C::NSC; d = idx / C::NSC; }
      int n = n0 + nl;
      bool ok = d < D && n < ncols && (MODE != 2 || n <= d);
      int64_t addr = (MODE == 0) ? (int64_t)n * ld + d : (int64_t)d * ld + n;
      T v = base[ok ? addr : 0];  // unconditional load of a valid address, then select: no branches
      r.x[e] = ok ? v : T(0);
    }
  }
}

template <typename T, int NB, int MODE>
__device__ __forceinline__ void stage_store(const StageRegs<T, NB>& r, T* __restrict__ slot, T* ybuf, T* wbuf, int tid) {
  using C = SmallCfg<T, NB>;
  constexpr int VEC = Mfma<T>::VEC;
  asm volatile("" : "+v"(tid));
  if constexpr (MODE == 3) {
    typedef T vecT __attribute__((ext_vector_type(VEC)));
    constexpr int VPC = C::DP / VEC;
#pragma unroll
    for (int e = 0; e < C::PER / VEC; ++e) {
      int vidx = tid + kThreads * e;
      int dv = vidx % VPC, nl = vidx / VPC;
      vecT v;
#pragma unroll
      for (int c = 0; c < VEC; ++c) v[c] = r.x[e * VEC + c];
      *reinterpret_cast<vecT*>(slot + frag_off(NB, dv * VEC, nl)) = v;
    }
  } else {
#pragma unroll
    for (int e = 0; e < C::PER; ++e) {
      int idx = tid + kThreads * e;
      int d, nl;
      if constexpr (MODE == 0) { d = idx % C::DP; nl = idx / C::DP; }
      else                     { nl = idx % C::NSC; d = idx / C::NSC; }
      slot[frag_off(NB, d, nl)] = r.x[e];
    }
  }
  if (tid < C::NSC) { ybuf[tid] = r.yv; wbuf[tid] = r.wv; }
}

// ---- MODE 4: ColVecs, 16-byte aligned columns, loaded straight into LDS (global_load_lds_dwordx4) ---------
// One wave-instruction moves 1 KiB = `FPG` consecutive fragments of the slot image (the image is lane-linear
// by construction, which is exactly what the LDS-DMA destination requires: M0 base + lane * 16 B); the
// per-lane GLOBAL address does the (column, row) -> fragment permutation.  No staging registers, no ds_write.
// Lanes whose element is outside the matrix (row >= D or column >= N) write zeros with a plain ds_write.
template <typename T, int NB, int KS = SmallCfg<T, NB>::KS>
__device__ __forceinline__ void stage_glds(T* __restrict__ slot, const BLR_GLOBAL T* __restrict__ base, int64_t ld, int D, int ncols,
                                           int n0, int wave, int lane) {
  constexpr int VEC = Mfma<T>::VEC;
  constexpr int FPG = (1024 / (int)sizeof(T)) / 64;  // fragments per wave-instruction: 2 (f64) / 4 (f32)
  constexpr int NG = KS * NB / FPG;                  // wave-instructions per stage
  static_assert((KS * NB) % FPG == 0, "slot must be a whole number of 1 KiB pieces");
  typedef T vecT __attribute__((ext_vector_type(VEC)));
  asm volatile("" : "+v"(lane));  // keep the index arithmetic inside the stage loop (see stage_load)
  const int e0 = lane * VEC;      // element offset inside the 1 KiB piece
  const int fl = e0 >> 6, ls = e0 & 63, q = ls >> 4, r = ls & 15;
#pragma unroll
  for (int g0 = 0; g0 < NG; g0 += kWaves) {
    const int g = g0 + wave;
    if (NG % kWaves == 0 || g < NG) {
      const int F = g * FPG + fl;  // flat fragment index j * NB + I
      const int j = F / NB, I = F - j * NB;
      const int n = n0 + 4 * j + q, d = 16 * I + r;
      T* dst = slot + g * (FPG * 64);  // wave-uniform
      if (d < D && n < ncols) {
        glds16((const BLR_GLOBAL void*)(base + (int64_t)n * ld + d), dst);
      } else {
        vecT z;
#pragma unroll
        for (int c = 0; c < VEC; ++c) z[c] = T(0);
        *reinterpret_cast<vecT*>(dst + e0) = z;
      }
    }
  }
}

// Full tiles (D == 16 NB, the stage entirely inside the matrix, NB a multiple of the fragments per piece): piece g of a
// stage covers fragments F = g FPG + fl with I = (g FPG) % NB + fl and j = (g FPG) / NB, so the address splits into a
// wave-uniform part  base + ((n0 + 4 j) ld + 16 ((g FPG) % NB)) elements  and a per-lane part  q ld + 16 fl + r  that is
// the same for every piece of every stage (`voff`, from glds_lane_offset).  No bounds checks, no per-lane 64-bit math.
template <typename T>
__device__ __forceinline__ unsigned glds_lane_offset(int64_t ld, int lane) {
  constexpr int VEC = Mfma<T>::VEC;
  const int e0 = lane * VEC;
  const int fl = e0 >> 6, ls = e0 & 63, q = ls >> 4, r = ls & 15;
  return (unsigned)(((int64_t)q * ld + 16 * fl + r) * (int64_t)sizeof(T));
}
template <typename T, int NB, int KS = SmallCfg<T, NB>::KS>
__device__ __forceinline__ void stage_glds_full(const T* __restrict__ slot, const BLR_GLOBAL T* __restrict__ base /*uniform*/,
                                                int64_t ld, int n0, int wave /*uniform*/, unsigned voff) {
  constexpr int FPG = (1024 / (int)sizeof(T)) / 64;
  constexpr int NG = KS * NB / FPG;
  static_assert(NB % FPG == 0 && (KS * NB) % FPG == 0, "pieces must not straddle k-steps");
  const unsigned slot_addr = uni((int)lds_addr_of(slot));
#pragma unroll
  for (int g0 = 0; g0 < NG; g0 += kWaves) {
    const int g = g0 + wave;
    if (NG % kWaves == 0 || g < NG) {
      const int j = (g * FPG) / NB, I0 = (g * FPG) % NB;
      const uint64_t saddr = (uint64_t)(uintptr_t)base + (uint64_t)(((int64_t)(n0 + 4 * j) * ld + 16 * I0) * (int64_t)sizeof(T));
      glds_s<16>(uni((int64_t)saddr), voff, slot_addr + (unsigned)(g * 1024));
    }
  }
}

// ---- k-step operands -----------------------------------------------------------------------------------
template <int NB>
struct WaveOps {  // SGPRs: LDS element offsets of the A-/B-side fragments of accumulator slot i (-1: unused)
  int offA[SmallCfg<double, NB>::TPW];
  int offB[SmallCfg<double, NB>::TPW];
};

template <typename T, int NB>
__device__ __forceinline__ WaveOps<NB> make_wave_ops(int wave) {
  WaveOps<NB> o;
#pragma unroll
  for (int i = 0; i < SmallCfg<T, NB>::TPW; ++i) {
    int I, K;
    const bool ok = wave_tile(NB, wave, i, I, K);
    o.offA[i] = ok ? I * 64 : -1;
    o.offB[i] = ok ? K * 64 : -1;
  }
  return o;
}

template <typename T, int NB>
struct KFrags {
  T a[(NB == 8) ? 2 : SmallCfg<T, NB>::TPW];  // A-side fragments (NB == 8: block rows W and 7-W)
  T b[SmallCfg<T, NB>::TPW];
  T w;
};

// WS >= 0: the wave index is a compile-time constant (NB == 8 fast path: phase_gram runs one copy of the stage
// loop per wave), so every fragment offset is an instruction immediate and the A-side choice needs no select.
template <typename T, int NB, int WS>
__device__ __forceinline__ void load_kfrags(KFrags<T, NB>& fr, const WaveOps<NB>& ops, const T* __restrict__ kimg,
                                            const T* __restrict__ wbuf, int j, int wave, int lane) {
  using C = SmallCfg<T, NB>;
  fr.w = wbuf[4 * j + (lane >> 4)];
  if constexpr (NB == 8 && WS >= 0) {
    fr.a[0] = kimg[WS * 64 + lane];
    fr.a[1] = kimg[(7 - WS) * 64 + lane];
#pragma unroll
    for (int i = 0; i < C::TPW; ++i) fr.b[i] = kimg[((i <= WS) ? i : i - (WS + 1)) * 64 + lane];
  } else if constexpr (WS >= 0) {  // round-robin map, static wave: tile t = WS + 4 i
#pragma unroll
    for (int i = 0; i < C::TPW; ++i) {
      const int t = WS + kWaves * i;
      if (t < C::NT) {
        fr.a[i] = kimg[tile_I(t) * 64 + lane];
        fr.b[i] = kimg[tile_J(t) * 64 + lane];
      }
    }
  } else {
    if constexpr (NB == 8) {
      fr.a[0] = kimg[wave * 64 + lane];
      fr.a[1] = kimg[(7 - wave) * 64 + lane];
    } else {
#pragma unroll
      for (int i = 0; i < C::TPW; ++i) fr.a[i] = kimg[(ops.offA[i] < 0 ? 0 : ops.offA[i]) + lane];
    }
#pragma unroll
    for (int i = 0; i < C::TPW; ++i) fr.b[i] = kimg[(ops.offB[i] < 0 ? 0 : ops.offB[i]) + lane];
  }
}

template <typename T, int NB, int WS>
__device__ __forceinline__ void mma_kstep(AccArr<T, NB>& acc, const KFrags<T, NB>& fr, const WaveOps<NB>& ops, int wave) {
  using C = SmallCfg<T, NB>;
#if BLR_EXP == 2
  acc[0][0] += fr.a[0] * fr.b[0] * fr.w;
  return;
#endif
  if constexpr (NB == 8) {
    const T alo = fr.a[0] * fr.w, ahi = fr.a[1] * fr.w;  // Sigma_y^-1 applied on the A side only
#pragma unroll
    for (int i = 0; i < C::TPW; ++i) {
      if constexpr (WS >= 0) {
        acc[i] = Mfma<T>::mma((i <= WS) ? alo : ahi, fr.b[i], acc[i]);
      } else {
        const T asel = (i <= wave) ? alo : ahi;  // wave-uniform select
        acc[i] = Mfma<T>::mma(asel, fr.b[i], acc[i]);
      }
    }
  } else if constexpr (WS >= 0) {
#pragma unroll
    for (int i = 0; i < C::TPW; ++i) {
      if (WS + kWaves * i < C::NT) acc[i] = Mfma<T>::mma(fr.a[i] * fr.w, fr.b[i], acc[i]);  // resolved at compile time
    }
  } else {
#pragma unroll
    for (int i = 0; i < C::TPW; ++i) {
      if (ops.offA[i] >= 0) acc[i] = Mfma<T>::mma(fr.a[i] * fr.w, fr.b[i], acc[i]);  // scalar branch
    }
  }
}

// per-column vector work for the 4 columns of one k-step: lane (r, q) holds rows 16I + r of column q.
// `gate` (0 or 1, wave-uniform) switches the accumulation off without a branch (prior pseudo-columns).
template <typename T, int NB>
__device__ __forceinline__ void vector_kstep(const T* __restrict__ kimg, T w, T yv, const T* __restrict__ mwl,
                                             double (&bacc)[NB], double& qacc, int lane, T gate) {
  T f[NB];
#pragma unroll
  for (int I = 0; I < NB; ++I) f[I] = kimg[I * 64 + lane];
  T mu = T(0);
#pragma unroll
  for (int I = 0; I < NB; ++I) mu += f[I] * mwl[16 * I + (lane & 15)];
  mu = row16_allreduce(mu);
  const T delta = (yv - mu) * gate;  // :82  y - mean(fx)
  const T rn = delta * w;
  if ((lane & 15) == 0) qacc += (double)delta * (double)rn;
#pragma unroll
  for (int I = 0; I < NB; ++I) bacc[I] += (double)f[I] * (double)rn;
}

// One stage (KS k-steps): MFMAs on the wave's tiles, software-pipelined one k-step deep -- the fragments of
// k-step j+1 are requested before the MFMAs of k-step j issue, so LDS latency hides under the MFMA pipe.
template <typename T, int NB, int WS>
__device__ __forceinline__ void compute_stage(const T* __restrict__ slot, const T* __restrict__ ybuf,
                                              const T* __restrict__ wbuf, const WaveOps<NB>& ops, AccArr<T, NB>& acc,
                                              double (&bacc)[NB], double& qacc, const T* __restrict__ mwl, int wave,
                                              int lane, bool is_data) {
  using C = SmallCfg<T, NB>;
  KFrags<T, NB> f0, f1;
  load_kfrags<T, NB, WS>(f0, ops, slot, wbuf, 0, wave, lane);
  if constexpr (WS >= 0) {
    // static wave: k-steps in groups of four, the owner k-step (jj == WS) carries the column-vector work in
    // the SAME basic block as its MFMAs so the scheduler can sink the VALU work into the MFMA shadows
    const T gate = is_data ? T(1) : T(0);
#pragma unroll 1
    for (int j = 0; j < C::KS; j += 4) {
#pragma unroll
      for (int jj = 0; jj < 4; jj += 2) {
        load_kfrags<T, NB, WS>(f1, ops, slot + (j + jj + 1) * NB * 64, wbuf, j + jj + 1, wave, lane);
        mma_kstep<T, NB, WS>(acc, f0, ops, wave);
        if (jj == WS && BLR_EXP != 3)
          vector_kstep<T, NB>(slot + (j + jj) * NB * 64, f0.w, ybuf[4 * (j + jj) + (lane >> 4)], mwl, bacc, qacc, lane, gate);
        if (j + jj + 2 < C::KS) load_kfrags<T, NB, WS>(f0, ops, slot + (j + jj + 2) * NB * 64, wbuf, j + jj + 2, wave, lane);
        mma_kstep<T, NB, WS>(acc, f1, ops, wave);
        if (jj + 1 == WS && BLR_EXP != 3)
          vector_kstep<T, NB>(slot + (j + jj + 1) * NB * 64, f1.w, ybuf[4 * (j + jj + 1) + (lane >> 4)], mwl, bacc, qacc,
                              lane, gate);
      }
    }
  } else {
#pragma unroll 1
    for (int j = 0; j < C::KS; j += 2) {
      load_kfrags<T, NB, WS>(f1, ops, slot + (j + 1) * NB * 64, wbuf, j + 1, wave, lane);
      mma_kstep<T, NB, WS>(acc, f0, ops, wave);
      if ((j & 3) == wave && is_data && BLR_EXP != 3)
        vector_kstep<T, NB>(slot + j * NB * 64, f0.w, ybuf[4 * j + (lane >> 4)], mwl, bacc, qacc, lane, T(1));
      if (j + 2 < C::KS) load_kfrags<T, NB, WS>(f0, ops, slot + (j + 2) * NB * 64, wbuf, j + 2, wave, lane);
      mma_kstep<T, NB, WS>(acc, f1, ops, wave);
      if (((j + 1) & 3) == wave && is_data && BLR_EXP != 3)
        vector_kstep<T, NB>(slot + (j + 1) * NB * 64, f1.w, ybuf[4 * (j + 1) + (lane >> 4)], mwl, bacc, qacc, lane, T(1));
    }
  }
}

// ---- NB == 8 (D in 113..128), static wave: the pinned-schedule stage ---------------------------------------
// Left to itself hipcc sinks the fragment reads of k-step j+1 to just in front of their first use and waits
// `lgkmcnt(0)` there: every k-step exposes one LDS round trip with the matrix pipe idle (measured: 59 % MFMA-busy,
// 0.55 of the f64 matrix peak -- which IS 64 cycles per v_mfma_f64_16x16x4 with the accumulators in VGPRs,
// profiles/r02_microbench_mfma_f64_probe.txt).  Here the order is pinned with sched_barrier: the reads of k-step j+1
// are ISSUED before the nine MFMAs of k-step j, so their latency hides under >= 576 cycles of matrix work and the
// wait the compiler places in front of k-step j+1's first use is a counted one that has already expired.
// Wave W needs block rows 0 .. 7-W only (its tiles are (W, 0..W) and (7-W, 0..7-W)); the k-step it owns for the
// column-vector work reads all eight.
template <typename T, int WS>
struct KF8 {
  T fb[8];  // fragment of block row I: X[16I + (l & 15), n0 + 4j + (l >> 4)]
  T w;      // Sigma_y^-1 of column 4j + (l >> 4)
  T yv;     // y of that column (owner k-steps only)
};

template <typename T, int WS, bool ISO>
__device__ __forceinline__ void load_kf8(KF8<T, WS>& f, const T* __restrict__ slot, const T* __restrict__ ybuf,
                                         const T* __restrict__ wbuf, int j /*compile-time after unrolling*/, int lane) {
  const bool owner = (j & 3) == WS;
  const T* kimg = slot + j * 8 * 64;
  if (!ISO) f.w = wbuf[4 * j + (lane >> 4)];
  if (owner) f.yv = ybuf[4 * j + (lane >> 4)];
#pragma unroll
  for (int I = 0; I < 8; ++I)
    if (owner || I <= 7 - WS) f.fb[I] = kimg[I * 64 + lane];
}

template <typename T, int WS, bool ISO>
__device__ __forceinline__ void mma_kf8(AccArr<T, 8>& acc, const KF8<T, WS>& f) {
  // Sigma_y^-1 on the A side only; isotropic noise: applied once to the finished accumulators instead
  const T alo = ISO ? f.fb[WS] : f.fb[WS] * f.w;
  const T ahi = ISO ? f.fb[7 - WS] : f.fb[7 - WS] * f.w;
#pragma unroll
  for (int i = 0; i < 9; ++i) acc[i] = Mfma<T>::mma((i <= WS) ? alo : ahi, f.fb[(i <= WS) ? i : i - (WS + 1)], acc[i]);
}

template <typename T, int WS, bool ISO>
__device__ __forceinline__ void vector_kf8(const KF8<T, WS>& f, T wiso, const T (&mwf)[8], double (&bacc)[8], double& qacc,
                                           int lane, T gate) {
  T mu = T(0);
#pragma unroll
  for (int I = 0; I < 8; ++I) mu += f.fb[I] * mwf[I];
  mu = row16_allreduce(mu);
  const T delta = (f.yv - mu) * gate;  // :82  y - mean(fx)
  const T rn = delta * (ISO ? wiso : f.w);
  if ((lane & 15) == 0) qacc += (double)delta * (double)rn;
#pragma unroll
  for (int I = 0; I < 8; ++I) bacc[I] += (double)f.fb[I] * (double)rn;
}

// the first k-step's fragments of a stage (issued right after the stage barrier, in front of the next stage's DMA issue)
template <typename T, int WS, bool ISO>
__device__ __forceinline__ void stage8_prologue(KF8<T, WS>& f0, const T* __restrict__ slot, const T* __restrict__ ybuf,
                                                const T* __restrict__ wbuf, int lane) {
  load_kf8<T, WS, ISO>(f0, slot, ybuf, wbuf, 0, lane);
  __builtin_amdgcn_sched_barrier(0);
}

template <typename T, int WS, bool ISO>
__device__ __forceinline__ void stage8_body(KF8<T, WS>& f0, const T* __restrict__ slot, const T* __restrict__ ybuf,
                                            const T* __restrict__ wbuf, AccArr<T, 8>& acc, double (&bacc)[8], double& qacc,
                                            const T (&mwf)[8], T wiso, int lane, T gate) {
  constexpr int KS = SmallCfg<T, 8>::KS;
  KF8<T, WS> f1;
#pragma unroll
  for (int j = 0; j < KS; j += 2) {
    load_kf8<T, WS, ISO>(f1, slot, ybuf, wbuf, j + 1, lane);
    __builtin_amdgcn_sched_barrier(0);
    mma_kf8<T, WS, ISO>(acc, f0);
    if ((j & 3) == WS && BLR_EXP != 3) vector_kf8<T, WS, ISO>(f0, wiso, mwf, bacc, qacc, lane, gate);
    __builtin_amdgcn_sched_barrier(0);
    if (j + 2 < KS) load_kf8<T, WS, ISO>(f0, slot, ybuf, wbuf, j + 2, lane);
    __builtin_amdgcn_sched_barrier(0);
    mma_kf8<T, WS, ISO>(acc, f1);
    if (((j + 1) & 3) == WS && BLR_EXP != 3) vector_kf8<T, WS, ISO>(f1, wiso, mwf, bacc, qacc, lane, gate);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---- NB == 8, isotropic noise, full tiles: the ring loop ------------------------------------------------------------
// The 64 KB staging area is a ring of FOUR half-stages (4 k-steps = 16 columns each).  While half h computes, half h+1 is
// already visible to every wave, half h+2 is landing and the pieces of half h+3 are issued one per k-step.  A piece has >= 5
// k-steps to land before the counted `s_waitcnt vmcnt(N)` that retires it.  ONE barrier per half, at its END: it publishes
// half h+2 and frees the slot of half h; because half h+1 was published a barrier earlier, the first fragments of half h+1
// are requested BEFORE the barrier and the first MFMA behind it issues at once.  Within a half every wave owns exactly one
// k-step of column-vector work, so all waves reach the barrier together.  y travels by LDS-DMA too (wave 0, one dword piece
// per half): the loop has no compiler-visible memory operation.
//
// Inside a k-step EVERYTHING that is not an MFMA is slotted, by hand, into the gaps between the nine MFMAs (sched_barrier
// pins the order): the fragment reads of the next k-step two blocks per gap, the DMA piece in a gap of its own, the
// column-vector work in eight short chunks.  A v_mfma_f64_16x16x4 occupies the matrix pipe for 64 cycles; whatever the
// wave issues behind it within those 64 cycles is free, anything longer leaves the pipe idle -- measured on this loop
// (tools/ring_probe.hip): one DMA piece issued BETWEEN two MFMA groups cost 117 cycles of a 576-cycle k-step, the reads
// and the vector work another ~100.
// MWZ: the prior mean is identically zero (checked once per regressor): mean(fx) = X'mw = 0 exactly, the dot products and
// the butterfly are skipped -- bit-identical to the general path, 17 of the 27 f64 vector operations of an owned k-step gone.
template <typename T, int WS, bool MWZ>
struct VecWork8 {
  T a, b, mu, rn;
  template <int C>
  __device__ __forceinline__ void step(const KF8<T, WS>& f, T wiso, const T (&mwf)[8], double (&bacc)[8], double& qacc, int lane) {
    if constexpr (C == 0 && !MWZ) { a = f.fb[0] * mwf[0]; a += f.fb[1] * mwf[1]; a += f.fb[2] * mwf[2]; a += f.fb[3] * mwf[3]; }
    if constexpr (C == 1 && !MWZ) { b = f.fb[4] * mwf[4]; b += f.fb[5] * mwf[5]; b += f.fb[6] * mwf[6]; b += f.fb[7] * mwf[7]; }
    if constexpr (C == 2 && !MWZ) { mu = a + b; mu += dpp_mov<0xB1>(mu); }   // 16-lane butterfly (row16_allreduce), one step per chunk
    if constexpr (C == 3 && !MWZ) { mu += dpp_mov<0x4E>(mu); }
    if constexpr (C == 4 && !MWZ) { mu += dpp_mov<0x141>(mu); }
    if constexpr (C == 5) {
      if constexpr (MWZ) mu = T(0);
      else mu += dpp_mov<0x140>(mu);
      const T delta = f.yv - mu;  // :82  y - mean(fx)
      rn = delta * wiso;
      if ((lane & 15) == 0) qacc += (double)delta * (double)rn;
    }
    if constexpr (C == 6) {
#pragma unroll
      for (int I = 0; I < 4; ++I) bacc[I] += (double)f.fb[I] * (double)rn;
    }
    if constexpr (C == 7) {
#pragma unroll
      for (int I = 4; I < 8; ++I) bacc[I] += (double)f.fb[I] * (double)rn;
    }
  }
};

// One k-step of the ring loop.  fc: fragments of this k-step (J = its index in the half; the wave owns the column-vector
// work iff J == WS).  fn <- fragments of the NEXT k-step, read from kimg_n / yb_n (JN = its index in its half).
// `piece` / `piece2`: callables issuing this k-step's LDS-DMA pieces (may do nothing).
template <typename T, int WS, bool MWZ, int J, int JN, typename P1, typename P2>
__device__ __forceinline__ void kstep_ring(AccArr<T, 8>& acc, const KF8<T, WS>& fc, KF8<T, WS>& fn, const T* __restrict__ kimg_n,
                                           const T* __restrict__ yb_n, T wiso, const T (&mwf)[8], double (&bacc)[8], double& qacc,
                                           int lane, P1 piece, P2 piece2) {
  constexpr bool OWN = ((J & 3) == WS) && (BLR_EXP != 3);   // (f32: halves of 8 k-steps, two of them a wave's own)
  constexpr bool OWN_N = ((JN & 3) == WS);
  constexpr int NBLK_N = OWN_N ? 8 : 8 - WS;  // blocks of the next k-step this wave needs
  VecWork8<T, WS, MWZ> vw;
  const T alo = fc.fb[WS], ahi = fc.fb[7 - WS];
  auto mma = [&](auto itag) {
    constexpr int i = decltype(itag)::value;
    acc[i] = Mfma<T>::mma((i <= WS) ? alo : ahi, fc.fb[(i <= WS) ? i : i - (WS + 1)], acc[i]);
  };
  auto rd = [&](auto itag) {
    constexpr int I = decltype(itag)::value;
    if constexpr (I < NBLK_N) fn.fb[I] = kimg_n[I * 64 + lane];
  };
#define BLR_SB __builtin_amdgcn_sched_barrier(0)
#define BLR_IC(k) std::integral_constant<int, k>{}
  mma(BLR_IC(0)); BLR_SB;
  rd(BLR_IC(0)); rd(BLR_IC(1));
  if constexpr (OWN) vw.template step<0>(fc, wiso, mwf, bacc, qacc, lane);
  BLR_SB; mma(BLR_IC(1)); BLR_SB;
  rd(BLR_IC(2)); rd(BLR_IC(3));
  if constexpr (OWN) vw.template step<1>(fc, wiso, mwf, bacc, qacc, lane);
  BLR_SB; mma(BLR_IC(2)); BLR_SB;
  rd(BLR_IC(4)); rd(BLR_IC(5));
  if constexpr (OWN) vw.template step<2>(fc, wiso, mwf, bacc, qacc, lane);
  BLR_SB; mma(BLR_IC(3)); BLR_SB;
  rd(BLR_IC(6)); rd(BLR_IC(7));
  if constexpr (OWN_N) fn.yv = yb_n[4 * JN + (lane >> 4)];
  if constexpr (OWN) vw.template step<3>(fc, wiso, mwf, bacc, qacc, lane);
  BLR_SB; mma(BLR_IC(4)); BLR_SB;
  piece();
  BLR_SB; mma(BLR_IC(5)); BLR_SB;
  if constexpr (OWN) vw.template step<4>(fc, wiso, mwf, bacc, qacc, lane);
  BLR_SB; mma(BLR_IC(6)); BLR_SB;
  piece2();
  if constexpr (OWN) vw.template step<5>(fc, wiso, mwf, bacc, qacc, lane);
  BLR_SB; mma(BLR_IC(7)); BLR_SB;
  if constexpr (OWN) vw.template step<6>(fc, wiso, mwf, bacc, qacc, lane);
  BLR_SB; mma(BLR_IC(8)); BLR_SB;
  if constexpr (OWN) vw.template step<7>(fc, wiso, mwf, bacc, qacc, lane);
  BLR_SB;
#undef BLR_SB
#undef BLR_IC
}

// NH = ring slots: 4 (the pipeline described above) or 3 (48 KB: half h+2 is issued during half h into the slot the barrier at
// the end of half h-1 freed, and must have landed by the end of half h -- enough when three workgroups share the CU and a
// half lasts three times as long)
template <typename T, int WS, bool MWZ, int NH = SmallCfg<T, 8>::RING_NH>
__device__ __forceinline__ void gram_iso_ring(T* __restrict__ ring, T* __restrict__ ybuf, const BLR_GLOBAL T* X /*uniform*/,
                                              const BLR_GLOBAL T* y /*uniform*/, int64_t ldx, int N, unsigned voff, int lane,
                                              AccArr<T, 8>& acc, double (&bacc)[8], double& qacc, const T (&mwf)[8], T wiso) {
  static_assert(NH == 3 || NH == 4, "ring depth");
  constexpr int NB = 8, HK = SmallCfg<T, 8>::RING_HK;
  constexpr int LA = NH - 1;                           // issue distance in halves
  auto slot_of = [](int h) { return NH == 4 ? (h & 3) : (h % 3); };
  constexpr int HALF = HK * NB * 64;                   // elements per half
  constexpr int HC = 4 * HK;                           // columns per half
  constexpr int FPG = (1024 / (int)sizeof(T)) / 64;    // fragments per 1 KiB piece
  constexpr int PW = HK * NB / FPG / kWaves;           // pieces per wave per half: 4 (f64) / 2 (f32)
  constexpr int PWT = PW + (WS == 0 ? 1 : 0);          // + the y piece of wave 0
  constexpr int YL = HC * (int)sizeof(T) / 4;          // lanes of the y piece: 32 (f64) / 16 (f32)
  unsigned ring_addr = lds_addr_of(ring), ybuf_addr = lds_addr_of(ybuf);
  // pinned in vector registers: rematerialised in the loop they cost a scalar-memory load plus an lgkmcnt(0) wait per use
  asm volatile("" : "+v"(ring_addr), "+v"(ybuf_addr));
  const int nh = N / HC;
  // Halves are issued strictly in order, so the global address of a piece is a running 64-bit scalar plus a per-piece constant:
  // two scalar adds per piece (the 64-bit multiplications of (16 h + 4 j) ldx per piece were a dozen scalar instructions each,
  // in the gap between two MFMAs)
  uint64_t nextX = (uint64_t)(uintptr_t)X, nextY = (uint64_t)(uintptr_t)y;
  const uint64_t stepX = (uint64_t)((int64_t)HC * ldx * (int64_t)sizeof(T));
  uint64_t offp[PW];
#pragma unroll
  for (int p = 0; p < PW; ++p) {
    const int g = p * kWaves + WS, j = (g * FPG) / NB, I0 = (g * FPG) % NB;
    offp[p] = (uint64_t)(((int64_t)(4 * j) * ldx + 16 * I0) * (int64_t)sizeof(T));
  }
  auto issue_half = [&](int h) {  // h only selects the ring slot; the addresses follow the call order
#ifndef RING_NODMA
    const unsigned slot = ring_addr + (unsigned)(slot_of(h) * HALF * (int)sizeof(T));
#pragma unroll
    for (int p = 0; p < PW; ++p)
      glds_s<16, 64, BLR_SMALL_NT>(uni((int64_t)(nextX + offp[p])), voff, slot + (unsigned)((p * kWaves + WS) * 1024));
    if constexpr (WS == 0)
      glds_s<4, YL>(uni((int64_t)nextY), (unsigned)(lane * 4), ybuf_addr + (unsigned)(slot_of(h) * HC * (int)sizeof(T)));
#endif
    nextX += stepX;
    nextY += (uint64_t)(HC * sizeof(T));
  };
  // retire everything but the youngest `keep_halves` halves of this wave's pieces
  auto retire = [&](int keep_halves) {
#ifndef RING_NORETIRE
    if (keep_halves >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PWT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  };
  for (int h = 0; h < LA && h < nh; ++h) issue_half(h);
  retire((NH == 4 && nh > 2) ? 1 : 0);  // halves 0 and 1 landed (this wave's pieces) ...
  __syncthreads();                      // ... and everybody's: visible
  KF8<T, WS> f0, f1;
  load_kf8<T, WS, true>(f0, ring, ybuf, ybuf, 0, lane);
  __builtin_amdgcn_sched_barrier(0);
  auto none = [] {};
#pragma unroll 1
  for (int h = 0; h < nh; ++h) {
    // here: halves h and h+1 are visible, f0 holds the fragments of (h, k-step 0), half h+2 is landing
    const T* slot = ring + slot_of(h) * HALF;
    const T* yb = ybuf + slot_of(h) * HC;
    const T* slot_n = ring + slot_of(h + 1) * HALF;
    const T* yb_n = ybuf + slot_of(h + 1) * HC;
    const bool more = h + LA < nh;  // slot (h + LA) % NH was freed by the barrier that ended half h-1
    // all pieces of half h+LA in ONE burst inside k-step 0 (back to back they cost ~46 cycles each, one at a time ~100)
    auto pall = [&] { if (more) issue_half(h + LA); };
    kstep_ring<T, WS, MWZ, 0, 1>(acc, f0, f1, slot + 1 * NB * 64, yb, wiso, mwf, bacc, qacc, lane, pall, none);
    kstep_ring<T, WS, MWZ, 1, 2>(acc, f1, f0, slot + 2 * NB * 64, yb, wiso, mwf, bacc, qacc, lane, none, none);
    kstep_ring<T, WS, MWZ, 2, 3>(acc, f0, f1, slot + 3 * NB * 64, yb, wiso, mwf, bacc, qacc, lane, none, none);
    if constexpr (HK == 8) {
      kstep_ring<T, WS, MWZ, 3, 4>(acc, f1, f0, slot + 4 * NB * 64, yb, wiso, mwf, bacc, qacc, lane, none, none);
      kstep_ring<T, WS, MWZ, 4, 5>(acc, f0, f1, slot + 5 * NB * 64, yb, wiso, mwf, bacc, qacc, lane, none, none);
      kstep_ring<T, WS, MWZ, 5, 6>(acc, f1, f0, slot + 6 * NB * 64, yb, wiso, mwf, bacc, qacc, lane, none, none);
      kstep_ring<T, WS, MWZ, 6, 7>(acc, f0, f1, slot + 7 * NB * 64, yb, wiso, mwf, bacc, qacc, lane, none, none);
      // the first fragments of half h+1 (visible since the previous barrier) are requested under the last k-step's MFMAs
      kstep_ring<T, WS, MWZ, 7, 0>(acc, f1, f0, slot_n, yb_n, wiso, mwf, bacc, qacc, lane, none, none);
    } else {
      kstep_ring<T, WS, MWZ, 3, 0>(acc, f1, f0, slot_n, yb_n, wiso, mwf, bacc, qacc, lane, none, none);
    }
    // ---- end of half h: publish half h+2 (its pieces were issued during half h-1), free the slot of half h
    if (h + 1 < nh) {
      if (h + 2 < nh) retire((NH == 4 && more) ? 1 : 0);
#ifndef RING_NOBAR
      __syncthreads();
#endif
    }
  }
}

// =========================================================================================================
// phase 1 (noinline): streaming Gram.  On exit: packed lower triangle of A in LDS (P), b in bvec,
// scr[4] = (y-m)' S (y-m), scr[5] = logdet Sigma_y.
// =========================================================================================================
template <typename T, int NB, int MODE>
BLR_PHASE void phase_gram(char* smem) {
  using C = SmallCfg<T, NB>;
  using acc4 = typename Mfma<T>::acc4;
  T* const slot0 = reinterpret_cast<T*>(smem);
  T* const P = reinterpret_cast<T*>(smem);
  double* const red = reinterpret_cast<double*>(smem);
  T* const ybuf = reinterpret_cast<T*>(smem + C::OFF_Y);
  T* const wbuf = reinterpret_cast<T*>(smem + C::OFF_W);
  T* const bvec = reinterpret_cast<T*>(smem + C::OFF_B);
  T* const mwl = reinterpret_cast<T*>(smem + C::OFF_MW);
  double* const scr = reinterpret_cast<double*>(smem + C::OFF_SCR);
  const RegCtx<T>* ctx = reinterpret_cast<const RegCtx<T>*>(smem + C::OFF_CTX);

  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));  // nothing derived from tid may be hoisted above this phase
  const int lane = tid & 63;
  const int wave = uni(tid >> 6);
  const BLR_GLOBAL T* X = as_global(uni(ctx->X));  // global_load, not flat_load (see as_global)
  const BLR_GLOBAL T* y = as_global(uni(ctx->y));
  const BLR_GLOBAL T* s = as_global(uni(ctx->s));
  const BLR_GLOBAL T* mw = as_global(uni(ctx->mw));
  const BLR_GLOBAL T* Lw = as_global(uni(ctx->Lw));
  const int64_t ldx = uni(ctx->ldx), ldl = uni(ctx->ldl);
  const int D = uni(ctx->D), N = uni(ctx->N);
  const int noise_kind = uni(ctx->noise_kind), prior_kind = uni(ctx->prior_kind);
  const WaveOps<NB> ops = make_wave_ops<T, NB>(wave);

  // accumulators start from the prior precision (dense: UPPER triangle of the caller's matrix, as LAPACK 'U';
  // diagonal tiles get both halves so they stay symmetric)
  const int nprior_stages = (prior_kind == PRIOR_UPPER_FACTOR) ? (D + C::NSC - 1) / C::NSC : 0;
  const bool diag_noise = (noise_kind == NOISE_DIAGONAL);
  const T s_iso = diag_noise ? T(1) : s[0];
  // NB == 8, isotropic noise, no pseudo-observation stages: the stage loop accumulates X X' unscaled and
  // A = Lw + (1 / sigma^2) X X' is formed once at the end (two f64 multiplies fewer per k-step on the matrix pipe's datapath)
  constexpr bool kGlds = (MODE == 4);
  constexpr int kFPG = (1024 / (int)sizeof(T)) / 64;
  // every stage a full tile: scalar-addressed LDS-DMA pieces (stage_glds_full)
  const bool full_tiles = kGlds && (NB % kFPG == 0) && D == C::DP && N > 0 && (N % C::NSC) == 0 &&
                          (3 * ldx + 64) * (int64_t)sizeof(T) < ((int64_t)1 << 31);
  const bool iso_fast = (NB == 8) && !diag_noise && nprior_stages == 0 && full_tiles;
  // prior precision in tile layout (dense: UPPER triangle of the caller's matrix, as LAPACK 'U'; diagonal tiles get both
  // halves so they stay symmetric).  Unconditional loads of clamped addresses, then selects: all of a thread's loads are in
  // flight together (one memory latency, not one per element).
  auto load_prior = [&](acc4 (&pr)[C::TPW]) {
    if (prior_kind == PRIOR_DENSE) {
#pragma unroll
      for (int i = 0; i < C::TPW; ++i) {
        int I, K;
        const bool tile_ok = wave_tile(NB, wave, i, I, K);
        const int col = 16 * K + (lane & 15);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int row = 16 * I + Mfma<T>::crow(lane, v);
          const bool ok = tile_ok && row < D && col < D;
          const int lo = min(row, col), hi = max(row, col);
          const T val = Lw[ok ? (int64_t)hi * ldl + lo : 0];
          pr[i][v] = ok ? val : T(0);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < C::TPW; ++i) {
        int I, K;
        const bool tile_ok = wave_tile(NB, wave, i, I, K);
        const int col = 16 * K + (lane & 15);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int row = 16 * I + Mfma<T>::crow(lane, v);
          const bool ok = tile_ok && I == K && row == col && row < D && prior_kind == PRIOR_DIAGONAL;
          const T val = (I == K) ? Lw[ok ? row : 0] : T(0);  // I == K is wave-uniform: off-diagonal tiles load nothing
          pr[i][v] = ok ? val : T(0);
        }
      }
    }
  };
  acc4 acc[C::TPW];
  if (iso_fast) {
#pragma unroll
    for (int i = 0; i < C::TPW; ++i) acc[i] = acc4{T(0), T(0), T(0), T(0)};
  } else {
    load_prior(acc);
  }
  double bacc[NB];
#pragma unroll
  for (int I = 0; I < NB; ++I) bacc[I] = 0.0;
  if (tid < C::DP) mwl[tid] = tid < D ? mw[tid] : T(0);  // visible after the first stage barrier
  double qacc = 0.0, lacc = 0.0;
  // first observation whose noise variance is not positive (reference :79, _cholesky(Sigma_y) throws PosDefException there)
  int bad_noise = diag_noise ? 0x7fffffff : ((s_iso > T(0)) ? 0x7fffffff : 1);

  const int ndata_stages = (N + C::NSC - 1) / C::NSC;
  const int nstages = nprior_stages + ndata_stages;

  StageRegs<T, kGlds ? 0 : NB> regs;  // MODE 4 keeps only yv / wv in registers
  T* ring_slot = slot0;               // where the data stage being issued lands (MODE 4)
  T* ring_ybuf = ybuf;                // ... and its y values (isotropic fast path: y travels by LDS-DMA too)
  int pending_n0 = 0;                 // first column of the stage whose y / variance sit in regs
  const unsigned voff = glds_lane_offset<T>(ldx, lane);
  auto issue = [&](int td, auto iso_tag) {  // prefetch data stage td: into registers, or (MODE 4) straight into its LDS slot
    constexpr bool ISO = decltype(iso_tag)::value;
    const int n0 = td * C::NSC;
    if constexpr (kGlds) {
      if constexpr (NB % kFPG == 0) {
        if (full_tiles) stage_glds_full<T, NB>(ring_slot, X, ldx, n0, wave, voff);
        else stage_glds<T, NB>(ring_slot, X, ldx, D, N, n0, wave, lane);
      } else {
        stage_glds<T, NB>(ring_slot, X, ldx, D, N, n0, wave, lane);
      }
    } else {
      stage_load<T, NB, MODE>(regs, X, ldx, D, N, n0, tid);
    }
    if constexpr (ISO) {
      // y of the stage: NSC elements = one dword piece (4 bytes per lane), no registers, no compiler-visible load
      static_assert(C::NSC * (int)sizeof(T) / 4 == 64 || C::NSC * (int)sizeof(T) / 4 == 32, "one dword piece");
      if (wave == 0)  // wave-uniform branch
        glds_s<4, C::NSC * (int)sizeof(T) / 4>(uni((int64_t)(uintptr_t)(y + n0)), (unsigned)(lane * 4),
                                               uni((int)lds_addr_of(ring_ybuf)));
    } else {
      // y and the noise variance of the stage travel through registers UNTOUCHED until the stage is consumed: a division or
      // a log right behind the loads makes wave 0 sit out a full memory latency inside issue() -- behind the DMA pieces it
      // has just queued -- with the other three waves waiting for it at the next barrier (measured at D = 64: 1.9 k of the
      // 10.5 k cycles of every stage)
      regs.yv = T(0);
      regs.wv = T(1);
      pending_n0 = n0;
      if (tid < C::NSC && n0 + tid < N) {
        regs.yv = y[n0 + tid];
        if (diag_noise) regs.wv = s[n0 + tid];
      }
    }
  };
  // raw variance -> weight, log-determinant and positivity (reference :79-84); called when the stage is consumed
  const T w_iso_const = T(1) / s_iso;
  auto finish_scalars = [&]() {
    if (tid < C::NSC) {
      const bool valid = pending_n0 + tid < N;
      const T sv = regs.wv;
      if (diag_noise) {
        regs.wv = valid ? T(1) / sv : T(0);                  // :79/:81  Sigma_y^-1 on the diagonal
        if (valid) {
          lacc += log((double)sv);                           // :84  logdet(Sigma_y)
          if (!(sv > T(0))) bad_noise = min(bad_noise, pending_n0 + tid + 1);  // also catches NaN
        }
      } else {
        regs.wv = valid ? w_iso_const : T(0);
      }
    }
  };

  __syncthreads();  // region 0 is free
  auto run_stages = [&](auto ws_tag, auto iso_tag) {
  constexpr int WS = decltype(ws_tag)::value;
  constexpr bool ISO = decltype(iso_tag)::value;
  constexpr bool kPinned = (NB == 8 && WS >= 0);  // pinned-schedule stage (stage8_body)
  constexpr int WSP = WS < 0 ? 0 : WS;
  T mwf[8];
  if constexpr (kPinned) {
#pragma unroll
    for (int I = 0; I < 8; ++I) mwf[I] = mwl[16 * I + (lane & 15)];
  }
  const T wiso = T(1) / s_iso;
  // MODE 4 with prior pseudo-stages: data stage 0 is issued after the last prior stage's barrier instead
  // (its slot is still being used by the prior stages before that)
  if (ndata_stages > 0 && !(kGlds && nprior_stages > 0)) {
    ring_slot = slot0 + (nprior_stages & 1) * C::SLOT;
    ring_ybuf = ybuf + (nprior_stages & 1) * C::NSC;
    issue(0, iso_tag);
  }
  BLR_GSTAMP_INIT;
  for (int t = 0; t < nstages; ++t) {
    const int sl = t & 1;
    T* slot = slot0 + sl * C::SLOT;
    const bool is_data = t >= nprior_stages;
    BLR_GSTAMP(4);
    if (!is_data) {
      // prior pseudo-observations (PDMat / carried-forward factor): a handful of stages, loaded synchronously
      StageRegs<T, NB> pr;
      const int n0 = t * C::NSC;
      stage_load<T, NB, 2>(pr, Lw, ldl, D, D, n0, tid);
      pr.yv = T(0);
      pr.wv = (tid < C::NSC && n0 + tid < D) ? T(1) : T(0);
      stage_store<T, NB, 2>(pr, slot, ybuf + sl * C::NSC, wbuf + sl * C::NSC, tid);
    } else if constexpr (kGlds) {
      if constexpr (!ISO) {
        finish_scalars();
        if (tid < C::NSC) { ybuf[sl * C::NSC + tid] = regs.yv; wbuf[sl * C::NSC + tid] = regs.wv; }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's LDS-DMA pieces of stage t have landed
    } else {
      finish_scalars();
      stage_store<T, NB, MODE>(regs, slot, ybuf + sl * C::NSC, wbuf + sl * C::NSC, tid);
    }
    BLR_GSTAMP(0);
    __syncthreads();
    BLR_GSTAMP(1);
    // pinned schedule: the first k-step's fragments are requested before the next stage's DMA pieces are issued
    [[maybe_unused]] KF8<T, WSP> f0;
    if constexpr (kPinned) stage8_prologue<T, WSP, ISO>(f0, slot, ybuf + sl * C::NSC, wbuf + sl * C::NSC, lane);
    const int tdn = t + 1 - nprior_stages;  // next data stage
#if BLR_EXP != 4
    const int tdn_min = (kGlds && nprior_stages > 0) ? 0 : 1;
    if (tdn >= tdn_min && tdn < ndata_stages) {  // in flight while this stage computes
      ring_slot = slot0 + ((t + 1) & 1) * C::SLOT;
      ring_ybuf = ybuf + ((t + 1) & 1) * C::NSC;
      issue(tdn, iso_tag);
    }
#endif
    BLR_GSTAMP(2);
    if constexpr (kPinned) {
      __builtin_amdgcn_sched_barrier(0);
      stage8_body<T, WSP, ISO>(f0, slot, ybuf + sl * C::NSC, wbuf + sl * C::NSC, acc, bacc, qacc, mwf, wiso, lane,
                               is_data ? T(1) : T(0));
    } else {
      compute_stage<T, NB, WS>(slot, ybuf + sl * C::NSC, wbuf + sl * C::NSC, ops, acc, bacc, qacc, mwl, wave, lane, is_data);
    }
    BLR_GSTAMP(3);
  }
  };
  // NB == 8 (the MFMA-bound headline shape) and NB == 4 (config 4, D = 64): one copy of the stage loop per wave,
  // wave index static
  if constexpr (NB == 8) {
    if (iso_fast) {
      if constexpr (kGlds) {
        T mwf[8];
#pragma unroll
        for (int I = 0; I < 8; ++I) mwf[I] = mwl[16 * I + (lane & 15)];
        const T wiso = T(1) / s_iso;
#ifdef BLR_GRAM_STAMPS
        const unsigned long long ck0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
        bool mwz = true;
#pragma unroll
        for (int I = 0; I < 8; ++I) mwz = mwz && (mwf[I] == T(0));
        mwz = __all(mwz);  // the 64 lanes of a wave hold all 128 entries of mw: wave-uniform, and the same in every wave
        auto ring = [&](auto ws, auto mz) {
          gram_iso_ring<T, decltype(ws)::value, decltype(mz)::value>(slot0, ybuf, X, y, ldx, N, voff, lane, acc, bacc, qacc, mwf, wiso);
        };
        auto ring_w = [&](auto mz) {
          switch (wave) {
            case 0: ring(std::integral_constant<int, 0>{}, mz); break;
            case 1: ring(std::integral_constant<int, 1>{}, mz); break;
            case 2: ring(std::integral_constant<int, 2>{}, mz); break;
            default: ring(std::integral_constant<int, 3>{}, mz); break;
          }
        };
        if (mwz) ring_w(std::true_type{});
        else ring_w(std::false_type{});
#ifdef BLR_GRAM_STAMPS
        if ((blockIdx.x & 255) == 0 && lane == 0) {  // whole-loop cycles and 100 MHz ticks: the clock the chip holds under this load
          g_gstamps[wave][5] += __builtin_amdgcn_s_memtime() - ck0;
          g_gstamps[wave][6] += __builtin_amdgcn_s_memrealtime() - rt0;
          g_gstamps[wave][7] += 1;
        }
        if (wave == 0 && lane == 0 && blockIdx.x < 8192) {
          g_wgclk[blockIdx.x][0] = __builtin_amdgcn_s_memtime() - ck0;
          g_wgclk[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime() - rt0;
        }
#endif
      }
      {
        acc4 pr[C::TPW];
        load_prior(pr);
        const T winv = T(1) / s_iso;
#pragma unroll
        for (int i = 0; i < C::TPW; ++i)
#pragma unroll
          for (int v = 0; v < 4; ++v) acc[i][v] = pr[i][v] + acc[i][v] * winv;
      }
    } else {
      switch (wave) {
        case 0: run_stages(std::integral_constant<int, 0>{}, std::false_type{}); break;
        case 1: run_stages(std::integral_constant<int, 1>{}, std::false_type{}); break;
        case 2: run_stages(std::integral_constant<int, 2>{}, std::false_type{}); break;
        default: run_stages(std::integral_constant<int, 3>{}, std::false_type{}); break;
      }
    }
  } else if constexpr (NB == 4) {
    switch (wave) {
      case 0: run_stages(std::integral_constant<int, 0>{}, std::false_type{}); break;
      case 1: run_stages(std::integral_constant<int, 1>{}, std::false_type{}); break;
      case 2: run_stages(std::integral_constant<int, 2>{}, std::false_type{}); break;
      default: run_stages(std::integral_constant<int, 3>{}, std::false_type{}); break;
    }
  } else {
    run_stages(std::integral_constant<int, -1>{}, std::false_type{});
  }

  // b partials -> LDS -> fixed-order sum
  __syncthreads();
  {
    const int q = lane >> 4, r = lane & 15;
#pragma unroll
    for (int I = 0; I < NB; ++I) red[(wave * 4 + q) * C::DP + 16 * I + r] = bacc[I];
  }
  __syncthreads();
  if (tid < C::DP) {
    double sum = 0.0;
#pragma unroll
    for (int p = 0; p < 16; ++p) sum += red[p * C::DP + tid];  // fixed order
    bvec[tid] = (T)sum;                                         // b = X S (y - X'mw)   (:57 Bt'dy, unwhitened)
  }
  const double quad = block_allreduce(qacc, scr, tid);          // (y-m)' S (y-m)
  double logdet_Sy = block_allreduce(lacc, scr, tid);
  if (!diag_noise) logdet_Sy = (double)N * log((double)s_iso);
  // (block_allreduce's barriers also fence the reads of `red` above)
  if (tid == 0) { scr[4] = quad; scr[5] = logdet_Sy; }
  {
    int* const iscr = reinterpret_cast<int*>(smem + C::OFF_SCR + 64);
    bad_noise = block_min_int(bad_noise, iscr, tid);
    if (tid == 0) iscr[6] = bad_noise;  // read back by the kernel after the phase's final barrier
  }

  // A: accumulators -> packed lower triangle
#pragma unroll
  for (int i = 0; i < C::TPW; ++i) {
    int I, K;
    if (wave_tile(NB, wave, i, I, K)) {
      const int col = 16 * K + (lane & 15);
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int row = 16 * I + Mfma<T>::crow(lane, v);
        if (col <= row) P[pidx(row, col)] = acc[i][v];
      }
    }
  }
  __syncthreads();
}

// =========================================================================================================
// phase 2 (noinline): blocked Cholesky of the packed lower triangle P, trailing matrix in MFMA accumulators.
// For each block column J:
//   (a) panel J is in P already (stored by the owners of tiles (I, J) at the end of the previous trailing update);
//   (b) every wave loads panel rows, ONE ROW PER LANE (lanes 0-15: the 16 diagonal-block rows, held
//       redundantly by all four waves so no cross-wave traffic is needed; lanes 16-63: 48 rows below),
//       and eliminates the 16 columns in registers -- pivots and multipliers travel by v_readlane;
//       the right-hand side b rides along (forward substitution u = L^-1 b for free);
//   (c) the finished panel goes back to P and all waves apply  A_IK -= L_IJ L_KJ'  to their
//       remaining tiles with 4 MFMAs per tile, reading both operands from P in fragment order.
// Two barriers per block column, no LDS round trip for the trailing matrix.  On exit P holds L (A = L L'),
// bvec holds u = L^-1 b.  Returns 0 or the LAPACK-style 1-based index of the failing leading minor.
// =========================================================================================================
// The factorisation is bound by INSTRUCTION ISSUE, not by arithmetic: one wave runs ~5 cycles per instruction here, and the
// first version spent ~1700 instructions per 16-column panel (540 in the elimination proper, 360 in predicated write-back
// stores, 670 in the trailing update -- packed-triangle index arithmetic per fragment element and an exec-mask branch per
// predicated store).  This form keeps the algorithm and removes the bookkeeping:
//   * a fragment address is  pidx(16 I, 0) [scalar] + 16 I r + pidx(r, 0) [one multiply-add per tile side] + an immediate;
//   * rows are loaded WITHOUT masking -- the entries to the right of the diagonal of a diagonal-block row are dead values
//     (never stored, never read by another lane), so whatever the packed triangle holds there is harmless;
//   * predicated stores go to a per-lane dummy word (the dinv area, unused until the back substitution) through an address
//     select instead of an exec-mask branch each;
//   * tiles are skipped by scalar branches on wave-uniform tile coordinates.
// TAG: a caller compiled for a different register budget (fused_i8_kernel: one wave per SIMD) instantiates its OWN copy -- a
// noinline function is compiled once per instantiation for the largest budget among its callers (traps)
template <typename T, int NB, int TAG = 0>
BLR_PHASE int phase_chol(char* smem, int D_in, int with_rhs_in) {
  using C = SmallCfg<T, NB>;
  using acc4 = typename Mfma<T>::acc4;
  constexpr int TPW = C::TPW;
  T* const P = reinterpret_cast<T*>(smem);
  T* const bvec = reinterpret_cast<T*>(smem + C::OFF_B);
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int wave = uni(tid >> 6);
  // (opaque to interprocedural constant propagation: with ONE caller that passes D = 128 -- the int8 kernel's translation unit --
  //  hipcc specialises and unrolls this function into three times the code with 270 scratch accesses)
  int D = uni(D_in);
  asm volatile("" : "+s"(D));
  int with_rhs_i = uni(with_rhs_in);
  asm volatile("" : "+s"(with_rhs_i));
  const bool with_rhs = with_rhs_i != 0;
  const int nblk = (D + 15) >> 4;
  const int r = lane & 15, q = lane >> 4;
  const int pr = (r * (r + 1)) >> 1;  // pidx(r, 0)
  T* const dummy = reinterpret_cast<T*>(smem + C::OFF_DINV) + r;
  int cr[4], pcr[4];  // C-layout rows of this lane inside a tile, and pidx(cr, 0)
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    cr[v] = Mfma<T>::crow(lane, v);
    pcr[v] = (cr[v] * (cr[v] + 1)) >> 1;
  }

  // tiles from P (diagonal tiles: mirror the lower half so the tile is symmetric); wave-uniform tile coordinates
  acc4 acc[TPW];
  int tI[TPW], tK[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    int I, K;
    const bool tile_ok = wave_tile(NB, wave, i, I, K);
    tI[i] = uni(I);
    tK[i] = uni((tile_ok && I < nblk) ? K : -1);  // -1: never updated, never stored
    const int col = 16 * K + r;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int row = 16 * I + cr[v];
      acc[i][v] = tile_ok ? P[pidx(max(row, col), min(row, col))] : T(0);
    }
  }

  int info = 0;
  BLR_STAMP_INIT;
  BLR_STAMP(0);
  for (int J = 0; J < nblk; ++J) {
    // (a) panel J is already in P: column 0 was never updated, and step (c) of iteration J-1 stored column J right
    //     after updating it -- one barrier per block column instead of two
    __syncthreads();
    BLR_STAMP(1);
    // (b) one row per lane: lanes 0-15 the diagonal-block rows (redundantly in all four waves), lanes 16-63 rows below
    const bool is_diag = lane < 16;
    const int ri = is_diag ? 16 * J + lane : 16 * (J + 1) + 48 * wave + (lane - 16);
    const bool active = ri < D;
    const int ria = active ? ri : 0;
    const int ncols = min(16, D - 16 * J);
    T* const rowp = P + (((ria * (ria + 1)) >> 1) + 16 * J);
    T arow[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) arow[c] = rowp[c];  // unmasked: see the header comment
    T bl = T(0);
    if (with_rhs) bl = bvec[ria];
    BLR_STAMP(2);
    // Elimination with DEFERRED scaling: column c stays unscaled while it is being used (multiplier
    // t = a_ic / d2, update a_ik -= t * a_kc, b_i -= t * b_c), so the serial chain per column is
    // readlane -> reciprocal -> multiply -> fma; the square root is taken off the critical path afterwards.
    T own_rsq = T(1);
    if (ncols == 16) {  // (scalar) a full panel -- every panel when 16 | D
      // One basic block, software-pipelined by hand: the serial chain of a column is  multiplier -> update of column c + 1 -> its pivot by
      // readlane -> reciprocal (five dependent operations), and the other 14 - c updates of column c are independent of it -- so the next
      // pivot's reciprocal is started right after the first update and the rest of the updates fill its latency.  The square roots leave the
      // loop altogether: the pivots stay unscaled in lane c's arow[c], ONE reciprocal square root over the 16 lanes afterwards, and the
      // columns are scaled by readlane.  Operation for operation the same arithmetic as the generic loop below (bit-identical L, u);
      // as 16 per-column blocks with the rsqrt chain inside each, the elimination was 410 cycles per column.
      T d2 = readlane(arow[0], 0);
      T rc = fast_rcp(d2);
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        if (!(d2 > T(0))) {  // wave-uniform (SGPR) and identical in all four waves
          if (info == 0) info = 16 * J + c + 1;
        }
        const T t = arow[c] * rc;
        if (c + 1 < 16) {
          arow[c + 1] = fused_madd(-t, readlane(arow[c], c + 1), arow[c + 1]);
          d2 = readlane(arow[c + 1], c + 1);
          rc = fast_rcp(d2);
        }
        const T bc = readlane(bl, c);
        if (lane > c) bl = fused_madd(-t, bc, bl);
#pragma unroll
        for (int k = c + 2; k < 16; ++k) arow[k] = fused_madd(-t, readlane(arow[c], k), arow[k]);  // unscaled A[16J + k][16J + c]
      }
      T piv = T(1);
#pragma unroll
      for (int c = 0; c < 16; ++c) piv = (lane == c) ? arow[c] : piv;
      own_rsq = fast_rsqrt(piv);
#pragma unroll
      for (int c = 0; c < 16; ++c) arow[c] *= readlane(own_rsq, c);  // L[i][c] = a_ic / sqrt(d2); lane c: d2 / sqrt(d2)
    } else {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      if (c < ncols) {
        const T d2 = readlane(arow[c], c);
        if (!(d2 > T(0))) {  // wave-uniform (SGPR) and identical in all four waves
          if (info == 0) info = 16 * J + c + 1;
        }
        const T t = arow[c] * fast_rcp(d2);
        const T bc = readlane(bl, c);
        if (lane > c) bl = fused_madd(-t, bc, bl);
#pragma unroll
        for (int k = c + 1; k < 16; ++k) {
          const T akc = readlane(arow[c], k);  // unscaled A[16J + k][16J + c]
          arow[k] = fused_madd(-t, akc, arow[k]);
        }
        const T rsq = fast_rsqrt(d2);
        if (lane == c) own_rsq = rsq;
        arow[c] = (lane == c) ? d2 * rsq : arow[c] * rsq;  // L[i][c] = a_ic / sqrt(d2); diagonal = sqrt(d2)
      }
    }
    }
    if (is_diag) bl *= own_rsq;  // u_c = b_c / L_cc for the diagonal-block rows
    BLR_STAMP(3);
    if (info != 0) break;  // uniform across the block: every wave factors the same diagonal rows
    {
      const int lim = active ? (is_diag ? lane : 15) : -1;  // columns 0 .. lim of this lane's row are stored
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        T* dst = (c <= lim) ? rowp + c : dummy;
        *dst = arow[c];
      }
      if (with_rhs && active && (!is_diag || wave == 0)) bvec[ri] = bl;
    }
    __syncthreads();
    BLR_STAMP(4);
    // (c) trailing update from the finished panel; block column J + 1 is final after it and goes straight back to P
    const int pc = pr + 16 * J + q;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      if (tK[i] > J) {  // scalar branch
        const int I = tI[i], K = tK[i];
        const T* pI = P + ((128 * I * I + 8 * I) + (16 * I) * r + pc);  // pidx(16 I + r, 16 J + q)
        const T* pK = P + ((128 * K * K + 8 * K) + (16 * K) * r + pc);
        T fa[4], fb[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { fa[ks] = pI[4 * ks]; fb[ks] = pK[4 * ks]; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc[i] = Mfma<T>::mma(-fa[ks], fb[ks], acc[i]);
        if (K == J + 1) {  // scalar
          const int sb = (128 * I * I + 8 * I) + 16 * K + r;
          if (I != K) {
#pragma unroll
            for (int v = 0; v < 4; ++v) P[sb + (16 * I) * cr[v] + pcr[v]] = acc[i][v];
          } else {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              T* dst = (r <= cr[v]) ? P + (sb + (16 * I) * cr[v] + pcr[v]) : dummy;
              *dst = acc[i][v];
            }
          }
        }
      }
    }
    BLR_STAMP(5);
  }
  __syncthreads();
  BLR_STAMP_FLUSH;
  return info;
}

// =========================================================================================================
// phase 3 (noinline, wave 0 does the serial part): m = L^-T u by column-oriented back substitution.
// Returns through LDS: bvec <- m (rows < D), scr[6] = |u|^2, scr[7] = logdet A.
// =========================================================================================================
#ifdef BLR_I8_STAMPS
__device__ unsigned long long g_i8stamps[8][20];  // (tools/i8_gram.hip; slots 12 - 15: inside phase_backsolve)
#define BLR_BS_STAMP(slot) do { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); if ((blockIdx.x % 257) == 0 && (threadIdx.x & 63) == 0) atomicAdd(&g_i8stamps[threadIdx.x >> 6][slot], t__ - bs_prev); bs_prev = t__; } while (0)
#define BLR_BS_STAMP_DECL unsigned long long bs_prev = __builtin_amdgcn_s_memtime()
#else
#define BLR_BS_STAMP(slot) do {} while (0)
#define BLR_BS_STAMP_DECL do {} while (0)
#endif
template <typename T, int NB, int TAG = 0>
BLR_PHASE void phase_backsolve(char* smem, int D_in, T* Tout_in, int64_t ldt_in) {
  using C = SmallCfg<T, NB>;
  T* const P = reinterpret_cast<T*>(smem);
  T* const bvec = reinterpret_cast<T*>(smem + C::OFF_B);
  T* const dinv = reinterpret_cast<T*>(smem + C::OFF_DINV);
  double* const scr = reinterpret_cast<double*>(smem + C::OFF_SCR);
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int wave = uni(tid >> 6);
  int D = uni(D_in);
  asm volatile("" : "+s"(D));  // (see phase_chol)
  BLR_GLOBAL T* const Tout = as_global(uni(Tout_in));
  const int64_t ldt = uni(ldt_in);
  BLR_BS_STAMP_DECL;
  if (tid < D) dinv[tid] = T(1) / P[pidx(tid, tid)];
  __syncthreads();
  BLR_BS_STAMP(12);
  if (wave != 0) {
    // T = L' (upper, column-major; strictly-lower part zero) goes out while wave 0 runs the serial substitution: the
    // three waves would otherwise sit at the closing barrier (:67, chol(...).U)
    if (Tout != nullptr) {
      for (int c = wave - 1; c < D; c += kWaves - 1) {
        const T* row = P + pidx(c, 0);
        BLR_GLOBAL T* out = Tout + (int64_t)c * ldt;
        const int r0 = lane, r1 = lane + 64;
        const T v0 = (r0 <= c) ? row[r0] : T(0);
        const T v1 = (r1 <= c) ? row[r1] : T(0);
        if (r0 < D) out[r0] = v0;
        if (r1 < D) out[r1] = v1;
      }
    }
  } else {
    const int i0 = lane, i1 = lane + 64;
    T b0 = i0 < D ? bvec[i0] : T(0);
    T b1 = i1 < D ? bvec[i1] : T(0);
    double uu = (double)b0 * (double)b0 + (double)b1 * (double)b1;
    uu = wave_allreduce(uu);
    // Column-oriented back substitution, 8 pivots per block: the 8 rows of L the block needs are loaded up front
    // (they do not depend on the running solution), so the serial chain per pivot is readlane -> multiply -> fma
    // with no LDS latency in it.  Reciprocal pivots live in registers (one per owned row).
    const T r0 = i0 < D ? dinv[i0] : T(0);
    const T r1 = i1 < D ? dinv[i1] : T(0);
    for (int kb = (BLR_EXP == 5 ? -1 : D - 1); kb >= 0; kb -= 8) {
      T row0[8], row1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = kb - u;
        const bool kv = k >= 0;
        const T* row = P + pidx(kv ? k : 0, 0);
        row0[u] = (kv && i0 < k) ? row[i0] : T(0);
        row1[u] = (kv && i1 < k) ? row[i1] : T(0);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = kb - u;
        if (k >= 0) {
          const bool lo = k < 64;
          const T mk = readlane(lo ? b0 : b1, k & 63) * readlane(lo ? r0 : r1, k & 63);
          if (lane == (k & 63)) { if (lo) b0 = mk; else b1 = mk; }
          b0 -= row0[u] * mk;
          b1 -= row1[u] * mk;
        }
      }
    }
    if (i0 < D) bvec[i0] = b0;
    if (i1 < D) bvec[i1] = b1;
    BLR_BS_STAMP(13);
    double ld = 0.0;
    if (i0 < D) ld += log((double)P[pidx(i0, i0)]);
    if (i1 < D) ld += log((double)P[pidx(i1, i1)]);
    ld = 2.0 * wave_allreduce(ld);  // logdet A
    if (lane == 0) { scr[6] = uu; scr[7] = ld; }
  }
  BLR_BS_STAMP(14);
  __syncthreads();
  BLR_BS_STAMP(15);
}

// ---- scalar right-looking Cholesky on the packed triangle (standalone chol_small_kernel only) --------------
template <typename T>
__device__ __forceinline__ int chol_packed(T* __restrict__ P, T* __restrict__ dinv, int D, int tid) {
  const int ti = tid >> 4, tk = tid & 15;
  int info = 0;
  __syncthreads();
  for (int j = 0; j < D; ++j) {
    const T ajj = P[pidx(j, j)];
    if (!(ajj > T(0))) { info = j + 1; break; }  // wave- and block-uniform: everyone reads the same word
    for (int i = j + 1 + ti; i < D; i += 16) {
      const T ci = P[pidx(i, j)] / ajj;
      T* row = P + pidx(i, 0);
      for (int k = j + 1 + tk; k <= i; k += 16) row[k] -= ci * P[pidx(k, j)];
    }
    __syncthreads();
  }
  if (info) return info;
  if (tid < D) dinv[tid] = sqrt(P[pidx(tid, tid)]);
  __syncthreads();
  for (int i = ti; i < D; i += 16) {
    T* row = P + pidx(i, 0);
    for (int k = tk; k < i; k += 16) row[k] /= dinv[k];
  }
  if (tid < D) P[pidx(tid, tid)] = dinv[tid];
  __syncthreads();
  return 0;
}

// ---- the kernel -----------------------------------------------------------------------------------
// Everything between the phases is a phase too ("glue"): the phases keep no callee-saved registers (BLR_PHASE), so whatever the KERNEL
// holds across a call goes to scratch -- with the glue inline that was 220 B per lane of loop invariants and live values (612 B while the
// phases still saved the ABI's callee-saved registers).  The kernel keeps the regressor index and the kernarg pointer, in scalar
// registers; the arguments are read from the kernarg segment where they are used (constant cache), `logdet Lw` waits in the LDS context.
template <typename T>
using KernArgPtr = const __attribute__((address_space(4))) PosteriorArgs<T>*;

// prior (reference :78) + per-regressor context.  Returns 0: go on, 1: this regressor is finished (skipped, or failed and reported)
template <typename T, int NB>
BLR_PHASE int glue_prior(char* smem, KernArgPtr<T> ap, int reg) {
  using C = SmallCfg<T, NB>;
  const __attribute__((address_space(4))) PosteriorArgs<T>& a = *ap;  // (the kernarg segment: scalar loads where a field is used)
  T* const P = reinterpret_cast<T*>(smem);
  double* const scr = reinterpret_cast<double*>(smem + C::OFF_SCR);
  int* const iscr = reinterpret_cast<int*>(smem + C::OFF_SCR + 64);
  RegCtx<T>* ctx = reinterpret_cast<RegCtx<T>*>(smem + C::OFF_CTX);
  const int tid = threadIdx.x;
  const int D = a.D;
  const double kNaN = __longlong_as_double(0x7ff8000000000000LL);
  if (a.retry_only) {
    if (a.info[reg] != kI8RetryCode) return 1;  // (uniform; the int8 kernel finished this regressor)
    if (tid == 0 && a.i8_handed_slice != nullptr) {
      atomicAdd(a.i8_handed_slice, 1ull);
      atomicAdd(a.i8_handed_tot, 1ull);
    }
  }
  const T* Lw = a.Lw + (int64_t)reg * a.strideLw;
  __syncthreads();  // previous regressor fully done with LDS
  if (tid == 0) {
    ctx->X = a.X + (int64_t)reg * a.strideX;
    ctx->y = a.y + (int64_t)reg * a.stridey;
    ctx->s = a.s + (int64_t)reg * a.strides;
    ctx->mw = a.mw + (int64_t)reg * a.stridemw;
    ctx->Lw = Lw;
    ctx->ldx = a.ldx;
    ctx->ldl = a.ldl;
    ctx->D = D;
    ctx->N = a.N;
    ctx->noise_kind = a.noise_kind;
    ctx->prior_kind = a.prior_kind;
  }
  int info = 0;
  double logdet_Lw = 0.0;
  if (a.prior_kind == PRIOR_DENSE) {
    // upper triangle (k <= i) of column i; columns over waves, rows over lanes: no per-element integer division
    for (int i = tid >> 6; i < D; i += kWaves)
      for (int k = tid & 63; k <= i; k += kWave) P[pidx(i, k)] = Lw[(int64_t)i * a.ldl + k];
    for (int idx = D * (D + 1) / 2 + tid; idx < C::PACKED; idx += kThreads) P[idx] = T(0);  // padded rows
    __syncthreads();
    info = phase_chol<T, NB>(smem, D, 0);  // :78
    double v = (info == 0 && tid < D) ? log((double)P[pidx(tid, tid)]) : 0.0;
    logdet_Lw = 2.0 * block_allreduce(v, scr, tid);
  } else {
    // diagonal entries of d (DIAGONAL) or of the factor U (UPPER_FACTOR) must be positive
    double v = 0.0;
    int bad = 0x7fffffff;
    if (tid < D) {
      T dv = (a.prior_kind == PRIOR_DIAGONAL) ? Lw[tid] : Lw[(int64_t)tid * a.ldl + tid];
      if (dv > T(0)) v = log((double)dv);
      else bad = tid + 1;
    }
    bad = block_min_int(bad, iscr, tid);
    if (bad != 0x7fffffff) info = bad;
    v = block_allreduce(v, scr, tid);
    logdet_Lw = (a.prior_kind == PRIOR_DIAGONAL) ? v : 2.0 * v;
  }
  if (info != 0) {  // block-uniform
    if (tid == 0) {
      a.info[reg] = info;
      if (a.logpdf) a.logpdf[reg] = kNaN;
    }
    return 1;
  }
  if (tid == 0) ctx->logdet_Lw = logdet_Lw;
  __syncthreads();
  return 0;
}

// after the Gram phase: the noise check of reference :79 and the posterior precision (:92).  Returns 1 when the regressor is finished.
template <typename T, int NB>
BLR_PHASE int glue_after_gram(char* smem, KernArgPtr<T> ap, int reg) {
  using C = SmallCfg<T, NB>;
  const __attribute__((address_space(4))) PosteriorArgs<T>& a = *ap;
  T* const P = reinterpret_cast<T*>(smem);
  int* const iscr = reinterpret_cast<int*>(smem + C::OFF_SCR + 64);
  const int tid = threadIdx.x;
  const int D = a.D;
  if (iscr[6] != 0x7fffffff) {  // Sigma_y is not positive definite (block-uniform): PosDefException(index), as :79 would throw
    if (tid == 0) {
      a.info[reg] = iscr[6];
      if (a.logpdf) a.logpdf[reg] = __longlong_as_double(0x7ff8000000000000LL);
    }
    return 1;
  }
  if (a.Lw_post) {  // posterior precision Lw' = A, full symmetric (:92)
    T* out = a.Lw_post + (int64_t)reg * a.strideLp;
    for (int c = tid >> 6; c < D; c += kWaves)
      for (int r = tid & 63; r < D; r += kWave) out[(int64_t)c * a.ldlp + r] = (r >= c) ? P[pidx(r, c)] : P[pidx(c, r)];
  }
  return 0;
}

// the regressor's status, and what goes out when it succeeded: mw' = mw + m (:68), the evidence (:84 + :57)
template <typename T, int NB>
BLR_PHASE void glue_finish(char* smem, KernArgPtr<T> ap, int reg, int info) {
  using C = SmallCfg<T, NB>;
  const __attribute__((address_space(4))) PosteriorArgs<T>& a = *ap;
  T* const bvec = reinterpret_cast<T*>(smem + C::OFF_B);
  double* const scr = reinterpret_cast<double*>(smem + C::OFF_SCR);
  RegCtx<T>* ctx = reinterpret_cast<RegCtx<T>*>(smem + C::OFF_CTX);
  const int tid = threadIdx.x;
  if (info != 0) {
    if (tid == 0) {
      a.info[reg] = info;
      if (a.logpdf) a.logpdf[reg] = __longlong_as_double(0x7ff8000000000000LL);
    }
    return;
  }
  if (a.mw_post && tid < a.D) a.mw_post[(int64_t)reg * a.stride_mwpost + tid] = (a.mw + (int64_t)reg * a.stridemw)[tid] + bvec[tid];  // :68
  if (tid == 0) {
    a.info[reg] = 0;
    if (a.logpdf) {
      const double LOG2PI = 1.8378770664093454835606594728112;
      a.logpdf[reg] = -0.5 * ((double)a.N * LOG2PI + scr[5] + scr[4] + scr[7] - ctx->logdet_Lw - scr[6]);  // :84 + :57
    }
  }
}

// Occupancy target: D <= 64 (NB <= 4) is HBM/latency-bound -- four workgroups per CU
// (<= 128 registers, ~35 KB of LDS each) keep 4 x 16 KB of LDS-DMA in flight per CU; D > 64 is MFMA-bound and
// needs the registers for accumulators: two workgroups per CU.
template <typename T, int NB, int MODE /* data loader: 0 ColVecs generic, 1 RowVecs, 3 ColVecs vector regs, 4 ColVecs LDS-DMA */>
__global__ __launch_bounds__(kThreads, (NB <= 4 ? 4 : (sizeof(T) == 4 ? BLR_F32_WAVES_PER_SIMD : 2))) void fused_small_kernel(PosteriorArgs<T> a_kernarg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const KernArgPtr<T> ap = (KernArgPtr<T>)__builtin_amdgcn_kernarg_segment_ptr();
  const __attribute__((address_space(4))) PosteriorArgs<T>& a = *ap;
  if (a.retry_only) {
    // follow-up of fused_i8_kernel: nearly always there is nothing to redo.  The workgroup looks at the status words of ALL its
    // regressors at once (one memory round trip) and leaves; one dependent load per regressor inside the loop below, on a grid of
    // one workgroup per regressor with 70 KB of LDS each, made this launch 12 us of a 4.1 ms step.
    int any = 0;
    for (int reg = blockIdx.x + threadIdx.x * gridDim.x; reg < a.B; reg += kThreads * gridDim.x) any |= (a.info[reg] == kI8RetryCode) ? 1 : 0;
    if (!__syncthreads_or(any)) return;
  }
  for (int reg = blockIdx.x; reg < a.B; reg += gridDim.x) {
    // ---- phase 0: prior -----------------------------------------------------------------------
    if (glue_prior<T, NB>(smem, ap, reg) != 0) continue;
    BLR_PSTAMP_INIT;
    // ---- phase 1: streaming Gram -> P, bvec, scr[4..5] ------------------------------------------------
    BLR_PSTAMP(0);
    phase_gram<T, NB, MODE>(smem);
    BLR_PSTAMP(1);
#if BLR_EXP >= 1 && BLR_EXP <= 4
    if (threadIdx.x == 0) {
      a.info[reg] = 0;
      if (a.logpdf) a.logpdf[reg] = reinterpret_cast<double*>(smem + SmallCfg<T, NB>::OFF_SCR)[4] + (double)reinterpret_cast<T*>(smem)[0];
    }
    continue;
#endif
    if (glue_after_gram<T, NB>(smem, ap, reg) != 0) continue;
    // ---- phase 2: blocked Cholesky + fused forward substitution ------------------------------------------
    BLR_PSTAMP(2);
    const int info = phase_chol<T, NB>(smem, a.D, 1);  // :86; T = L' is chol(Lw + G).U (:67)
    BLR_PSTAMP(3);
    // ---- phase 3: back substitution (wave 0) with T = L' written by the other three waves, evidence ------
    BLR_PSTAMP(4);
    if (info == 0) phase_backsolve<T, NB>(smem, a.D, a.T_post ? a.T_post + (int64_t)reg * a.strideT : (T*)nullptr, a.ldt);
    BLR_PSTAMP(5);
    glue_finish<T, NB>(smem, ap, reg, info);
    BLR_PSTAMP(6);
#ifdef BLR_GRAM_STAMPS
#ifdef BLR_PSTAMP_ALL
    if (threadIdx.x == 0) atomicAdd(&g_pstamps[15], 1ull);
#else
    if (blockIdx.x == 0 && threadIdx.x == 0) g_pstamps[15] += 1;
#endif
#endif
  }
}

}  // namespace blr
