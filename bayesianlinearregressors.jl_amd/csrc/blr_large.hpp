// Large-D path (D > 128): the same direct Gram form as blr_fused_small.hpp, split over workgroups.
//
//   colstats_kernel      mu_n = x_n'mw, r_n = (y_n - mu_n)/s_n, partial q = sum delta r, l = sum log s   (:82-84)
//   gram_tile_kernel     128x128 macro tile of  sum_n x_n w_n x_n'  over one N-slice (split-K), MFMA 16x16x4,
//                        X staged in fragment order by LDS-DMA; diagonal tiles also give  b_I = X_I r      (:86, :57)
//   gram_reduce_kernel   fixed-order sum of the split partials + prior precision -> augmented matrix Abar
//   panel_chain_kernel   (blr_panel.hpp) wave-specialised Cholesky of one 128x128 diagonal block with 64 rows of the
//                        block column below riding along per workgroup: L_pp and X <- X L_pp^-T in ONE launch
//   trsm_block_kernel    X <- X L_pp^-T for row blocks against an already factored L_pp (tall-matrix sweeps of the marginal /
//                        gradient / multi-output paths): left-looking 16-column chunks on MFMA
//   (trailing update)    gram_tile_kernel again, X := the finished panel of L, subtracting in place
//   backsolve_wave_kernel  m = L^-T u as a wavefront over the row blocks (one workgroup each), logdet A, |u|^2,
//                        posterior mean, evidence
//   transpose_out_kernel T = L' (upper, column-major) for the caller
//
// Abar is (DP + 128) x DP, column-major, ld = DP + 128, DP = 128 ceil(D/128): rows [0, DP) hold the lower
// triangle of A (padding: unit diagonal), row DP holds b' -- the right-hand side is carried as one more ROW
// of the matrix, so the panel TRSM and the trailing updates perform the forward substitution u = L^-1 b
// without any extra kernel (same trick as the fused small-D kernel).
#pragma once
#include "blr_aux_kernels.hpp"
#include "blr_fused_small.hpp"
#include "blr_panel.hpp"

namespace blr {

constexpr int kPB = 128;  // panel / macro-tile edge

// Gram launch geometry (f32): k-steps per stage and workgroups per CU the kernel is compiled for
#ifndef BLR_GRAM_KS_F32
#define BLR_GRAM_KS_F32 8
#endif
#ifndef BLR_GRAM_WGS
#define BLR_GRAM_WGS 2
#endif
template <typename T>
struct LargeCfg {
  static constexpr int KS = (sizeof(T) == 4) ? BLR_GRAM_KS_F32 : 4;   // k-steps per stage (32 / 16 columns)
  static constexpr int NSC = 4 * KS;
  static constexpr int SIDE = KS * 8 * 64;              // elements of one operand side per slot
  static constexpr int SLOT = 2 * SIDE;                 // A side + B side
  static constexpr int OFF_W = 2 * SLOT * (int)sizeof(T);          // wbuf[2][NSC], rbuf[2][NSC]
  static constexpr int OFF_R = OFF_W + 2 * NSC * (int)sizeof(T);
  static constexpr int OFF_RED = 0;  // 16 x 128 doubles of b partials: aliases the slots (used after the last stage)
  static constexpr int LDS_BYTES = (OFF_R + 2 * NSC * (int)sizeof(T) + 15) & ~15;
  static_assert(2 * SLOT * (int)sizeof(T) >= 16 * 128 * 8, "b-partial scratch must fit in the slots");
};

// Regressor g of a group (posterior_large_group): pointers into the per-regressor workspaces move by a byte stride
template <typename P>
__device__ __forceinline__ P* ws_shift(P* p, int64_t bytes) {
  return p ? reinterpret_cast<P*>(reinterpret_cast<uintptr_t>(p) + (uintptr_t)bytes) : p;
}

// ---- column statistics -------------------------------------------------------------------------------------
template <typename T>
struct ColstatsArgs {
  const T* X; int64_t ldx;
  const T* y; const T* s; const T* mw;
  T* r;               // [N]  delta_n / s_n
  T* mu;              // [N]  x_n'mw, or NULL (multi-output evidence: the residuals of the other columns of Y need it)
  T* w;               // [N]  1 / s_n for the Gram launch (diagonal noise; may be NULL): an exact division here instead of a
                      //      reciprocal approximation per wave and half-stage inside the matrix loop
  double* qpart;      // [gridDim.x]
  double* lpart;      // [gridDim.x]
  unsigned* noise_info;  // atomicMin target, 0xFFFFFFFF = every variance positive; else 1-based index of the first bad one (NULL: off)
  int layout, noise_kind, D, N;
  // blockIdx.y = regressor of a group: element strides of the caller's arrays, byte stride of r / qpart / lpart / noise_info
  int64_t grp_X, grp_y, grp_s, grp_mw, grp_ws;
  int w_sqrt;         // w receives sqrt(1 / s_n): the planes path scales BOTH operands of the Gram product (blr_planes.hpp)
  // X == NULL: the design matrix is a random-Fourier basis that is never materialised (blr_posterior_rff_f32 at D > 128):
  // phi_f(x_n) = rff_scale cos(Omega_f' x_n + phase_f) is evaluated here when the prior mean is not zero
  const T* rff_Xin; int64_t rff_ldxin; const T* rff_Omega; int64_t rff_ldo; const T* rff_phase; T rff_scale; int rff_Din;
};

template <typename T>
__global__ __launch_bounds__(kThreads) void colstats_kernel(ColstatsArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* const mwl = reinterpret_cast<T*>(smem);
  double* const scr = reinterpret_cast<double*>(smem + (((size_t)a.D * sizeof(T) + 15) & ~(size_t)15));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int D = a.D, N = a.N;
  if (const int64_t g = blockIdx.y) {
    if (a.X) a.X += g * a.grp_X;
    a.y += g * a.grp_y; a.s += g * a.grp_s; a.mw += g * a.grp_mw;
    a.r = ws_shift(a.r, g * a.grp_ws); a.mu = ws_shift(a.mu, g * a.grp_ws); a.w = ws_shift(a.w, g * a.grp_ws); a.qpart = ws_shift(a.qpart, g * a.grp_ws); a.lpart = ws_shift(a.lpart, g * a.grp_ws);
    a.noise_info = ws_shift(a.noise_info, g * a.grp_ws);
  }
  int mw_nonzero = 0;
  for (int d = tid; d < D; d += kThreads) {
    const T m = a.mw[d];
    mwl[d] = m;
    mw_nonzero |= (m != T(0));  // (NaN counts as non-zero: the general path propagates it)
  }
  mw_nonzero = __syncthreads_or(mw_nonzero);
  const bool diag = a.noise_kind == NOISE_DIAGONAL;
  const T s_iso = diag ? T(1) : a.s[0];
  double q = 0.0, l = 0.0;
  unsigned bad = 0xFFFFFFFFu;  // reference :79: _cholesky(Sigma_y) throws at the first variance that is not positive
  typedef T vecT __attribute__((ext_vector_type(Mfma<T>::VEC)));
  constexpr int VEC = Mfma<T>::VEC;
  const bool vec_ok = a.X != nullptr && a.layout == LAYOUT_COLVECS && (D % VEC) == 0 && (a.ldx % VEC) == 0 && ((uintptr_t)a.X % 16) == 0;
  if (!mw_nonzero) {
    // zero prior mean (the reference's usual prior, and SURVEY 8(d)'s): X'mw = 0 exactly, so delta = y and this pass does not
    // have to read X at all (42 us of HBM streaming at D = 1024, N = 65536)
    for (int n = blockIdx.x * kThreads + tid; n < N; n += gridDim.x * kThreads) {
      const T sv = diag ? a.s[n] : s_iso;
      if (!(sv > T(0))) bad = min(bad, (unsigned)(n + 1));
      const T delta = a.y[n];
      const T rn = delta / sv;
      a.r[n] = rn;
      if (a.mu) a.mu[n] = T(0);
      if (a.w) a.w[n] = a.w_sqrt ? (T)sqrt((double)(T(1) / sv)) : T(1) / sv;
      q += (double)delta * (double)rn;
      if (diag) l += log((double)sv);
    }
  } else if (a.X == nullptr) {
    // random-Fourier basis, prior mean != 0: X'mw needs the features; one column per wave, lanes stride over the features (each lane
    // evaluates its own: Din multiply-adds and one cosine), fixed-order butterfly
    if constexpr (sizeof(T) == 4) {
      for (int n = blockIdx.x * kWaves + wave; n < N; n += gridDim.x * kWaves) {
        const T* xin = a.rff_Xin + (int64_t)n * a.rff_ldxin;
        double mu = 0.0;
        for (int f = lane; f < D; f += 64) {
          const T* om = a.rff_Omega + (int64_t)f * a.rff_ldo;
          float acc = a.rff_phase[f];
          for (int k = 0; k < a.rff_Din; ++k) acc = __builtin_fmaf(om[k], xin[k], acc);
          mu += (double)(a.rff_scale * rff_cos(acc)) * (double)mwl[f];
        }
        mu = wave_allreduce(mu);
        const T sv = diag ? a.s[n] : s_iso;
        if (!(sv > T(0))) bad = min(bad, (unsigned)(n + 1));
        const T delta = a.y[n] - (T)mu;
        const T rn = delta / sv;
        if (lane == 0) {
          a.r[n] = rn;
          if (a.mu) a.mu[n] = (T)mu;
          if (a.w) a.w[n] = a.w_sqrt ? (T)sqrt((double)(T(1) / sv)) : T(1) / sv;
          q += (double)delta * (double)rn;
          if (diag) l += log((double)sv);
        }
      }
    }
  } else if (vec_ok) {
    // two columns per wave per step, 16-byte loads, up to 8 loads in flight per lane before the first use
    const int DV = D / VEC;
    const vecT* mwv = reinterpret_cast<const vecT*>(mwl);
    const int wid = blockIdx.x * kWaves + wave, nw = gridDim.x * kWaves;
    for (int n = 2 * wid; n < N; n += 2 * nw) {
      const bool two = n + 1 < N;
      const vecT* c0 = reinterpret_cast<const vecT*>(a.X + (int64_t)n * a.ldx);
      const vecT* c1 = reinterpret_cast<const vecT*>(a.X + (int64_t)(two ? n + 1 : n) * a.ldx);
      double mu0 = 0.0, mu1 = 0.0;
      for (int i0 = 0; i0 < DV; i0 += 256) {
        vecT v0[4], v1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * 64 + lane;
          const bool in = i < DV;
          v0[u] = in ? c0[i] : vecT(T(0));
          v1[u] = in ? c1[i] : vecT(T(0));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * 64 + lane;
          if (i < DV) {
            const vecT m = mwv[i];
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
              mu0 += (double)v0[u][e] * (double)m[e];
              mu1 += (double)v1[u][e] * (double)m[e];
            }
          }
        }
      }
      mu0 = wave_allreduce(mu0);
      mu1 = wave_allreduce(mu1);
      if (lane < 2 && (lane == 0 || two)) {
        const int nn = n + lane;
        const double mu = lane == 0 ? mu0 : mu1;
        const T sv = diag ? a.s[nn] : s_iso;
        if (!(sv > T(0))) bad = min(bad, (unsigned)(nn + 1));
        const T delta = a.y[nn] - (T)mu;
        const T rn = delta / sv;
        a.r[nn] = rn;
        if (a.mu) a.mu[nn] = (T)mu;
        if (a.w) a.w[nn] = a.w_sqrt ? (T)sqrt((double)(T(1) / sv)) : T(1) / sv;
        q += (double)delta * (double)rn;
        if (diag) l += log((double)sv);
      }
    }
  } else if (a.layout == LAYOUT_COLVECS) {
    // one column per wave: lanes stride over d (coalesced), fixed-order butterfly
    for (int n = blockIdx.x * kWaves + wave; n < N; n += gridDim.x * kWaves) {
      const T* col = a.X + (int64_t)n * a.ldx;
      double mu = 0.0;
      for (int d = lane; d < D; d += 64) mu += (double)col[d] * (double)mwl[d];
      mu = wave_allreduce(mu);
      const T sv = diag ? a.s[n] : s_iso;
      if (!(sv > T(0))) bad = min(bad, (unsigned)(n + 1));
      const T delta = a.y[n] - (T)mu;
      const T rn = delta / sv;
      if (lane == 0) {
        a.r[n] = rn;
        if (a.mu) a.mu[n] = (T)mu;
        if (a.w) a.w[n] = a.w_sqrt ? (T)sqrt((double)(T(1) / sv)) : T(1) / sv;
        q += (double)delta * (double)rn;
        if (diag) l += log((double)sv);
      }
    }
  } else {
    // RowVecs: one column per thread, coalesced along n
    for (int n = blockIdx.x * kThreads + tid; n < N; n += gridDim.x * kThreads) {
      double mu = 0.0;
      for (int d = 0; d < D; ++d) mu += (double)a.X[(int64_t)d * a.ldx + n] * (double)mwl[d];
      const T sv = diag ? a.s[n] : s_iso;
      if (!(sv > T(0))) bad = min(bad, (unsigned)(n + 1));
      const T delta = a.y[n] - (T)mu;
      const T rn = delta / sv;
      a.r[n] = rn;
      if (a.mu) a.mu[n] = (T)mu;
      if (a.w) a.w[n] = a.w_sqrt ? (T)sqrt((double)(T(1) / sv)) : T(1) / sv;
      q += (double)delta * (double)rn;
      if (diag) l += log((double)sv);
    }
  }
  q = block_allreduce(q, scr, tid);
  l = block_allreduce(l, scr, tid);
  if (tid == 0) { a.qpart[blockIdx.x] = q; a.lpart[blockIdx.x] = l; }
  if (a.noise_info && bad != 0xFFFFFFFFu) atomicMin(a.noise_info, bad);  // rare path; min is order-independent
}

// ---- 128 x 128 macro tile of a (weighted) Gram matrix ---------------------------------------------------------
// mode_out 0: write the tile (and, for diagonal tiles with r != NULL, b_I) to the split-partial workspace
// mode_out 1: subtract the tile in place from C (trailing update of the blocked Cholesky; single split)
// ---- f32, full off-diagonal macro tile, LDS-DMA: the ring loop of the D = 128 kernel (gram_iso_ring) carried over ----------
// The stage loop of gram_tile_kernel waits vmcnt(0) + s_barrier once per 8 k-steps, issues the next stage's 8 LDS-DMA pieces per
// wave in one go right behind the barrier and leaves the order of fragment reads and MFMAs to hipcc: 68 % of the f32 matrix
// peak on the tiles it computes (config 3, PMC: SQ_VALU_MFMA_BUSY 69 %).  Here the 64 KB staging area is a ring of FOUR
// half-stages (4 k-steps = 16 columns, A side + B side 16 KB); while half h computes, half h+1 is visible, h+2 is landing and
// the four pieces (+ the weights piece of wave 0) of half h+3 are issued in one burst from inline asm; arrival is a counted
// s_waitcnt vmcnt(n) at the END of a half, one barrier per half.  Inside a k-step the order is pinned with sched_barrier: the
// 9 fragment reads of the NEXT k-step and the DMA burst sit between the 16 MFMAs of this one.
template <typename T>
struct GFrag {
  T a[4], b[4];
};

// one k-step: the MFMAs of `fc` (A side scaled by w: Sigma_y^-1 of this lane's column, 1 when SCALE is off) with the reads of
// the next k-step's fragments and one slot of side work (`side`: LDS-DMA pieces, weight reads / reciprocals) between them.
// ROLE 0: all 16 tiles of the wave's 64 x 64 quadrant.  Diagonal macro tiles (only their lower triangle is ever read) deal
// their 36 useful 16 x 16 tiles as 10 + 10 + 8 + 8:  ROLE 1 (the two diagonal quadrants): the 10 tiles with k <= i;
// ROLE 2 (the quadrant below the diagonal, shared by two waves): tile rows 0, 1 of the wave's half, 8 tiles, fn.a[0..1] only;
// `side2(piece)`, pieces 0..3, is more side work dealt between its last MFMAs (the b partials of the wave that used to sit out).
template <typename T, bool SCALE, int ROLE, typename P, typename P2>
__device__ __forceinline__ void gram_kstep_ring(typename Mfma<T>::acc4 (&acc)[4][4], const GFrag<T>& fc, GFrag<T>& fn, T w,
                                                const T* __restrict__ kA_n, const T* __restrict__ kB_n, P side, P2 side2) {
  T fa[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) fa[i] = SCALE ? fc.a[i] * w : fc.a[i];  // Sigma_y^-1 on the A side only
  auto mma = [&](auto itag, auto ktag) {
    constexpr int i = decltype(itag)::value, k = decltype(ktag)::value;
    acc[i][k] = Mfma<T>::mma(fa[i], fc.b[k], acc[i][k]);
  };
#define BLR_SB __builtin_amdgcn_sched_barrier(0)
#define BLR_M(i, k) BLR_SB; mma(std::integral_constant<int, i>{}, std::integral_constant<int, k>{}); BLR_SB
  if constexpr (ROLE == 0) {
    BLR_M(0, 0);
    fn.a[0] = kA_n[0]; fn.a[1] = kA_n[64];
    BLR_M(0, 1);
    fn.a[2] = kA_n[128]; fn.a[3] = kA_n[192];
    BLR_M(0, 2);
    fn.b[0] = kB_n[0]; fn.b[1] = kB_n[64];
    BLR_M(0, 3);
    fn.b[2] = kB_n[128]; fn.b[3] = kB_n[192];
    BLR_M(1, 0);
    side();
    BLR_M(1, 1); BLR_M(1, 2); BLR_M(1, 3);
    BLR_M(2, 0); BLR_M(2, 1); BLR_M(2, 2); BLR_M(2, 3);
    BLR_M(3, 0); BLR_M(3, 1); BLR_M(3, 2); BLR_M(3, 3);
  } else if constexpr (ROLE == 1) {
    BLR_M(0, 0);
    fn.a[0] = kA_n[0]; fn.a[1] = kA_n[64];
    BLR_M(1, 0);
    fn.a[2] = kA_n[128]; fn.a[3] = kA_n[192];
    BLR_M(1, 1);
    fn.b[0] = kB_n[0]; fn.b[1] = kB_n[64];
    BLR_M(2, 0);
    fn.b[2] = kB_n[128]; fn.b[3] = kB_n[192];
    BLR_M(2, 1);
    side();
    BLR_M(2, 2); BLR_M(3, 0); BLR_M(3, 1); BLR_M(3, 2); BLR_M(3, 3);
  } else {
    BLR_M(0, 0);
    fn.a[0] = kA_n[0]; fn.a[1] = kA_n[64];
    BLR_M(0, 1);
    fn.b[0] = kB_n[0]; fn.b[1] = kB_n[64];
    BLR_M(0, 2);
    fn.b[2] = kB_n[128]; fn.b[3] = kB_n[192];
    BLR_M(0, 3);
    side();
    BLR_M(1, 0);
    side2(std::integral_constant<int, 0>{});
    BLR_M(1, 1);
    side2(std::integral_constant<int, 1>{});
    BLR_M(1, 2);
    side2(std::integral_constant<int, 2>{});
    BLR_M(1, 3);
    side2(std::integral_constant<int, 3>{});
  }
#undef BLR_SB
#undef BLR_M
}

// ---- the same product on the bf16 matrix cores: an fp32 number is EXACTLY three bf16 numbers ------------------------------------------------
// (24 mantissa bits = 3 x 8: bf16(x), bf16 of the remainder, the rest); of the nine products the six with hh, hm, mh, mm, hl, lh
// keep fp32-level accuracy under fp32 accumulation (tools/bf3_unit.hip: 2.6e-7 of sum |a b| against 2.1e-7 for an fp32 fma chain), and
// v_mfma_f32_32x32x16_bf16 runs at 2.2 PFLOP/s against 0.155 for v_mfma_f32_16x16x4_f32 (profiles/r05_microbench_bf16_probe.txt):
// 24 instructions per wave and half-stage instead of 64.  The operands come from the SAME fragment registers: after
// v_permlane16_swap_b32 of the fragments of row blocks I0 and I0 + 1, lane (r16, q) holds row block I0 + (q & 1) with column
// 4 ks + 2 (q >> 1) in the first register and 4 ks + 2 (q >> 1) + 1 in the second -- a consistent 32-row operand whose eight k slots are
// the four k-steps of a half x {first, second}.  Both sides are built the same way, so slot j pairs the same column on both.
typedef float gram_f16v __attribute__((ext_vector_type(16)));
typedef __bf16 gram_bf8 __attribute__((ext_vector_type(8)));
typedef unsigned gram_u4 __attribute__((ext_vector_type(4)));
struct Bf3Planes { gram_u4 h, m, l; };  // eight bf16 k slots per lane and plane
typedef __bf16 gram_bf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned bf3_pk(float lo, float hi) {  // v_cvt_pk_bf16_f32: round to nearest even, `lo` to bits 15:0
  const gram_bf2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}
// Rounded to NEAREST at every level (a split by masking truncates: same-sign residuals, and the three dropped products add up coherently
// over the observations -- 6e-8 of a diagonal entry per 16 columns in tools/bf3_unit.hip, 2.4e-6 of A at config 5 against 3.7e-7 for fp32
// LAPACK; rounded, the residuals are signed and zero-mean: 1.4e-7 of sum |a b| against 2.1e-7 for an fp32 fma chain)
__device__ __forceinline__ Bf3Planes bf3_split_pack(const float (&f0)[4], const float (&f1)[4]) {  // fragments of row blocks I0, I0 + 1, four k-steps
  Bf3Planes p;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(f0[ks]), __float_as_uint(f1[ks]), false, false);
    const float x = __uint_as_float(sw[0]), y = __uint_as_float(sw[1]);  // slots 2 ks, 2 ks + 1
    const unsigned h = bf3_pk(x, y);
    const float xr = x - __uint_as_float(h << 16), yr = y - __uint_as_float(h & 0xffff0000u);
    const unsigned m = bf3_pk(xr, yr);
    const float xl = xr - __uint_as_float(m << 16), yl = yr - __uint_as_float(m & 0xffff0000u);
    p.h[ks] = h;
    p.m[ks] = m;
    p.l[ks] = bf3_pk(xl, yl);
  }
  return p;
}
__device__ __forceinline__ gram_f16v bf3_mma(const Bf3Planes& a, const Bf3Planes& b, gram_f16v c) {  // smallest terms first
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gram_bf8, a.l), __builtin_bit_cast(gram_bf8, b.h), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gram_bf8, a.h), __builtin_bit_cast(gram_bf8, b.l), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gram_bf8, a.m), __builtin_bit_cast(gram_bf8, b.m), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gram_bf8, a.m), __builtin_bit_cast(gram_bf8, b.h), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gram_bf8, a.h), __builtin_bit_cast(gram_bf8, b.m), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gram_bf8, a.h), __builtin_bit_cast(gram_bf8, b.h), c, 0, 0, 0);
  return c;
}

// columns [c0, c0 + 16 nh) of the operand rows rowA.. (A side) and rowB.. (B side); baseA / baseB point at (row, column 0).
// DIAGT: diagonal macro tile -- one side only (B = A) and only the 36 tiles of 16 x 16 on or below the diagonal: waves 0 and 3
// take the 10 of their diagonal quadrant (acc[i][k], k <= i), waves 2 and 1 share the quadrant below the diagonal (its tile rows
// 0-1 / 2-3, acc[0..1][k]); wave 1 -- which used to sit out -- also accumulates b_I = X_I r (r != NULL).  10 MFMAs per k-step
// on the critical waves instead of 16: the host gives diagonal tiles longer column ranges (fewer splits) to match.
template <typename T, bool SCALE, bool DIAGT, bool PRE = false /* s holds the weights 1 / s_n themselves */, bool BF3 = false /* off-diagonal tiles on the bf16 matrix cores */>
__device__ __forceinline__ void gram_ring_loop(T* __restrict__ ring, T* __restrict__ wring, T* __restrict__ rring,
                                               const BLR_GLOBAL T* baseA, int64_t ldA, const BLR_GLOBAL T* baseB, int64_t ldB,
                                               const BLR_GLOBAL T* s /* + c0; SCALE */, const BLR_GLOBAL T* r /* + c0 or NULL; DIAGT */,
                                               int c0, int nh, unsigned voffA, unsigned voffB, int lane, int wave /*uniform*/,
                                               typename Mfma<T>::acc4 (&acc)[4][4], double (&bacc)[8]) {
  static_assert(sizeof(T) == 4, "four halves of 16 KB: f32 only");
  constexpr int SIDE = 4 * 8 * 64;                   // elements of one side of a half
  constexpr int HALF = 2 * SIDE;
  constexpr int HC = 16;                             // columns per half
  const int wr = wave >> 1, wc = wave & 1;
  unsigned ring_addr = lds_addr_of(ring), wring_addr = lds_addr_of(wring), rring_addr = lds_addr_of(rring);
  asm volatile("" : "+v"(ring_addr), "+v"(wring_addr), "+v"(rring_addr));  // pinned in VGPRs (see gram_iso_ring)
  const bool w_piece = SCALE && wave == 0;
  const bool b_wave = DIAGT && wave == 1;            // the wave that also accumulates the b partials
  const bool r_piece = b_wave && r != nullptr;
  const int npieces = 2 + (DIAGT ? 0 : 2) + (w_piece ? 1 : 0) + (r_piece ? 1 : 0);  // LDS-DMA instructions of this wave per half
  // this wave's pieces of a half: g = wave (k-step wave / 2, row blocks 4 (wave & 1) ..) and g = 4 + wave, on both sides.  Their
  // global addresses advance by 16 columns per half: 64-bit scalar adds, no multiplications in the loop
  const int jp = wave >> 1, I0 = 4 * (wave & 1);
  uint64_t nextA = (uint64_t)(uintptr_t)baseA + (uint64_t)(((int64_t)(c0 + 4 * jp) * ldA + 16 * I0) * (int64_t)sizeof(T));
  uint64_t nextB = (uint64_t)(uintptr_t)baseB + (uint64_t)(((int64_t)(c0 + 4 * jp) * ldB + 16 * I0) * (int64_t)sizeof(T));
  uint64_t nextS = (uint64_t)(uintptr_t)s, nextR = (uint64_t)(uintptr_t)r;
  const uint64_t stepA = (uint64_t)(16 * ldA * (int64_t)sizeof(T)), stepB = (uint64_t)(16 * ldB * (int64_t)sizeof(T));
  const uint64_t p1A = (uint64_t)(8 * ldA * (int64_t)sizeof(T)), p1B = (uint64_t)(8 * ldB * (int64_t)sizeof(T));  // g + 4: two k-steps on
  int hi = 0;  // half the next issue belongs to
  auto issue_first = [&]() {   // pieces g = wave of both sides (+ the variances / the residuals)
    const unsigned slot = ring_addr + (unsigned)((hi & 3) * HALF * (int)sizeof(T) + wave * 1024);
    glds_s<16>(uni((int64_t)nextA), voffA, slot);
    if constexpr (!DIAGT) glds_s<16>(uni((int64_t)nextB), voffB, slot + (unsigned)(SIDE * (int)sizeof(T)));
    if constexpr (SCALE) {
      if (w_piece)  // the 16 raw variances of the half: one dword piece
        glds_s<4, 16>(uni((int64_t)nextS), (unsigned)(lane * 4), wring_addr + (unsigned)((hi & 3) * HC * (int)sizeof(T)));
    }
    if constexpr (DIAGT) {
      if (r_piece) glds_s<4, 16>(uni((int64_t)nextR), (unsigned)(lane * 4), rring_addr + (unsigned)((hi & 3) * HC * (int)sizeof(T)));
    }
  };
  auto issue_second = [&]() {  // pieces g = 4 + wave, then on to the next half
    const unsigned slot = ring_addr + (unsigned)((hi & 3) * HALF * (int)sizeof(T) + (4 + wave) * 1024);
    glds_s<16>(uni((int64_t)(nextA + p1A)), voffA, slot);
    if constexpr (!DIAGT) glds_s<16>(uni((int64_t)(nextB + p1B)), voffB, slot + (unsigned)(SIDE * (int)sizeof(T)));
    nextA += stepA; nextB += stepB; nextS += HC * sizeof(T); nextR += HC * sizeof(T);
    ++hi;
  };
  auto retire = [&](bool keep_one) {  // wait for everything but the youngest half of this wave's pieces
    if (!keep_one) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
    switch (npieces) {  // wave-uniform
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    }
  };
  for (int h = 0; h < 3 && h < nh; ++h) { issue_first(); issue_second(); }
  retire(nh > 2);
  __syncthreads();
  // the main loop, once per wave role (the role is wave-uniform; each instance is its own pinned instruction stream)
  auto run = [&](auto role_tag) {
    constexpr int ROLE = decltype(role_tag)::value;
    GFrag<T> f0 = {}, f1 = {};
    // weights of this lane's column in the four k-steps of the current half (wcur) and of the next one (wnxt): every wave turns
    // the raw variances into reciprocals itself, half a stage ahead of their use, off the MFMA issue path
    T wcur[4] = {T(1), T(1), T(1), T(1)}, wnxt[4] = {T(1), T(1), T(1), T(1)};
    const int offB = DIAGT ? 0 : SIDE;  // diagonal tile: both operands come from the one side
    // first tile row (A side) and first tile column (B side) of this wave's fragments
    const int rowA0 = ROLE == 2 ? 4 + 2 * (wave == 1 ? 1 : 0) : 4 * wr;
    const int colB0 = ROLE == 2 ? 0 : 4 * wc;
    constexpr int NA = ROLE == 2 ? 2 : 4;
    // b partials (ROLE 2, wave 1): the eight A-side fragments of a k-step and this lane's r, read one k-step ahead of their use
    T bf[8] = {T(0), T(0), T(0), T(0), T(0), T(0), T(0), T(0)};
    T brn = T(0), brn_next = T(0);
    const bool do_b = ROLE == 2 && b_wave && r != nullptr;
    {
      const T* kA = ring + rowA0 * 64 + lane;
      const T* kB = ring + offB + colB0 * 64 + lane;
#pragma unroll
      for (int i = 0; i < 4; ++i) { f0.a[i] = i < NA ? kA[i * 64] : T(0); f0.b[i] = kB[i * 64]; }
      if constexpr (SCALE) {
#pragma unroll
        for (int j = 0; j < 4; ++j) wcur[j] = PRE ? wring[4 * j + (lane >> 4)] : fast_rcp(wring[4 * j + (lane >> 4)]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int h = 0; h < nh; ++h) {
      // here: halves h and h+1 are visible, f0 = fragments of (h, k-step 0), half h+2 is landing
      const T* slot = ring + (h & 3) * HALF;
      const T* slot_n = ring + ((h + 1) & 3) * HALF;
      const T* kA = slot + rowA0 * 64 + lane;
      const T* kB = slot + offB + colB0 * 64 + lane;
      const T* wv_n = wring + ((h + 1) & 3) * HC + (lane >> 4);
      const T* rb = rring + (h & 3) * HC + (lane >> 4);
      const bool more = h + 3 < nh;  // slot (h + 3) % 4 was freed by the barrier that ended half h-1
      auto side0 = [&] { if (more) issue_first(); };
      auto side1 = [&] { if (more) issue_second(); };
      auto side2 = [&] {
        if constexpr (SCALE) {
#pragma unroll
          for (int j = 0; j < 4; ++j) wnxt[j] = wv_n[4 * j];  // raw variances of half h+1 (visible since the last barrier)
        }
      };
      auto side3 = [&] {
        if constexpr (SCALE && !PRE) {
#pragma unroll
          for (int j = 0; j < 4; ++j) wnxt[j] = fast_rcp(wnxt[j]);
        }
      };
      // b partials, two tile rows per piece: use what the previous k-step read, then read this k-step's fragments and r
      auto bpiece = [&](int j, auto ptag) {
        constexpr int pc = decltype(ptag)::value;
        if constexpr (ROLE == 2) {
          if (do_b) {
#pragma unroll
            for (int i = 2 * pc; i < 2 * pc + 2; ++i) {
              bacc[i] += (double)bf[i] * (double)brn;
              bf[i] = slot[(j * 8 + i) * 64 + lane];
            }
            if constexpr (pc == 0) brn_next = rb[4 * j];
            if constexpr (pc == 3) brn = brn_next;
          }
        }
      };
      gram_kstep_ring<T, SCALE, ROLE>(acc, f0, f1, wcur[0], kA + 1 * 512, kB + 1 * 512, side0, [&](auto pt) { bpiece(0, pt); });
      gram_kstep_ring<T, SCALE, ROLE>(acc, f1, f0, wcur[1], kA + 2 * 512, kB + 2 * 512, side1, [&](auto pt) { bpiece(1, pt); });
      gram_kstep_ring<T, SCALE, ROLE>(acc, f0, f1, wcur[2], kA + 3 * 512, kB + 3 * 512, side2, [&](auto pt) { bpiece(2, pt); });
      gram_kstep_ring<T, SCALE, ROLE>(acc, f1, f0, wcur[3], slot_n + rowA0 * 64 + lane, slot_n + offB + colB0 * 64 + lane, side3,
                                      [&](auto pt) { bpiece(3, pt); });
#pragma unroll
      for (int j = 0; j < 4; ++j) wcur[j] = wnxt[j];
      // ---- end of half h: publish half h+2, free the slot of half h
      if (h + 1 < nh) {
        if (h + 2 < nh) retire(more);
        __syncthreads();
      }
    }
    if constexpr (ROLE == 2) {
      if (do_b) {
#pragma unroll
        for (int i = 0; i < 8; ++i) bacc[i] += (double)bf[i] * (double)brn;
      }
    }
  };
  // the whole quadrant on the bf16 matrix cores (see bf3_split_pack): per half the 32 fragments of its four k-steps are read, scaled,
  // split and packed, then 2 x 2 tiles of 32 x 32 take six products each; the finished quadrant goes back to the 16 x 16 accumulator
  // layout of the kernel's epilogues through the (then free) ring, 16 KiB per wave
  auto run_bf3 = [&]() {
    gram_f16v accb[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int v = 0; v < 16; ++v) accb[i][k][v] = 0.f;
    T wcur[4] = {T(1), T(1), T(1), T(1)}, wnxt[4] = {T(1), T(1), T(1), T(1)};
    const int rowA0 = 4 * wr, colB0 = 4 * wc;
    if constexpr (SCALE) {
#pragma unroll
      for (int j = 0; j < 4; ++j) wcur[j] = PRE ? wring[4 * j + (lane >> 4)] : fast_rcp(wring[4 * j + (lane >> 4)]);
    }
#pragma unroll 1
    for (int h = 0; h < nh; ++h) {
      const T* slot = ring + (h & 3) * HALF;
      const T* kA = slot + rowA0 * 64 + lane;
      const T* kB = slot + SIDE + colB0 * 64 + lane;
      const T* wv_n = wring + ((h + 1) & 3) * HC + (lane >> 4);
      const bool more = h + 3 < nh;  // slot (h + 3) % 4 was freed by the barrier that ended half h-1
      float fa[4][4], fb[4][4];  // [row block][k-step]
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) { fa[i][j] = kA[j * 512 + i * 64]; fb[i][j] = kB[j * 512 + i * 64]; }
      if (more) { issue_first(); issue_second(); }
      if constexpr (SCALE) {
#pragma unroll
        for (int j = 0; j < 4; ++j) wnxt[j] = PRE ? wv_n[4 * j] : fast_rcp(wv_n[4 * j]);  // half h+1: visible since the last barrier
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) fa[i][j] *= wcur[j];  // Sigma_y^-1 on the A side only
      }
      const Bf3Planes a0 = bf3_split_pack(fa[0], fa[1]), a1 = bf3_split_pack(fa[2], fa[3]);
      const Bf3Planes b0 = bf3_split_pack(fb[0], fb[1]), b1 = bf3_split_pack(fb[2], fb[3]);
      accb[0][0] = bf3_mma(a0, b0, accb[0][0]);
      accb[0][1] = bf3_mma(a0, b1, accb[0][1]);
      accb[1][0] = bf3_mma(a1, b0, accb[1][0]);
      accb[1][1] = bf3_mma(a1, b1, accb[1][1]);
#pragma unroll
      for (int j = 0; j < 4; ++j) wcur[j] = wnxt[j];
      if (h + 1 < nh) {
        if (h + 2 < nh) retire(more);
        __syncthreads();
      }
    }
    __syncthreads();  // everybody is done with the ring
    float* const Q = ring + wave * 4096;  // this wave's 64 x 64 quadrant, row-major
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int v = 0; v < 16; ++v) Q[(32 * i + 8 * (v >> 2) + 4 * (lane >> 5) + (v & 3)) * 64 + 32 * k + (lane & 31)] = accb[i][k][v];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[i][k][v] += Q[(16 * i + Mfma<T>::crow(lane, v)) * 64 + 16 * k + (lane & 15)];
  };
  // diagonal macro tile on the bf16 matrix cores: the lower triangle as 10 tiles of 32 x 32 (pi, pk), pk <= pi, dealt 3 + 2 + 3 + 2 --
  // wave 0: (0,0) (1,0) (1,1); wave 1: (3,0) (3,1) and the b partials (it issues their LDS-DMA piece); wave 2: (2,0) (2,1) (2,2);
  // wave 3: (3,2) (3,3) -- 18 matrix instructions per half on the critical waves against 24 of an off-diagonal tile: the ratio the
  // host's split plan assumes (kDiagCost).  The A role of a row pair carries Sigma_y^-1, the B role does not: up to four splits per wave.
  // The finished triangle goes through the ring (128 x 128 floats) into the accumulator dealing the epilogue expects.
  auto run_bf3_diag = [&](auto wtag) {
    constexpr int W = decltype(wtag)::value;
    constexpr int NTL = (W == 0 || W == 2) ? 3 : 2;
    constexpr int PI[3] = {W == 0 ? 0 : (W == 2 ? 2 : 3), W == 0 ? 1 : (W == 2 ? 2 : 3), W == 0 ? 1 : 2};
    constexpr int PK[3] = {W == 0 ? 0 : (W == 3 ? 2 : 0), W == 0 ? 0 : (W == 3 ? 3 : 1), W == 0 ? 1 : 2};
    constexpr unsigned AMASK = W == 0 ? 0x3u : (W == 2 ? 0x4u : 0x8u);                       // row pairs in the A role (scaled)
    constexpr unsigned BMASK = W == 0 ? 0x3u : (W == 1 ? 0x3u : (W == 2 ? 0x7u : 0xcu));     // row pairs in the B role
    gram_f16v accb[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int v = 0; v < 16; ++v) accb[i][v] = 0.f;
    T wcur[4] = {T(1), T(1), T(1), T(1)}, wnxt[4] = {T(1), T(1), T(1), T(1)};
    if constexpr (SCALE) {
#pragma unroll
      for (int j = 0; j < 4; ++j) wcur[j] = PRE ? wring[4 * j + (lane >> 4)] : fast_rcp(wring[4 * j + (lane >> 4)]);
    }
    const bool do_b = W == 1 && r != nullptr;
#pragma unroll 1
    for (int h = 0; h < nh; ++h) {
      const T* slot = ring + (h & 3) * HALF;
      const T* kA = slot + lane;  // fragment (k-step j, row block i) at kA[j * 512 + i * 64]
      const T* wv_n = wring + ((h + 1) & 3) * HC + (lane >> 4);
      const T* rb = rring + (h & 3) * HC + (lane >> 4);
      const bool more = h + 3 < nh;
      Bf3Planes PS[4], PU[4];
      auto do_pair = [&](auto ptag) {
        constexpr int p = decltype(ptag)::value;
        if constexpr (((AMASK | BMASK) >> p) & 1u) {
          float f0[4], f1[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) { f0[j] = kA[j * 512 + (2 * p) * 64]; f1[j] = kA[j * 512 + (2 * p + 1) * 64]; }
          if constexpr ((BMASK >> p) & 1u) PU[p] = bf3_split_pack(f0, f1);
          if constexpr ((AMASK >> p) & 1u) {
            if constexpr (SCALE) {
#pragma unroll
              for (int j = 0; j < 4; ++j) { f0[j] *= wcur[j]; f1[j] *= wcur[j]; }
              PS[p] = bf3_split_pack(f0, f1);
            } else if constexpr ((BMASK >> p) & 1u) {
              PS[p] = PU[p];
            } else {
              PS[p] = bf3_split_pack(f0, f1);
            }
          }
        }
      };
      do_pair(std::integral_constant<int, 0>{}); do_pair(std::integral_constant<int, 1>{});
      do_pair(std::integral_constant<int, 2>{}); do_pair(std::integral_constant<int, 3>{});
      if constexpr (W == 1) {
        if (do_b) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const double brn = (double)rb[4 * j];
#pragma unroll
            for (int i = 0; i < 8; ++i) bacc[i] += (double)kA[j * 512 + i * 64] * brn;
          }
        }
      }
      if (more) { issue_first(); issue_second(); }
      if constexpr (SCALE) {
#pragma unroll
        for (int j = 0; j < 4; ++j) wnxt[j] = PRE ? wv_n[4 * j] : fast_rcp(wv_n[4 * j]);
      }
      accb[0] = bf3_mma(PS[PI[0]], PU[PK[0]], accb[0]);
      accb[1] = bf3_mma(PS[PI[1]], PU[PK[1]], accb[1]);
      if constexpr (NTL == 3) accb[2] = bf3_mma(PS[PI[2]], PU[PK[2]], accb[2]);
#pragma unroll
      for (int j = 0; j < 4; ++j) wcur[j] = wnxt[j];
      if (h + 1 < nh) {
        if (h + 2 < nh) retire(more);
        __syncthreads();
      }
    }
    __syncthreads();  // everybody is done with the ring
    auto put_tile = [&](auto ttag) {
      constexpr int tl = decltype(ttag)::value;
      if constexpr (tl < NTL) {
#pragma unroll
        for (int v = 0; v < 16; ++v)
          ring[(32 * PI[tl] + 8 * (v >> 2) + 4 * (lane >> 5) + (v & 3)) * 128 + 32 * PK[tl] + (lane & 31)] = accb[tl][v];
      }
    };
    put_tile(std::integral_constant<int, 0>{}); put_tile(std::integral_constant<int, 1>{}); put_tile(std::integral_constant<int, 2>{});
    __syncthreads();
    // the dealing of the f32 roles (gram_tile_kernel's epilogue): waves 0, 3: k <= i of their diagonal quadrant; waves 2, 1: tile rows 4-5 / 6-7
    constexpr bool half_role = W == 1 || W == 2;
    constexpr int tr0 = half_role ? 4 + 2 * (W == 1 ? 1 : 0) : 4 * (W >> 1), tc0 = half_role ? 0 : 4 * (W & 1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (half_role ? i >= 2 : k > i) continue;
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[i][k][v] += ring[(16 * (tr0 + i) + Mfma<T>::crow(lane, v)) * 128 + 16 * (tc0 + k) + (lane & 15)];
      }
  };
  if constexpr (DIAGT && BF3) {
    switch (wave) {  // (uniform)
      case 0: run_bf3_diag(std::integral_constant<int, 0>{}); break;
      case 1: run_bf3_diag(std::integral_constant<int, 1>{}); break;
      case 2: run_bf3_diag(std::integral_constant<int, 2>{}); break;
      default: run_bf3_diag(std::integral_constant<int, 3>{}); break;
    }
  } else if constexpr (!DIAGT && BF3) {
    run_bf3();
  } else if constexpr (!DIAGT) {
    run(std::integral_constant<int, 0>{});
  } else {
    if (wave == 0 || wave == 3) run(std::integral_constant<int, 1>{});
    else run(std::integral_constant<int, 2>{});
  }
}

template <typename T>
struct GramTileArgs {
  const T* X; int64_t ldx;   // operand matrix; element (d, n) at X[d + n*ldx] (ColVecs) or X[n + d*ldx] (RowVecs)
  int layout;                // LAYOUT_COLVECS / LAYOUT_ROWVECS / 2 = upper factor as pseudo-columns (mask n <= d)
  int use_dma;               // ColVecs, 16-byte aligned: LDS-DMA staging (1: + the ring loop for full off-diagonal f32 tiles, 2: stage loop only)
  int bf3;                   // f32 ring loop: full off-diagonal macro tiles as bf16 x 3 products on the bf16 matrix cores (see bf3_split_pack)
  const T* s; int noise_kind;  // weights w_n = 1/s_n; s == NULL: w = 1
  const T* r;                // delta_n / s_n for the b partials (NULL: skip)
  const T* wpre;             // diagonal noise: 1 / s_n precomputed (colstats_kernel), used by the ring loop instead of s; may be NULL
  int D;                     // rows of the operand
  int n_begin, n_end;        // column range of the whole contraction
  int nsplit;                // split-K factor over [n_begin, n_end)
  int nsplit_diag;           // tri 1 only, 0 = nsplit: the diagonal macro tiles' own (smaller) split factor; the launch then
                             // holds (ntiles - nblocks) nsplit off-diagonal work items followed by nblocks nsplit_diag diagonal ones
  int nlong;                 // with nsplit_diag > 0: the first `nlong` strictly lower tiles have nsplit - 1 (longer) column ranges;
                             // work items in dispatch order: diagonal tiles, long ranges, short ranges (GramPlan, blr_abi.hip)
  int tile_i0, tile_j0;      // first row-block / col-block index of the tile grid
  int ntile_rows;            // row-block count of the rectangle (tri 3) / triangle (tri 4)
  int extra_row;             // tri 4: row block of the extra row of tiles
  int tri;                   // 1: lower-triangular enumeration t -> (I >= J); 0: column of tiles (I = i0 + t, J = j0);
                             // 2: row of tiles (I = i0, J = j0 + t); 3: rectangle of ntile_rows x (ntiles / ntile_rows) tiles;
                             // 4: lower triangle over ntile_rows blocks + one extra row of tiles (single launch)
  T* Gpart;                  // mode 0: [nsplit][ntiles][128*128] column-major tiles (row = A-side row)
  double* bpart;             // mode 0: [nsplit][nblocks][128]
  int ntiles, nblocks;
  T* C; int64_t ldc;         // mode 1: C[row + col*ldc] -= tile
  int mode_out;
  int xcd_swizzle;           // remap blockIdx so that one XCD owns whole N-slices (split-K launches)
  const T* XB; int64_t ldxb; int DB;  // optional SECOND operand for the B side (rows rowB.. of XB, DB rows; same layout);
                                      // NULL: B side = X (Gram / trailing updates)
  // gridDim.y regressors of a group in one launch: element strides of X and s, byte stride of r / Gpart / bpart
  int64_t grp_X, grp_s, grp_ws;
};

// BF3K: the instantiation whose full f32 ring tiles run on the bf16 matrix cores (the posterior's Gram launch); a kernel of its own so
// that the plain f32 one keeps its register allocation (sharing one kernel cost the f32 path 15 %)
template <typename T, bool BF3K = false>
__global__ __launch_bounds__(kThreads, (sizeof(T) == 4 ? BLR_GRAM_WGS : 2)) void gram_tile_kernel(GramTileArgs<T> a) {
  using L = LargeCfg<T>;
  using acc4 = typename Mfma<T>::acc4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* const slot0 = reinterpret_cast<T*>(smem);
  T* const wbuf = reinterpret_cast<T*>(smem + L::OFF_W);
  T* const rbuf = reinterpret_cast<T*>(smem + L::OFF_R);
  double* const red = reinterpret_cast<double*>(smem + L::OFF_RED);
  int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  // Workgroups are dealt round-robin over the 8 XCDs (observed, speed only): give each XCD a CONTIGUOUS range of
  // (split, tile) work items, so that the tiles of one N-slice -- which all stream the same columns of X -- share one
  // L2 instead of pulling the slice into all eight (bijective for any grid size).
  int w = blockIdx.x + blockIdx.y * gridDim.x;  // (dispatch order: x fastest)
  if (a.xcd_swizzle == 1) {
    const int nwg = gridDim.x * gridDim.y, xcd = w & 7, qq = nwg >> 3, rr = nwg & 7;
    w = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (w >> 3);
  } else if (a.xcd_swizzle == 2 && gridDim.y == 1) {
    // A planned launch (three kinds of work items, below): the dispatch order of the KINDS is the plan, so the remapping stays
    // inside a kind -- each XCD gets a contiguous run of the kind's range-major list (the k-th workgroup of XCD x within the
    // kind takes the k-th item of x's run).  Bijective for any sizes: XCD y holds ceil((n - f_y) / 8) of a kind's n workgroups,
    // f_y = its first position there.
    const int nd = a.nblocks * a.nsplit_diag, nl = a.nlong * (a.nsplit - 1);
    const int base = w < nd ? 0 : (w < nd + nl ? nd : nd + nl);
    const int n = w < nd ? nd : (w < nd + nl ? nl : (int)gridDim.x - nd - nl);
    const int xcd = w & 7, k = (w - base - ((xcd - base) & 7)) >> 3;
    int start = 0;
    for (int y = 0; y < xcd; ++y) {
      const int fy = (y - base) & 7;
      start += fy < n ? (n - fy + 7) >> 3 : 0;
    }
    w = base + start + k;
  }
  if (gridDim.y > 1) {  // regressor of the group, then the work item within it
    const int64_t g = w / (int)gridDim.x;
    w -= (int)g * (int)gridDim.x;
    a.X += g * a.grp_X;
    if (a.s) a.s += g * a.grp_s;
    a.r = ws_shift(a.r, g * a.grp_ws); a.wpre = ws_shift(a.wpre, g * a.grp_ws); a.Gpart = ws_shift(a.Gpart, g * a.grp_ws);
    a.bpart = ws_shift(a.bpart, g * a.grp_ws);
    a.C = ws_shift(a.C, g * a.grp_ws);  // (mode 1: the tall matrix of a grouped sweep lives in the regressor's workspace)
  }
  int t = w % a.ntiles, sidx = w / a.ntiles;
  int nsplit_here = a.nsplit;
  int I, J;
  if (a.tri == 1 && a.nsplit_diag > 0) {
    // three kinds of work items, longest first: the diagonal tiles with a.nsplit_diag column ranges each, then the strictly
    // lower tiles o = I (I - 1) / 2 + J: those with o < a.nlong have a.nsplit - 1 ranges, the others a.nsplit
    const int n_off = a.ntiles - a.nblocks;
    const int nd = a.nblocks * a.nsplit_diag;
    if (w < nd) {
      I = J = w % a.nblocks;
      sidx = w / a.nblocks;
      nsplit_here = a.nsplit_diag;
    } else {
      const int nl = a.nlong * (a.nsplit - 1);
      int o;
      if (w < nd + nl) {
        const int wl = w - nd;
        o = wl % a.nlong;
        sidx = wl / a.nlong;
        nsplit_here = a.nsplit - 1;
      } else {
        const int ws = w - nd - nl, ns = n_off - a.nlong;
        o = a.nlong + ws % ns;
        sidx = ws / ns;
      }
      int ii = 1;
      while ((ii + 1) * ii / 2 <= o) ++ii;
      I = ii;
      J = o - ii * (ii - 1) / 2;
    }
    t = I * (I + 1) / 2 + J;
    I += a.tile_i0;
    J += a.tile_j0;
  } else if (a.tri == 1) {
    int ii = 0;
    while ((ii + 1) * (ii + 2) / 2 <= t) ++ii;
    I = a.tile_i0 + ii;
    J = a.tile_j0 + (t - ii * (ii + 1) / 2);
  } else if (a.tri == 0) {
    I = a.tile_i0 + t;
    J = a.tile_j0;
  } else if (a.tri == 2) {  // a row of tiles: fixed row block, column blocks j0 + t
    I = a.tile_i0;
    J = a.tile_j0 + t;
  } else if (a.tri == 3) {  // rectangle: ntile_rows row blocks x (ntiles / ntile_rows) column blocks
    I = a.tile_i0 + t % a.ntile_rows;
    J = a.tile_j0 + t / a.ntile_rows;
  } else {  // lower triangle over ntile_rows blocks, then one extra row of tiles (row block `extra_row`)
    const int ntri = a.ntile_rows * (a.ntile_rows + 1) / 2;
    if (t < ntri) {
      int ii = 0;
      while ((ii + 1) * (ii + 2) / 2 <= t) ++ii;
      I = a.tile_i0 + ii;
      J = a.tile_j0 + (t - ii * (ii + 1) / 2);
    } else {
      I = a.extra_row;
      J = a.tile_j0 + (t - ntri);
    }
  }
  const bool diag_tile = (I == J) && a.XB == nullptr;
  const int rowA = I * kPB, rowB = J * kPB;
  const int span = a.n_end - a.n_begin;
  const int per = ((span + nsplit_here - 1) / nsplit_here + L::NSC - 1) / L::NSC * L::NSC;  // whole stages per split
  const int c0 = a.n_begin + sidx * per;
  const int c1 = min(a.n_end, c0 + per);
  const int nstages = c1 > c0 ? (c1 - c0 + L::NSC - 1) / L::NSC : 0;
  const bool diag_noise = a.noise_kind == NOISE_DIAGONAL;
  const T s_iso = (a.s && !diag_noise) ? a.s[0] : T(1);
  const bool want_b = diag_tile && a.r != nullptr && a.mode_out == 0;
  // the 64 x 64 quadrant above the diagonal of a diagonal macro tile is never read (the reduction and the in-place update
  // skip col > row): its wave stays in the barriers and the staging but leaves the matrix pipe to the co-resident workgroup
  const bool upper_quadrant = diag_tile && wr == 0 && wc == 1;

  acc4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[i][j][v] = T(0);
  double bacc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) bacc[i] = 0.0;

  // stage loader: one operand side (128 rows starting at row0) into dst
  // full tiles (128 rows, the whole stage inside the column range): scalar-addressed LDS-DMA pieces -- one SGPR base per
  // piece plus a per-lane offset that never changes -- instead of a bounds-checked 64-bit address per lane and piece (the
  // piece-issue cost was the measured 18 % of this kernel)
  const unsigned voffA = glds_lane_offset<T>(a.ldx, lane);
  const unsigned voffB = glds_lane_offset<T>(a.XB ? a.ldxb : a.ldx, lane);
  auto load_side = [&](T* dst, const T* base, int64_t ldb, int Drows, int row0, int n0, unsigned voff) {
    const int rows = min(kPB, Drows - row0);  // may be <= 0 for padding blocks: everything zero-filled
    if (a.use_dma) {
      if (rows == kPB && n0 + L::NSC <= c1 && (3 * ldb + 64) * (int64_t)sizeof(T) < ((int64_t)1 << 31))
        stage_glds_full<T, 8, L::KS>(dst, as_global(base + row0), ldb, n0, wave, voff);
      else
        stage_glds<T, 8, L::KS>(dst, as_global(base + row0), ldb, rows, c1, n0, wave, lane);
    } else {
      int tt = tid;
      asm volatile("" : "+v"(tt));
      for (int idx = tt; idx < L::SIDE; idx += kThreads) {
        int d, nl;
        if (a.layout == LAYOUT_COLVECS) { d = idx % kPB; nl = idx / kPB; }
        else                            { nl = idx % L::NSC; d = idx / L::NSC; }
        const int n = n0 + nl;
        const int dg = row0 + d;
        bool ok = d < rows && n < c1 && (a.layout != 2 || n <= dg);
        int64_t addr = (a.layout == LAYOUT_COLVECS) ? (int64_t)n * ldb + dg : (int64_t)dg * ldb + n;
        T v = base[ok ? addr : 0];
        dst[frag_off(8, d, nl)] = ok ? v : T(0);
      }
    }
  };
  const T* const baseB = a.XB ? a.XB : a.X;
  const int64_t ldB = a.XB ? a.ldxb : a.ldx;
  const int rowsB = a.XB ? a.DB : a.D;
  // per-column scalars (1/s_n, r_n) of a stage travel through registers one stage ahead: loading them inside issue() and
  // storing them to LDS right away made wave 0 sit out a full memory latency at the top of EVERY stage, with the other
  // three waves waiting for it at the next barrier (measured: 791 -> see DESIGN.md us for the c3 Gram launch)
  T s_reg = T(1), r_reg = T(0);
  bool in_reg = false;
  auto load_scalars = [&](int st) {
    in_reg = false;
    if (tid < L::NSC) {
      const int n = c0 + st * L::NSC + tid;
      if (st < nstages && n < c1) {
        in_reg = true;
        s_reg = a.s ? (diag_noise ? a.s[n] : s_iso) : T(1);
        r_reg = want_b ? a.r[n] : T(0);
      }
    }
  };
  auto issue = [&](int st) {
    const int n0 = c0 + st * L::NSC;
    T* slot = slot0 + (st & 1) * L::SLOT;
    // the scalars first: consuming them makes hipcc wait vmcnt(0) (it cannot see the asm-issued pieces), which is free HERE
    // -- nothing is in flight right after the stage barrier -- and a full memory latency if it comes after this stage's pieces
    if (tid < L::NSC) {
      wbuf[(st & 1) * L::NSC + tid] = in_reg ? T(1) / s_reg : T(0);
      rbuf[(st & 1) * L::NSC + tid] = in_reg ? r_reg : T(0);
    }
    asm volatile("" ::: "memory");
    load_side(slot, a.X, a.ldx, a.D, rowA, n0, voffA);
    if (!diag_tile) load_side(slot + L::SIDE, baseB, ldB, rowsB, rowB, n0, voffB);
    load_scalars(st + 1);  // in flight until the next issue()
  };

  bool ring_done = false;
  if constexpr (sizeof(T) == 4) {
    // full macro tiles through the ring loop (gram_ring_loop); everything else -- partial tiles, ragged column ranges,
    // register staging, f64 -- stays on the stage loop below
    const int ncol = c1 - c0;
    const bool ring_ok = a.use_dma == 1 && ncol >= 16 && (ncol & 15) == 0 && min(kPB, a.D - rowA) == kPB &&
                         (diag_tile || min(kPB, rowsB - rowB) == kPB) && (3 * a.ldx + 64) * (int64_t)sizeof(T) < ((int64_t)1 << 31) &&
                         (3 * ldB + 64) * (int64_t)sizeof(T) < ((int64_t)1 << 31);
    if (ring_ok) {  // block-uniform
      const BLR_GLOBAL T* bA = as_global(a.X + rowA);
      const BLR_GLOBAL T* bB = as_global(baseB + rowB);
      const bool scale = a.s != nullptr && diag_noise;
      const bool pre = scale && a.wpre != nullptr;
      const BLR_GLOBAL T* sp = scale ? as_global((pre ? a.wpre : a.s) + c0) : (const BLR_GLOBAL T*)nullptr;
      const BLR_GLOBAL T* rp = want_b ? as_global(a.r + c0) : (const BLR_GLOBAL T*)nullptr;
      const int nh = ncol >> 4;
#define BLR_RING(SC, DG, PR) gram_ring_loop<T, SC, DG, PR>(slot0, wbuf, rbuf, bA, a.ldx, bB, ldB, sp, rp, c0, nh, voffA, voffB, lane, wave, acc, bacc)
      if constexpr (BF3K) {
      if (diag_tile) {
#define BLR_RING3D(SC, PR) gram_ring_loop<T, SC, true, PR, true>(slot0, wbuf, rbuf, bA, a.ldx, bB, ldB, sp, rp, c0, nh, voffA, voffB, lane, wave, acc, bacc)
        if (pre) BLR_RING3D(true, true);
        else if (scale) BLR_RING3D(true, false);
        else BLR_RING3D(false, false);
#undef BLR_RING3D
      } else {
#define BLR_RING3(SC, PR) gram_ring_loop<T, SC, false, PR, true>(slot0, wbuf, rbuf, bA, a.ldx, bB, ldB, sp, rp, c0, nh, voffA, voffB, lane, wave, acc, bacc)
        if (pre) BLR_RING3(true, true);
        else if (scale) BLR_RING3(true, false);
        else BLR_RING3(false, false);
#undef BLR_RING3
      }
      } else if (diag_tile) {
        if (pre) BLR_RING(true, true, true);
        else if (scale) BLR_RING(true, true, false);
        else BLR_RING(false, true, false);
      } else {
        if (pre) BLR_RING(true, false, true);
        else if (scale) BLR_RING(true, false, false);
        else BLR_RING(false, false, false);
      }
#undef BLR_RING
      if (a.s != nullptr && !diag_noise) {  // isotropic noise: applied once to the finished tile
        const T wi = T(1) / s_iso;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[i][k][v] *= wi;
      }
      ring_done = true;
    }
  }
  load_scalars(0);
  if (nstages > 0 && !ring_done) issue(0);
  for (int st = 0; st < (ring_done ? 0 : nstages); ++st) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (st + 1 < nstages) issue(st + 1);
    const T* sA = slot0 + (st & 1) * L::SLOT;
    const T* sB = diag_tile ? sA : sA + L::SIDE;
    const T* wb = wbuf + (st & 1) * L::NSC;
    const T* rb = rbuf + (st & 1) * L::NSC;
    // software pipeline: fragments of k-step j+1 are requested before the 16 MFMAs of k-step j issue
    T na[4], nb[4], nw;
    auto load_frags = [&](int j) {
      nw = wb[4 * j + (lane >> 4)];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        na[i] = sA[(j * 8 + 4 * wr + i) * 64 + lane];
        nb[i] = sB[(j * 8 + 4 * wc + i) * 64 + lane];
      }
    };
    load_frags(0);
#pragma unroll
    for (int j = 0; j < L::KS; ++j) {
      T fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { fa[i] = na[i] * nw; fb[i] = nb[i]; }
      if (j + 1 < L::KS) load_frags(j + 1);
      if (!upper_quadrant) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[i][k] = Mfma<T>::mma(fa[i], fb[k], acc[i][k]);
      }
      if (want_b && (j & 3) == wave) {
        const T rn = rb[4 * j + (lane >> 4)];
#pragma unroll
        for (int i = 0; i < 8; ++i) bacc[i] += (double)sA[(j * 8 + i) * 64 + lane] * (double)rn;
      }
    }
    // the barrier at the top of the next iteration protects the slot that issue(st + 2) overwrites
  }

  // ---- epilogue ------------------------------------------------------------------------------------------
  if (a.mode_out == 0) {
    T* out = a.Gpart + ((int64_t)sidx * a.ntiles + t) * (kPB * kPB);
    // which 16 x 16 tile acc[i][k] is: the wave's quadrant, or -- diagonal macro tile through the ring loop -- the dealing of
    // gram_ring_loop (waves 0, 3: k <= i of their diagonal quadrant; waves 2, 1: tile rows 4-5 / 6-7, columns 0-3)
    const bool dealt = ring_done && diag_tile;
    const bool half_role = dealt && (wave == 1 || wave == 2);
    const int tr0 = half_role ? 4 + 2 * (wave == 1 ? 1 : 0) : 4 * wr, tc0 = half_role ? 0 : 4 * wc;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (dealt && (half_role ? i >= 2 : k > i)) continue;  // wave-uniform
        const int col = 16 * (tc0 + k) + (lane & 15);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int row = 16 * (tr0 + i) + Mfma<T>::crow(lane, v);
          out[col * kPB + row] = acc[i][k][v];  // column-major tile: the reduce pass is coalesced both ways
        }
      }
    if (want_b) {
      __syncthreads();
      const int q = lane >> 4, r = lane & 15;
#pragma unroll
      for (int i = 0; i < 8; ++i) red[(wave * 4 + q) * kPB + 16 * i + r] = bacc[i];
      __syncthreads();
      if (tid < kPB) {
        double sum = 0.0;
#pragma unroll
        for (int p = 0; p < 16; ++p) sum += red[p * kPB + tid];
        a.bpart[((int64_t)sidx * a.nblocks + I) * kPB + tid] = sum;
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int col = rowB + 16 * (4 * wc + k) + (lane & 15);
        if constexpr (sizeof(T) == 4) {
          // f32 C layout: a lane holds 4 CONSECUTIVE rows of one column -> one 16-byte read-modify-write
          const int row0 = rowA + 16 * (4 * wr + i) + 4 * (lane >> 4);
          if (!diag_tile || col <= row0 + 3) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4* pc = reinterpret_cast<f4*>(a.C + (int64_t)col * a.ldc + row0);
            f4 c = *pc;
#pragma unroll
            for (int v = 0; v < 4; ++v) c[v] -= acc[i][k][v];  // rows above the diagonal of a diagonal tile: never read
            *pc = c;
          }
        } else {
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int row = rowA + 16 * (4 * wr + i) + Mfma<T>::crow(lane, v);
            if (!diag_tile || col <= row) a.C[(int64_t)col * a.ldc + row] -= acc[i][k][v];
          }
        }
      }
  }
}

// ---- split reduction + prior -> Abar ---------------------------------------------------------------------------
template <typename T>
struct ReduceArgs {
  const T* Gpart; const double* bpart;
  int nsplit_total;          // data splits (+1 if a prior-factor pseudo split is present)
  int nsplit_diag;           // 0, or the number of data splits of the DIAGONAL tiles (< the others'; see GramTileArgs)
  int pseudo_split;          // 1: the last of the nsplit_total partials is the prior factor's (present for every tile)
  int nlong;                 // strictly lower tiles o < nlong hold one data partial less (see GramTileArgs)
  int ntiles, nblocks;       // lower-triangular macro tiles, row blocks
  int nsplit_b;              // 0, or the number of b partials when they do not come from the Gram launch (planes_kernel's column chunks)
  int multi_block;           // 0, or the index (= D's row blocks) of the ONE extra row block of a multi-output call: its macro tiles
                             // (multi_block, J) hold b_s' for the output columns s -- rows DP + s of Abar, no prior; row DP itself (column
                             // 0) keeps the fp64 partial sums of the planes pass; nblocks counts the extra block
  const T* Lw; int64_t ldl; int prior_kind;
  int D, DP;
  T* Abar; int64_t lda;      // (DP + 128) x DP
  T* Lw_post; int64_t ldlp;  // optional full symmetric copy of A (D x D)
  int64_t grp_Lw, grp_Lp, grp_ws;  // blockIdx.z = regressor of a group: element strides of Lw / Lw_post, byte stride of Gpart / bpart / Abar
};

template <typename T>
__global__ __launch_bounds__(kThreads) void gram_reduce_kernel(ReduceArgs<T> a) {
  const int t = blockIdx.x;  // macro tile, or ntiles + I for the rhs row of block I
  const int tid = threadIdx.x;
  constexpr int kChunk = kPB * kPB / 16;  // gridDim.y = 16 chunks per tile
  const int e_begin = blockIdx.y * kChunk, e_end = e_begin + kChunk;
  if (const int64_t g = blockIdx.z) {
    a.Gpart = ws_shift(a.Gpart, g * a.grp_ws); a.bpart = ws_shift(a.bpart, g * a.grp_ws); a.Abar = ws_shift(a.Abar, g * a.grp_ws);
    a.Lw += g * a.grp_Lw;
    if (a.Lw_post) a.Lw_post += g * a.grp_Lp;
  }
  if (t < a.ntiles) {
    int ii = 0;
    while ((ii + 1) * (ii + 2) / 2 <= t) ++ii;
    const int I = ii, J = t - ii * (ii + 1) / 2;
    const bool mrows = a.multi_block > 0 && I == a.multi_block;
    if (mrows && J == I) return;  // (residuals x residuals: nobody's)
    // one 16-byte vector of 4 (f32) / 2 (f64) consecutive rows per thread and pass; the partials of 8 splits are requested
    // before the first is used (one load per split and a dependent add behind it made this kernel a ~50 us latency chain)
    constexpr int VEC = Mfma<T>::VEC;
    typedef T vecT __attribute__((ext_vector_type(Mfma<T>::VEC)));
    const int64_t sstride = (int64_t)a.ntiles * (kPB * kPB);
    for (int e = e_begin + tid * VEC; e < e_end; e += kThreads * VEC) {
      const int rl = e % kPB, cl = e / kPB;  // column-major tiles: consecutive threads -> consecutive rows
      const int row0 = I * kPB + rl, col = J * kPB + cl;
      if (col > row0 + VEC - 1) continue;
      const T* src = a.Gpart + (int64_t)t * (kPB * kPB) + e;
      vecT sum = vecT(T(0));
      // a diagonal tile with its own split factor: its data partials, then (if present) the pseudo split at the END of the stack
      const bool long_tile = a.nlong > 0 && I != J && I * (I - 1) / 2 + J < a.nlong;
      const bool short_stack = (a.nsplit_diag > 0 && I == J) || long_tile;
      const int ndata = !short_stack ? a.nsplit_total : (long_tile ? a.nsplit_total - a.pseudo_split - 1 : a.nsplit_diag);
      int sp = 0;
      for (; sp + 8 <= ndata; sp += 8) {
        vecT v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const vecT*>(src + (sp + u) * sstride);
#pragma unroll
        for (int u = 0; u < 8; ++u) sum += v[u];  // fixed order
      }
      if (sp < ndata) {  // the remaining (< 8) partials: requested together as well
        vecT v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (sp + u < ndata) ? *reinterpret_cast<const vecT*>(src + (sp + u) * sstride) : vecT(T(0));
#pragma unroll
        for (int u = 0; u < 8; ++u) sum += v[u];  // fixed order (zeros beyond ndata change nothing)
      }
      if (short_stack && a.pseudo_split) sum += *reinterpret_cast<const vecT*>(src + (a.nsplit_total - 1) * sstride);
#pragma unroll
      for (int q = 0; q < VEC; ++q) {
        const int row = row0 + q;
        if (col > row) continue;
        T val = sum[q];
        if (mrows) {  // b_s[col], s = rl + q >= 1
          if (rl + q >= 1) a.Abar[(int64_t)col * a.lda + row] = col < a.D ? val : T(0);
          continue;
        }
        if (row < a.D) {
          if (a.prior_kind == PRIOR_DENSE) val += a.Lw[(int64_t)row * a.ldl + col];  // upper entry (col, row)
          else if (a.prior_kind == PRIOR_DIAGONAL && row == col) val += a.Lw[row];
        } else {
          val = (row == col) ? T(1) : T(0);  // padding: unit diagonal
        }
        a.Abar[(int64_t)col * a.lda + row] = val;
        if (a.Lw_post && row < a.D) {
          a.Lw_post[(int64_t)col * a.ldlp + row] = val;
          a.Lw_post[(int64_t)row * a.ldlp + col] = val;
        }
      }
    }
  } else {
    const int I = t - a.ntiles;
    for (int e = e_begin + tid; e < e_end; e += kThreads) {  // rhs row block: row 0 = b', the rest zero
      const int rl = e % kPB, cl = e / kPB;
      const int col = I * kPB + cl;
      T v = T(0);
      if (rl == 0 && col < a.D) {
        double sum = 0.0;
        const int nb = a.nsplit_b > 0 ? a.nsplit_b : a.nsplit_total;
        const double* src = a.bpart + (int64_t)I * kPB + cl;
        const int64_t sst = (int64_t)a.nblocks * kPB;
        int sp = 0;
        for (; sp + 16 <= nb; sp += 16) {  // sixteen partials requested before the first is used (a load per partial and a dependent add: 0.5 us each)
          double v[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) v[u] = src[(sp + u) * sst];
#pragma unroll
          for (int u = 0; u < 16; ++u) sum += v[u];  // fixed order
        }
        for (; sp < nb; ++sp) sum += src[sp * sst];
        v = (T)sum;
      }
      if (a.multi_block > 0 && rl != 0) continue;  // (those rows are the other output columns': written above)
      a.Abar[(int64_t)col * a.lda + a.DP + rl] = v;
    }
  }
}


// ---- block copies with many loads in flight -----------------------------------------------------------------------
// A plain `for (idx = tid; ...) P[..] = g[..]` loop issues ONE global load per iteration and waits for it (~1-2 us
// each): 64 iterations serialise to ~50 us for a 128 x 128 block.  These helpers issue 16 loads before the first use.
// All blocks handled here are 16-byte aligned with a leading dimension that is a multiple of 128 (workspace matrices),
// so a 128 x 128 block is fetched as 16-byte vectors with EVERY load of the thread issued before the first use.
template <typename T, int ROWS>
struct BlockVec {
  static constexpr int VEC = Mfma<T>::VEC;
  static constexpr int VPC = ROWS / VEC;                     // vectors per column
  static constexpr int NV = kPB * ROWS / (VEC * kThreads);   // vectors per thread
  typedef T vecT __attribute__((ext_vector_type(Mfma<T>::VEC)));
  vecT v[NV];
  // ROWS x 128 column-major block at blk (element (r, c) at blk[r + c*ld])
  __device__ __forceinline__ void load(const T* __restrict__ blk, int64_t ld, int tid) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int vi = u * kThreads + tid;
      v[u] = *reinterpret_cast<const vecT*>(blk + (int64_t)(vi / VPC) * ld + (vi % VPC) * VEC);
    }
  }
  // lower triangle of a 128 x 128 block -> packed P[pidx(r, c)]
  __device__ __forceinline__ void to_packed_lower(T* __restrict__ P, int tid) const {
    static_assert(ROWS == kPB, "square block");
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int vi = u * kThreads + tid;
      const int c = vi / VPC, r0 = (vi % VPC) * VEC;
#pragma unroll
      for (int e = 0; e < VEC; ++e)
        if (r0 + e >= c) P[pidx(r0 + e, c)] = v[u][e];
    }
  }
  // upper-stored factor block (U = L'): element (r, c) of the block is L[c][r] -> P[pidx(c, r)]
  __device__ __forceinline__ void to_packed_from_upper(T* __restrict__ P, int tid) const {
    static_assert(ROWS == kPB, "square block");
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int vi = u * kThreads + tid;
      const int c = vi / VPC, r0 = (vi % VPC) * VEC;
#pragma unroll
      for (int e = 0; e < VEC; ++e)
        if (c >= r0 + e) P[pidx(c, r0 + e)] = v[u][e];
    }
  }
  // rows image Xs[r * ldxs + c]; rows >= nr are zero-filled
  __device__ __forceinline__ void to_rows(T* __restrict__ Xs, int ldxs, int nr, int tid) const {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int vi = u * kThreads + tid;
      const int c = vi / VPC, r0 = (vi % VPC) * VEC;
#pragma unroll
      for (int e = 0; e < VEC; ++e) Xs[(r0 + e) * ldxs + c] = (r0 + e < nr) ? v[u][e] : T(0);
    }
  }
};

template <typename T>
__device__ __forceinline__ void load_lower_block_to_packed(T* __restrict__ P, const T* __restrict__ blk, int64_t ld, int tid) {
  BlockVec<T, kPB> b;
  b.load(blk, ld, tid);
  b.to_packed_lower(P, tid);
}
template <typename T>
__device__ __forceinline__ void store_packed_to_lower_block(const T* __restrict__ P, T* __restrict__ blk, int64_t ld, int tid) {
  using BV = BlockVec<T, kPB>;
#pragma unroll 4
  for (int u = 0; u < BV::NV; ++u) {
    const int vi = u * kThreads + tid;
    const int c = vi / BV::VPC, r0 = (vi % BV::VPC) * BV::VEC;
    if (r0 + BV::VEC - 1 < c) continue;  // entirely above the diagonal
    typename BV::vecT o;
#pragma unroll
    for (int e = 0; e < BV::VEC; ++e) o[e] = (r0 + e >= c) ? P[pidx(r0 + e, c)] : blk[(int64_t)c * ld + r0 + e];
    *reinterpret_cast<typename BV::vecT*>(blk + (int64_t)c * ld + r0) = o;
  }
}
// the same packed image from an UPPER factor block (U = L'): P[pidx(r, c)] = U[c, r]
template <typename T>
__device__ __forceinline__ void load_upper_block_to_packed(T* __restrict__ P, const T* __restrict__ blk, int64_t ld, int tid) {
  BlockVec<T, kPB> b;
  b.load(blk, ld, tid);
  b.to_packed_from_upper(P, tid);
}

// ---- panel factorisation (diagonal block AND the rows below it in one launch): panel_chain_kernel, blr_panel.hpp ------------

// ---- trailing update of the blocked Cholesky: C -= L_I L_J' for every 64 x 64 sub-tile below panel p -----------------------
// The update has K = 128 only, so it is pure latency: one workgroup per 64 x 64 sub-tile issues ALL its loads at once
// (both 64 x 128 operand blocks as 16-byte vectors, the C sub-tile straight into the MFMA accumulators), writes the
// operands to LDS in fragment order, runs 32 k-steps x 4 MFMAs per wave and stores.  One barrier, no stage loop
// (the general split-K tile kernel took ~20 us per panel for the same work).
template <typename T>
struct TrailCfg {
  static constexpr int SB = 64;                                    // sub-tile edge
  static constexpr int SIDE = (kPB / 4) * (SB / 16) * 64;          // elements of one operand image [32][4][64]
  static constexpr int LDS_BYTES = 2 * SIDE * (int)sizeof(T);      // 64 KB (f32) / 128 KB (f64)
};

template <typename T>
__global__ __launch_bounds__(kThreads) void trail_update_kernel(T* M, int64_t ld, int p, int ntri /* 64-row blocks in the triangle */,
                                                                int row_tri0 /* first row of the triangle */,
                                                                int row_extra0 /* first extra row (rhs rows) */,
                                                                const int32_t* info, int ntiles /* all sub-tiles of the launch */,
                                                                int64_t batch_stride = 0, int info_stride = 0) {
  using Cfg = TrailCfg<T>;
  M += (int64_t)blockIdx.y * batch_stride;  // blockIdx.y: independent factorisations stepping together (panel_chain_kernel)
  info += (int64_t)blockIdx.y * info_stride;
  using acc4 = typename Mfma<T>::acc4;
  constexpr int VEC = Mfma<T>::VEC;
  constexpr int SB = Cfg::SB;
  typedef T vecT __attribute__((ext_vector_type(Mfma<T>::VEC)));
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* const As = reinterpret_cast<T*>(smem);
  T* const Bs = As + Cfg::SIDE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  // sub-tile (si, sj): lower triangle over ntri sub-blocks, then the extra row blocks x ntri columns
  const int ntt = ntri * (ntri + 1) / 2;
  // (a launch of more than two sub-tiles per CU -- c5's first panels: 525 -- runs as 512 workgroups, some taking a second
  // sub-tile, instead of a second round of a dozen workgroups that doubles the latency of the step)
  for (int t0 = blockIdx.x; t0 < ntiles; t0 += gridDim.x) {
  int t = t0, rowA, rowB;
  bool diag = false;
  if (t < ntt) {
    int si = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while ((si + 1) * (si + 2) / 2 <= t) ++si;
    while (si * (si + 1) / 2 > t) --si;
    const int sj = t - si * (si + 1) / 2;
    rowA = row_tri0 + si * SB;
    rowB = row_tri0 + sj * SB;
    diag = si == sj;
  } else {
    t -= ntt;
    rowA = row_extra0 + (t / ntri) * SB;
    rowB = row_tri0 + (t % ntri) * SB;
  }
  const T* panel = M + (int64_t)p * kPB * ld;

  // ---- every load of the workgroup in flight
  constexpr int VPC = SB / VEC;                       // vectors per operand column
  constexpr int NV = kPB * SB / (VEC * kThreads);     // vectors per thread per operand
  vecT va[NV], vb[NV];
#pragma unroll
  for (int u = 0; u < NV; ++u) {
    const int vi = u * kThreads + tid;
    const int64_t off = (int64_t)(vi / VPC) * ld + (vi % VPC) * VEC;
    va[u] = *reinterpret_cast<const vecT*>(panel + rowA + off);
    vb[u] = *reinterpret_cast<const vecT*>(panel + rowB + off);
  }
  acc4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int col = rowB + 16 * (2 * wc + k) + (lane & 15);
      if constexpr (sizeof(T) == 4) {
        const int row0 = rowA + 16 * (2 * wr + i) + 4 * (lane >> 4);
        acc[i][k] = *reinterpret_cast<const acc4*>(M + (int64_t)col * ld + row0);
      } else {
#pragma unroll
        for (int v = 0; v < 4; ++v)
          acc[i][k][v] = M[(int64_t)col * ld + rowA + 16 * (2 * wr + i) + Mfma<T>::crow(lane, v)];
      }
    }
  if (*info != 0) return;
  // ---- fragment-order images: [k-step][row block][lane], lane l <-> (row 16I + (l & 15), column 4kk + (l >> 4))
#pragma unroll
  for (int u = 0; u < NV; ++u) {
    const int vi = u * kThreads + tid;
    const int k = vi / VPC, r0 = (vi % VPC) * VEC;
    const int idx = (((k >> 2) * (SB / 16) + (r0 >> 4)) << 6) + ((k & 3) << 4) + (r0 & 15);
    *reinterpret_cast<vecT*>(As + idx) = -va[u];
    *reinterpret_cast<vecT*>(Bs + idx) = vb[u];
  }
  __syncthreads();
#pragma unroll 8
  for (int kk = 0; kk < kPB / 4; ++kk) {
    T fa[2], fb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      fa[i] = As[((kk * (SB / 16) + 2 * wr + i) << 6) + lane];
      fb[i] = Bs[((kk * (SB / 16) + 2 * wc + i) << 6) + lane];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int k = 0; k < 2; ++k) acc[i][k] = Mfma<T>::mma(fa[i], fb[k], acc[i][k]);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int col = rowB + 16 * (2 * wc + k) + (lane & 15);
      if constexpr (sizeof(T) == 4) {
        const int row0 = rowA + 16 * (2 * wr + i) + 4 * (lane >> 4);
        if (!diag || col <= row0 + 3) *reinterpret_cast<acc4*>(M + (int64_t)col * ld + row0) = acc[i][k];
      } else {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int row = rowA + 16 * (2 * wr + i) + Mfma<T>::crow(lane, v);
          if (!diag || col <= row) M[(int64_t)col * ld + row] = acc[i][k][v];
        }
      }
    }
  __syncthreads();  // the operand images are rewritten by the next sub-tile
  }
}

// ---- X <- X L_pp^-T for one block of RB rows below the diagonal block ------------------------------------------------
template <typename T>
struct TrsmCfg {
  static constexpr int RB = 64;                            // rows per workgroup (f32 at 128 rows needs 108 KB of LDS: cannot share a CU with a Gram workgroup)
  static constexpr int LDX = kPB + 1;                      // padded row stride of the X image (conflict-free)
  static constexpr int OFF_X = ((kPB * (kPB + 1) / 2) * (int)sizeof(T) + 15) & ~15;
  static constexpr int OFF_DI = OFF_X + RB * LDX * (int)sizeof(T);
  static constexpr int LDI = 17;                           // padded row stride of the 16 x 16 inverse blocks
  static constexpr int OFF_LI = (OFF_DI + kPB * (int)sizeof(T) + 15) & ~15;
  static constexpr int LDS_BYTES = ((OFF_LI + kPB * LDI * (int)sizeof(T)) + 15) & ~15;
};

// X <- X L^-T on an LDS-resident block: Xs[RB][LDX] rows, L packed lower in P, dinv = 1 / diag(L), Linv = scratch for
// the inverses of the 16 x 16 diagonal blocks of L; `nchunks` 16-column chunks.
// Rows are independent, so each wave owns its row tiles for the whole sweep and the chunk loop needs NO workgroup
// barrier: chunk J first receives  - sum_{K<J} X_K L_JK'  (MFMA, left-looking), then is multiplied by inv(L_JJ)' (MFMA
// again: the per-row substitution of the first version serialised 16 steps per chunk on the vector ALU behind two
// barriers).  The 16 x 16 inverses are formed once per block by 16 lanes each (forward substitution of a unit column).
template <typename T>
__device__ __forceinline__ void trsm_prepare(const T* __restrict__ P, const T* __restrict__ dinv, T* __restrict__ Linv, int nchunks,
                                             int tid) {
  using Cfg = TrsmCfg<T>;
  constexpr int LI = Cfg::LDI;
  if (tid < 16 * nchunks) {
    // column j of inv(L_JJ) by forward substitution, COLUMN-oriented: once x_k is known every pending row takes its update at
    // once (independent multiply-adds) -- the row-oriented form summed k < i terms one after the other for each i: a chain of
    // 136 dependent operations instead of 32
    const int j = tid & 15, j0 = 16 * (tid >> 4);
    T sacc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) sacc[i] = (i == j) ? T(1) : T(0);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const T xk = sacc[k] * dinv[j0 + k];
      Linv[(j0 + k) * LI + j] = xk;
#pragma unroll
      for (int i = k + 1; i < 16; ++i) sacc[i] -= P[pidx(j0, j0) + i * j0 + (i * (i + 1)) / 2 + k] * xk;
    }
  }
  __syncthreads();
}

// the sweep proper; Linv from trsm_prepare.  Ends with a workgroup barrier (Xs complete for everybody).
template <typename T>
__device__ __forceinline__ void trsm_sweep(T* __restrict__ Xs, const T* __restrict__ P, const T* __restrict__ Linv, int nchunks,
                                           int lane, int wave) {
  using Cfg = TrsmCfg<T>;
  using acc4 = typename Mfma<T>::acc4;
  constexpr int NT = Cfg::RB / 64;  // row tiles per wave
  constexpr int LI = Cfg::LDI;
  const int fr = lane & 15, fq = lane >> 4;
  for (int J = 0; J < nchunks; ++J) {
    acc4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int v = 0; v < 4; ++v)
        acc[t][v] = Xs[(16 * (wave + kWaves * t) + Mfma<T>::crow(lane, v)) * Cfg::LDX + 16 * J + (lane & 15)];
    for (int K = 0; K < J; ++K) {
      T fl[4], fx[NT][4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        fl[ks] = P[pidx(16 * J + fr, 16 * K + 4 * ks + fq)];
#pragma unroll
        for (int t = 0; t < NT; ++t) fx[t][ks] = Xs[(16 * (wave + kWaves * t) + fr) * Cfg::LDX + 16 * K + 4 * ks + fq];
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = Mfma<T>::mma(-fx[t][ks], fl[ks], acc[t]);
    }
    // C layout -> LDS -> A fragments (wave-local: LDS operations of one wave complete in order)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int v = 0; v < 4; ++v)
        Xs[(16 * (wave + kWaves * t) + Mfma<T>::crow(lane, v)) * Cfg::LDX + 16 * J + (lane & 15)] = acc[t][v];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    acc4 o[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) o[t] = acc4{T(0), T(0), T(0), T(0)};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const T fi = Linv[(16 * J + fr) * LI + 4 * ks + fq];  // B[k][j] = inv(L_JJ)[j][k]
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const T fu = Xs[(16 * (wave + kWaves * t) + fr) * Cfg::LDX + 16 * J + 4 * ks + fq];
        o[t] = Mfma<T>::mma(fu, fi, o[t]);
      }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int v = 0; v < 4; ++v)
        Xs[(16 * (wave + kWaves * t) + Mfma<T>::crow(lane, v)) * Cfg::LDX + 16 * J + (lane & 15)] = o[t][v];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
}

// X <- X L^-1 (the OTHER triangular solve: G L = Z), same conventions as trsm_sweep; chunks from the last to the first:
// G_J = (Z_J - sum_{K>J} G_K L_KJ) inv(L_JJ).  With trsm_sweep before it this applies A^-1 = L^-T L^-1 to every row.
template <typename T>
__device__ __forceinline__ void trsm_sweep_back(T* __restrict__ Xs, const T* __restrict__ P, const T* __restrict__ Linv, int nchunks,
                                                int lane, int wave) {
  using Cfg = TrsmCfg<T>;
  using acc4 = typename Mfma<T>::acc4;
  constexpr int NT = Cfg::RB / 64;
  constexpr int LI = Cfg::LDI;
  const int fr = lane & 15, fq = lane >> 4;
  for (int J = nchunks - 1; J >= 0; --J) {
    acc4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int v = 0; v < 4; ++v)
        acc[t][v] = Xs[(16 * (wave + kWaves * t) + Mfma<T>::crow(lane, v)) * Cfg::LDX + 16 * J + (lane & 15)];
    for (int K = J + 1; K < nchunks; ++K) {
      T fl[4], fx[NT][4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        fl[ks] = P[pidx(16 * K + 4 * ks + fq, 16 * J + fr)];  // B[k][j] = L[16K + k][16J + j]
#pragma unroll
        for (int t = 0; t < NT; ++t) fx[t][ks] = Xs[(16 * (wave + kWaves * t) + fr) * Cfg::LDX + 16 * K + 4 * ks + fq];
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = Mfma<T>::mma(-fx[t][ks], fl[ks], acc[t]);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int v = 0; v < 4; ++v)
        Xs[(16 * (wave + kWaves * t) + Mfma<T>::crow(lane, v)) * Cfg::LDX + 16 * J + (lane & 15)] = acc[t][v];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    acc4 o[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) o[t] = acc4{T(0), T(0), T(0), T(0)};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const T fi = Linv[(16 * J + 4 * ks + fq) * LI + fr];  // B[k][j] = inv(L_JJ)[k][j]
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const T fu = Xs[(16 * (wave + kWaves * t) + fr) * Cfg::LDX + 16 * J + 4 * ks + fq];
        o[t] = Mfma<T>::mma(fu, fi, o[t]);
      }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int v = 0; v < 4; ++v)
        Xs[(16 * (wave + kWaves * t) + Mfma<T>::crow(lane, v)) * Cfg::LDX + 16 * J + (lane & 15)] = o[t][v];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
}

template <typename T>
__device__ __forceinline__ void trsm_core(T* __restrict__ Xs, const T* __restrict__ P, const T* __restrict__ dinv,
                                          T* __restrict__ Linv, int nchunks, int tid, int lane, int wave) {
  trsm_prepare<T>(P, dinv, Linv, nchunks, tid);
  trsm_sweep<T>(Xs, P, Linv, nchunks, lane, wave);
}

template <typename T>
struct RowSqArgs {      // optional fused epilogue of the marginal stream: var_n = |Y_n|^2 + s_n   (:40-43)
  double* acc;          // [N] running row sums of squares (NULL: off); panel 0 writes, later panels add
  T* var; const T* s;   // last panel: var[n] = acc[n] + s_n
  int noise_kind, N, first, last;
  int64_t grp_ws, grp_s;  // blockIdx.y = regressor of a group: byte stride of acc / var, element stride of s
};

template <typename T>
__global__ __launch_bounds__(kThreads) void trsm_block_kernel(T* Abar, int64_t lda, int p, int row_begin, int nrows_total,
                                                              const int32_t* info, RowSqArgs<T> rs, int64_t grp_ws = 0) {
  __builtin_amdgcn_s_setprio(3);  // latency-critical chain kernel: issue ahead of co-resident Gram waves
  if (const int64_t g = blockIdx.y) {  // regressor of a group: the tall matrix by grp_ws bytes, one status word each
    Abar = ws_shift(Abar, g * grp_ws); info += g;
    rs.acc = ws_shift(rs.acc, g * rs.grp_ws); rs.var = ws_shift(rs.var, g * rs.grp_ws);
    if (rs.s) rs.s += g * rs.grp_s;
  }
  using Cfg = TrsmCfg<T>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* const P = reinterpret_cast<T*>(smem);                    // packed lower triangle of L_pp
  T* const Xs = reinterpret_cast<T*>(smem + Cfg::OFF_X);      // [RB][LDX]
  T* const dinv = reinterpret_cast<T*>(smem + Cfg::OFF_DI);   // 1 / L_pp[c][c]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int r0 = row_begin + blockIdx.x * Cfg::RB;            // first global row of this block
  const int nr = min(Cfg::RB, nrows_total - r0);              // a multiple of 128 rows in every caller: whole vectors
  const T* Lpp = Abar + (int64_t)p * kPB * lda + (int64_t)p * kPB;
  T* Xg = Abar + (int64_t)p * kPB * lda + r0;
  {
    // both blocks in flight at once (one round of memory latency instead of eight)
    BlockVec<T, kPB> lb;
    BlockVec<T, Cfg::RB> xb;
    lb.load(Lpp, lda, tid);
    xb.load(Xg, lda, tid);
    if (*info != 0) return;
    lb.to_packed_lower(P, tid);
    xb.to_rows(Xs, Cfg::LDX, nr, tid);
  }
  __syncthreads();
  if (tid < kPB) dinv[tid] = T(1) / P[pidx(tid, tid)];
  __syncthreads();

  trsm_core<T>(Xs, P, dinv, reinterpret_cast<T*>(smem + Cfg::OFF_LI), 8, tid, lane, wave);
  if (rs.acc != nullptr && tid < Cfg::RB) {
    // the finished 128 columns of this row never change again: fold them into the row's sum of squares now
    const int n = r0 - row_begin + tid;
    if (n < rs.N) {
      const T* xr = Xs + tid * Cfg::LDX;
      double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
#pragma unroll 8
      for (int c = 0; c < kPB; c += 4) {
        const double v0 = (double)xr[c], v1 = (double)xr[c + 1], v2 = (double)xr[c + 2], v3 = (double)xr[c + 3];
        q0 += v0 * v0; q1 += v1 * v1; q2 += v2 * v2; q3 += v3 * v3;
      }
      double tot = (q0 + q1) + (q2 + q3);
      if (!rs.first) tot += rs.acc[n];
      if (rs.last) rs.var[n] = (T)tot + ((rs.noise_kind == NOISE_DIAGONAL) ? rs.s[n] : rs.s[0]);
      else rs.acc[n] = tot;
    }
  }
  {
    using BV = BlockVec<T, Cfg::RB>;
#pragma unroll 4
    for (int u = 0; u < BV::NV; ++u) {
      const int vi = u * kThreads + tid;
      const int c = vi / BV::VPC, rr = (vi % BV::VPC) * BV::VEC;
      if (rr < nr) {
        typename BV::vecT o;
#pragma unroll
        for (int e = 0; e < BV::VEC; ++e) o[e] = Xs[(rr + e) * Cfg::LDX + c];
        *reinterpret_cast<typename BV::vecT*>(Xg + (int64_t)c * lda + rr) = o;
      }
    }
  }
}

// ---- wavefront back substitution  m = L^-T u  over NC workgroups per right-hand side -------------------------------
// Workgroup q owns row block q: it prefetches its diagonal block, subtracts Tf[q rows, p cols] m_p for every finished
// block p > q as soon as m_p arrives (the Tf sub-block is already in registers by then), solves its 128 x 128 triangle
// (one wave, 8 pivots per step, rows of L prefetched, reciprocal pivots in registers) and publishes m_q.
// Hand-off = 8-byte {payload, tag} granules written with agent-scope (write-through) stores and polled with agent-scope
// loads by the lanes that need them: ONE memory round trip per hop, no flag, no fence, no memset -- the tag is a
// per-launch epoch and the exchange buffer is only ever written by this kernel.  (q, rhs) come from a ticket taken in
// START order, so a workgroup only waits for workgroups that are already running: no assumption on dispatch order or
// co-residency.  Grid = NC x S workgroups (S right-hand sides: posterior 1, weight draws S).
// Measured at D = 2048, f32: one streaming workgroup 445 us -> flag + fence wavefront 280 us -> this form: see DESIGN.md.
template <typename T>
struct WaveSolveArgs {
  const T* Tf; int64_t ldtf;  // DP x DP upper factor U = L' (column-major, unit padding)
  int D, DP;
  const T* rhs; int64_t ldrhs, rhs_inc;  // rhs s, entry j: rhs[s*ldrhs + j*rhs_inc]
  unsigned long long* xchg;              // [S][DP][2] tagged granules (handle-owned, zero at allocation)
  unsigned* ticket;                      // the handle's counter words, all device-side so that a captured launch can be REPLAYED:
                                         // [0] start-order tickets, [2] finished workgroups, [3] launches so far (tag of the
                                         // exchange granules = [3] + 1); the workgroup that finishes last zeroes [0] and [2]
                                         // and bumps [3]
  const T* add; T* out; int64_t ldout;   // out[s*ldout + j] = add[j] + m_j  (j < D); out may be NULL
  // evidence assembly (posterior; S == 1), done by the workgroup that finishes last (q = 0); logpdf may be NULL
  const double* qpart; const double* lpart; int nparts;
  const double* logdet_Lw_dev;
  int noise_kind; const T* s; int N;
  double n_total;                        // > 0: number of observations behind the (summed) statistics (N-sharded finish)
  double* logpdf; int32_t* info; const int32_t* chol_info;
  // status precedence of the reference (:78 prior, :79 noise, :86 posterior): chol_info is seeded with the prior's status
  const int32_t* prior_info;     // may be NULL
  const unsigned* noise_info;    // colstats_kernel's atomicMin target (0xFFFFFFFF = ok); may be NULL
  // a group of regressors in one launch (posterior_large_group): system g of `group` has its OWN factor and statistics.
  // 0: one factor, gridDim.y right-hand sides.  Otherwise the pointers above are regressor 0's and regressor g's are
  // ws_stride bytes further for everything that lives in the per-regressor workspace (Tf, rhs, qpart, lpart,
  // logdet_Lw_dev, chol_info, prior_info, noise_info), add_stride / s_stride elements for the caller's arrays, out by
  // ldout, logpdf and info by one.
  int group;
  int64_t ws_stride, add_stride, s_stride;
  // 1: nobody wants m (logpdf without posterior: out == NULL) -- the evidence needs |u|^2 and the factor's diagonal only, both
  // there once the factorisation is; Tf may then be the LOWER factor itself (only its diagonal is read), no hop is made
  int evidence_only;
};

__device__ __forceinline__ void xchg_put(unsigned long long* g, float v, unsigned tag) {
  __hip_atomic_store(g, ((unsigned long long)tag << 32) | (unsigned)__float_as_int(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void xchg_put(unsigned long long* g, double v, unsigned tag) {
  __hip_atomic_store(g, ((unsigned long long)tag << 32) | (unsigned)__double2loint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(g + 1, ((unsigned long long)tag << 32) | (unsigned)__double2hiint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long xchg_poll(const unsigned long long* g, unsigned tag) {
  unsigned long long w = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while ((unsigned)(w >> 32) != tag) {
    __builtin_amdgcn_s_sleep(1);
    w = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return w;
}
__device__ __forceinline__ void xchg_get(const unsigned long long* g, unsigned tag, float* v) {
  *v = __int_as_float((int)(unsigned)xchg_poll(g, tag));
}
__device__ __forceinline__ void xchg_get(const unsigned long long* g, unsigned tag, double* v) {
  const unsigned lo = (unsigned)xchg_poll(g, tag), hi = (unsigned)xchg_poll(g + 1, tag);
  *v = __hiloint2double((int)hi, (int)lo);
}

// ---- 16 x 16 upper-triangular tiles inverted in registers: one row per lane, the four 16-lane DPP rows of a wave each on their
// own tile.  u[k] = U(c, k) (k >= c; the lane's row c), dinv = 1 / U(c, c); out: v[k] = Uinv(c, k).  Row C of the scaled inverse
// V~ = diag(U) Uinv is final once the rows below it have been folded in (C runs down), and every broadcast of it is the DPP
// source of the multiply-add itself (blr_panel.hpp).
template <typename T, int C>
__device__ __forceinline__ void tri16_inv_col(const T (&u)[16], T (&v)[16], T dinv, int c) {
  if constexpr (C >= 1) {
    const T rsC = mov_bc16_gap<C>(dinv);
    const T t = (c < C) ? -u[C] * rsC : T(0);  // -U(c, C) / U(C, C) for the rows above row C
#pragma unroll
    for (int k = C; k < 16; ++k) fmac_bc16_gap<C>(v[k], v[k], t);
    tri16_inv_col<T, C - 1>(u, v, dinv, c);
  }
}
template <typename T>
__device__ __forceinline__ void tri16_inv(const T (&u)[16], T (&v)[16], T dinv, int c) {
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = (k == c) ? T(1) : T(0);
  tri16_inv_col<T, 15>(u, v, dinv, c);
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] *= dinv;
}
// x(c) = sum_k Uinv(c, k) r(k) for the tile whose 16 right-hand-side entries sit in this lane's DPP row of `rr`
template <typename T>
__device__ __forceinline__ T tri16_apply(const T (&v)[16], T rr) {
  T x0 = T(0), x1 = T(0), x2 = T(0), x3 = T(0);
  fmac_bc16_gap<0>(x0, rr, v[0]);   fmac_bc16<1>(x1, rr, v[1]);   fmac_bc16<2>(x2, rr, v[2]);   fmac_bc16<3>(x3, rr, v[3]);
  fmac_bc16<4>(x0, rr, v[4]);       fmac_bc16<5>(x1, rr, v[5]);   fmac_bc16<6>(x2, rr, v[6]);   fmac_bc16<7>(x3, rr, v[7]);
  fmac_bc16<8>(x0, rr, v[8]);       fmac_bc16<9>(x1, rr, v[9]);   fmac_bc16<10>(x2, rr, v[10]); fmac_bc16<11>(x3, rr, v[11]);
  fmac_bc16<12>(x0, rr, v[12]);     fmac_bc16<13>(x1, rr, v[13]); fmac_bc16<14>(x2, rr, v[14]); fmac_bc16<15>(x3, rr, v[15]);
  return (x0 + x1) + (x2 + x3);
}
// one 16-row tile of the back substitution of a 128-row block (wave 0; b0 / b1: rows lane and 64 + lane of the right-hand side).
// The tile's solution comes from its inverted diagonal tile; the rows above the tile then take the tile's 16 columns of U in
// one go (their entries of L are requested before the solution exists, so only readlane + multiply-add follow it).
template <typename T, int TT>
__device__ __forceinline__ void back_tile(const T* __restrict__ P, const T (&vA)[16], const T (&vB)[16], T& b0, T& b1, int lane) {
  constexpr bool HI = TT >= 4;
  T l0[16], l1[16];
  if constexpr (TT >= 1) {
#pragma unroll
    for (int c = 0; c < 16; ++c) l0[c] = P[pidx(16 * TT + c, 0) + lane];       // L(16 TT + c, lane): rows 0..63 (masked below)
  }
  if constexpr (TT >= 5) {
#pragma unroll
    for (int c = 0; c < 16; ++c) l1[c] = P[pidx(16 * TT + c, 0) + 64 + lane];  // rows 64..127
  }
  T& rr = HI ? b1 : b0;
  const T x = tri16_apply<T>(HI ? vB : vA, rr);
  if ((lane >> 4) == (TT & 3)) rr = x;
  T d0 = T(0), d1 = T(0);
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const T xc = readlane(x, 16 * (TT & 3) + c);
    if constexpr (TT >= 1) d0 += l0[c] * xc;
    if constexpr (TT >= 5) d1 += l1[c] * xc;
  }
  if constexpr (TT >= 1) { if (lane < (TT >= 4 ? 64 : 16 * TT)) b0 -= d0; }
  if constexpr (TT >= 5) { if (lane < 16 * (TT - 4)) b1 -= d1; }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void backsolve_wave_kernel(WaveSolveArgs<T> a) {
  constexpr int P_BYTES = ((kPB * (kPB + 1) / 2) * (int)sizeof(T) + 15) & ~15;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* const P = reinterpret_cast<T*>(smem);                 // packed lower triangle of L_qq
  T* const mp = reinterpret_cast<T*>(smem + P_BYTES);      // [128] m_p
  T* const part = mp + kPB;                                // [256] partial sums
  double* const scr = reinterpret_cast<double*>(part + 2 * kPB);  // [8]
  int* const tick = reinterpret_cast<int*>(scr + 8);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int D = a.D, DP = a.DP, NC = DP / kPB;
  if (tid == 0) {
    tick[0] = (int)atomicAdd(a.ticket, 1u);
    tick[1] = (int)__hip_atomic_load(a.ticket + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // stable until the launch's last exit
  }
  __syncthreads();
  const int ticket = tick[0];
  const unsigned epoch = (unsigned)tick[1] % 0xFFFFFFFFu + 1u;  // never 0: that is the never-written state of a granule
  // every exit goes through here: the last workgroup out re-arms the counters for the next launch (or the next replay)
  auto leave = [&]() {
    __syncthreads();
    if (tid == 0) {
      const unsigned total = gridDim.x * gridDim.y;
      if (atomicAdd(a.ticket + 2, 1u) == total - 1u) {
        __hip_atomic_store(a.ticket + 0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.ticket + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.ticket + 3, (unsigned)tick[1] + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  };
  const int q = NC - 1 - ticket % NC;
  const int64_t sidx = ticket / NC;
  if (a.group) {  // this workgroup's regressor of the group (uniform over the workgroup)
    const int64_t wb = sidx * a.ws_stride;
    a.Tf = ws_shift(a.Tf, wb); a.qpart = ws_shift(a.qpart, wb); a.lpart = ws_shift(a.lpart, wb);
    a.logdet_Lw_dev = ws_shift(a.logdet_Lw_dev, wb); a.chol_info = ws_shift(a.chol_info, wb);
    a.prior_info = ws_shift(a.prior_info, wb); a.noise_info = ws_shift(a.noise_info, wb);
    a.add += sidx * a.add_stride;
    if (a.s) a.s += sidx * a.s_stride;
    if (a.logpdf) a.logpdf += sidx;
    if (a.info) a.info += sidx;
  }
  const bool evidence = a.info != nullptr && q == 0 && (sidx == 0 || a.group);
  const double kNaN = __longlong_as_double(0x7ff8000000000000LL);
  {
    int st = 0;
    if (a.prior_info && *a.prior_info != 0) st = *a.prior_info;
    else if (a.noise_info && *a.noise_info != 0xFFFFFFFFu) st = (int)*a.noise_info;
    else if (a.chol_info && *a.chol_info != 0) st = *a.chol_info;
    if (st != 0) {  // uniform over the grid: nobody waits
      if (evidence && tid == 0) { *a.info = st; if (a.logpdf) *a.logpdf = kNaN; }
      leave();
      return;
    }
  }
  const T* rhs = a.rhs + sidx * a.ldrhs;
  unsigned long long* xg = a.xchg + (sidx * DP) * 2;
  if (a.evidence_only && !evidence) {  // (uniform over the workgroup; nobody waits for anybody in this mode)
    leave();
    return;
  }

  if (!a.evidence_only) load_upper_block_to_packed(P, a.Tf + (int64_t)q * kPB * a.ldtf + (int64_t)q * kPB, a.ldtf, tid);
  // wave 0 solves: lane l owns rows l and l + 64 of the block
  const int i0 = lane, i1 = lane + 64;
  T b0 = T(0), b1 = T(0);
  if (wave == 0) {
    const int j0 = q * kPB + i0, j1 = q * kPB + i1;
    b0 = (j0 < D) ? rhs[(int64_t)j0 * a.rhs_inc] : T(0);
    b1 = (j1 < D) ? rhs[(int64_t)j1 * a.rhs_inc] : T(0);
  }
  double uu = 0.0, ld = 0.0, qs = 0.0, ls = 0.0;
  if (evidence) {  // |u|^2, logdet A and the column-statistics partials: off the critical path (this workgroup waits longest)
    for (int j = tid; j < D; j += kThreads) {
      const T uj = rhs[(int64_t)j * a.rhs_inc];
      uu += (double)uj * (double)uj;
      ld += log((double)a.Tf[(int64_t)j * a.ldtf + j]);
    }
    for (int i = tid; i < a.nparts; i += kThreads) { qs += a.qpart[i]; ls += a.lpart[i]; }
    uu = block_allreduce(uu, scr, tid);
    ld = 2.0 * block_allreduce(ld, scr, tid);
    qs = block_allreduce(qs, scr, tid);
    ls = block_allreduce(ls, scr, tid);
  }
  auto write_evidence = [&]() {
    *a.info = 0;
    if (a.logpdf) {
      const double LOG2PI = 1.8378770664093454835606594728112;
      const double nobs = a.n_total > 0.0 ? a.n_total : (double)a.N;
      const double logdet_Sy = (a.noise_kind == NOISE_DIAGONAL) ? ls : nobs * log((double)a.s[0]);
      *a.logpdf = -0.5 * (nobs * LOG2PI + logdet_Sy + qs + ld - *a.logdet_Lw_dev - uu);
    }
  };
  if (a.evidence_only) {  // (only the evidence workgroup of each system gets here)
    if (tid == 0) write_evidence();
    leave();
    return;
  }
  __syncthreads();  // P complete
  // the eight 16 x 16 diagonal tiles of U_qq inverted (wave 0, DPP row t % 4 holds tile t: vA tiles 0-3, vB tiles 4-7): off the
  // critical path -- this workgroup waits for its predecessors anyway -- and it takes the 128 serial pivots (3.8 us of a
  // 6 us hop) out of the chain
  T vA[16], vB[16];
  if (wave == 0) {
    const int c = lane & 15, tq = lane >> 4;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int d0 = 16 * (tq + 4 * pass);  // first row / column of this lane's tile
      T u[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) u[k] = (k >= c) ? P[pidx(d0 + k, d0 + c)] : T(0);  // U(c, k) = L(k, c)
      const T dinv = T(1) / P[pidx(d0 + c, d0 + c)];
      if (pass == 0) tri16_inv<T>(u, vA, dinv, c);
      else tri16_inv<T>(u, vB, dinv, c);
    }
  }

  const int r = tid & (kPB - 1), half = tid >> 7;
  T acc = T(0);
  for (int p = NC - 1; p > q; --p) {
    // the factor sub-block does not depend on m_p: have it in registers before waiting
    const T* tp = a.Tf + ((int64_t)p * kPB + half * 64) * a.ldtf + (int64_t)q * kPB + r;
    T tv[64];
#pragma unroll
    for (int c = 0; c < 64; ++c) tv[c] = tp[(int64_t)c * a.ldtf];
    if (tid < kPB) {
      T v;
      xchg_get(xg + (int64_t)(p * kPB + tid) * 2, epoch, &v);
      mp[tid] = v;
    }
    __syncthreads();
    T a0 = T(0), a1 = T(0), a2 = T(0), a3 = T(0);
#pragma unroll
    for (int c = 0; c < 64; c += 4) {
      a0 += tv[c] * mp[half * 64 + c];
      a1 += tv[c + 1] * mp[half * 64 + c + 1];
      a2 += tv[c + 2] * mp[half * 64 + c + 2];
      a3 += tv[c + 3] * mp[half * 64 + c + 3];
    }
    acc += (a0 + a1) + (a2 + a3);
    __syncthreads();  // mp is rewritten by the next block
  }
  part[tid] = acc;
  __syncthreads();
  if (wave == 0) {
    b0 -= part[i0] + part[kPB + i0];
    b1 -= part[i1] + part[kPB + i1];
    // back substitution, 16 rows at a time from the last tile up
    back_tile<T, 7>(P, vA, vB, b0, b1, lane);
    back_tile<T, 6>(P, vA, vB, b0, b1, lane);
    back_tile<T, 5>(P, vA, vB, b0, b1, lane);
    back_tile<T, 4>(P, vA, vB, b0, b1, lane);
    back_tile<T, 3>(P, vA, vB, b0, b1, lane);
    back_tile<T, 2>(P, vA, vB, b0, b1, lane);
    back_tile<T, 1>(P, vA, vB, b0, b1, lane);
    back_tile<T, 0>(P, vA, vB, b0, b1, lane);
    const int j0 = q * kPB + i0, j1 = q * kPB + i1;
    xchg_put(xg + (int64_t)j0 * 2, b0, epoch);
    xchg_put(xg + (int64_t)j1 * 2, b1, epoch);
    if (a.out) {
      if (j0 < D) a.out[sidx * a.ldout + j0] = a.add[j0] + b0;
      if (j1 < D) a.out[sidx * a.ldout + j1] = a.add[j1] + b1;
    }
  }
  if (evidence && tid == 0) write_evidence();
  leave();
}

// logdet of a factored (DP x DP lower, ld) matrix restricted to the first D diagonal entries: 2 sum log L_ii
template <typename T>
__global__ __launch_bounds__(kThreads) void logdet_kernel(const T* Lf, int64_t ld, int D, double* out, int64_t grp_ws = 0,
                                                          const int32_t* info_src = nullptr, int32_t* info_dst = nullptr) {
  if (const int64_t g = blockIdx.x) {  // regressor of a group: everything lives in its workspace (byte stride)
    Lf = ws_shift(Lf, g * grp_ws); out = ws_shift(out, g * grp_ws);
    info_src = ws_shift(info_src, g * grp_ws); info_dst = ws_shift(info_dst, g * grp_ws);
  }
  __shared__ double scr[8];
  double v = 0.0;
  for (int j = threadIdx.x; j < D; j += kThreads) v += log((double)Lf[(int64_t)j * ld + j]);
  v = block_allreduce(v, scr, threadIdx.x);
  if (threadIdx.x == 0) {
    *out = 2.0 * v;
    if (info_dst) *info_dst = *info_src;  // (the status of this factorisation seeds the next one's)
  }
}

// logdet of a diagonal (kind 2) or of an upper factor's diagonal (kind 1); info = first non-positive entry
// Scratch initialisation that rides along with prior_diag_kernel (every dependent dispatch costs ~5 us on this part: three
// hipMemsetAsync and a 4-byte device-to-device copy were 20 us of an update).  All pointers optional.
struct ScratchInit {
  unsigned* words16 = nullptr;   // 16 words set to zero BEFORE the kernel's own results land in them
  unsigned* ones = nullptr;      // one word set to 0xFFFFFFFF
  double* zeros = nullptr;       // nzeros doubles set to zero (all workgroups of the launch share the work)
  long long nzeros = 0;
  int32_t* info_copy = nullptr;  // receives the same status as `info`
};

template <typename T>
__global__ __launch_bounds__(kThreads) void prior_diag_kernel(const T* Lw, int64_t ldl, int kind, int D, double* out,
                                                              int32_t* info, ScratchInit init = ScratchInit(), int64_t grp_Lw = 0,
                                                              int64_t grp_ws = 0) {
  if (const int64_t g = blockIdx.y) {  // regressor of a group: Lw by elements, everything else lives in its workspace
    Lw += g * grp_Lw;
    out = ws_shift(out, g * grp_ws); info = ws_shift(info, g * grp_ws);
    init.words16 = ws_shift(init.words16, g * grp_ws); init.ones = ws_shift(init.ones, g * grp_ws);
    init.zeros = ws_shift(init.zeros, g * grp_ws); init.info_copy = ws_shift(init.info_copy, g * grp_ws);
  }
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < init.nzeros; i += (long long)gridDim.x * kThreads)
    init.zeros[i] = 0.0;
  if (blockIdx.x != 0) return;
  if (init.words16 && threadIdx.x < 16) init.words16[threadIdx.x] = (init.words16 + threadIdx.x == init.ones) ? 0xFFFFFFFFu : 0u;
  if (init.ones && threadIdx.x == 0 && (init.words16 == nullptr || init.ones < init.words16 || init.ones >= init.words16 + 16))
    *init.ones = 0xFFFFFFFFu;
  __shared__ double scr[8];
  __shared__ int iscr[8];
  double v = 0.0;
  int bad = 0x7fffffff;
  for (int j = threadIdx.x; j < D; j += kThreads) {
    const T d = (kind == PRIOR_DIAGONAL) ? Lw[j] : Lw[(int64_t)j * ldl + j];
    if (d > T(0)) v += log((double)d);
    else bad = min(bad, j + 1);
  }
  bad = block_min_int(bad, iscr, threadIdx.x);  // (barriers: the words above are zero before thread 0 writes below)
  v = block_allreduce(v, scr, threadIdx.x);
  if (threadIdx.x == 0) {
    if (out) *out = (kind == PRIOR_DIAGONAL) ? v : 2.0 * v;
    *info = (bad == 0x7fffffff) ? 0 : bad;
    if (init.info_copy) *init.info_copy = (bad == 0x7fffffff) ? 0 : bad;
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void prior_copy_kernel(const T* Lw, int64_t ldl, int D, int DP, T* W, int64_t ldw,
                                                              int64_t grp_Lw = 0, int64_t grp_W = 0) {
  Lw += (int64_t)blockIdx.y * grp_Lw;  // blockIdx.y: regressor of a group
  W += (int64_t)blockIdx.y * grp_W;
  for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < (int64_t)DP * DP; e += (int64_t)gridDim.x * kThreads) {
    const int col = (int)(e / DP), row = (int)(e % DP);
    if (row < col) continue;
    T v;
    if (row < D) v = Lw[(int64_t)row * ldl + col];  // upper entry (col, row)
    else v = (row == col) ? T(1) : T(0);
    W[(int64_t)col * ldw + row] = v;
  }
}

// T = L' : upper factor, column-major, strictly-lower part zero (Tout2 optional second destination, D2 x D2)
template <typename T>
__global__ __launch_bounds__(kThreads) void transpose_out_kernel(const T* Lf, int64_t ld, int D, T* Tout, int64_t ldt,
                                                                 T* Tout2, int64_t ldt2, int D2, int64_t zstride = 0,
                                                                 int64_t zstride2 = 0, const int32_t* st_prior = nullptr,
                                                                 const unsigned* st_noise = nullptr, const int32_t* st_chol = nullptr,
                                                                 int64_t st_stride = 0) {
  __shared__ T tile[32][33];
  Lf += (int64_t)blockIdx.z * zstride;  // blockIdx.z: regressor of a group (Lf and Tout in per-regressor workspaces)
  Tout += (int64_t)blockIdx.z * zstride;
  if (Tout2) Tout2 += (int64_t)blockIdx.z * zstride2;
  // the CALLER's factor is written only for a regressor whose update succeeded (status words of its workspace, byte stride
  // st_stride): a failed update leaves mw_post / T_post as they were, at every D (include/blr_mi355x.h; the in-place state of
  // blr_update_factor_* survives a bad batch)
  if (Tout2) {
    const int64_t sb = (int64_t)blockIdx.z * st_stride;
    const int32_t* p0 = ws_shift(st_prior, sb);
    const unsigned* p1 = ws_shift(st_noise, sb);
    const int32_t* p2 = ws_shift(st_chol, sb);
    if ((p0 && *p0 != 0) || (p1 && *p1 != 0xFFFFFFFFu) || (p2 && *p2 != 0)) Tout2 = nullptr;
  }
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;  // bx: row block of L, by: col block of L
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) {
    const int row = bx + tx, col = by + k;
    tile[k][tx] = (row < D && col < D && row >= col) ? Lf[(int64_t)col * ld + row] : T(0);
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int trow = by + tx, tcol = bx + k;  // T[trow, tcol] = L[tcol, trow]
    if (trow < D && tcol < D) Tout[(int64_t)tcol * ldt + trow] = tile[tx][k];
    if (Tout2 && trow < D2 && tcol < D2) Tout2[(int64_t)tcol * ldt2 + trow] = tile[tx][k];
  }
}

// ---- large-D weight draws ( reference :46-52: w = mw + chol(Lw).U \\ z ) ---------------------------------------------
// upper factor (D x D, ldu) -> DP x DP upper, zero below the diagonal, unit padding: the Tf operand of the wavefront solve
template <typename T>
__global__ __launch_bounds__(kThreads) void upper_pad_kernel(const T* U, int64_t ldu, int D, int DP, T* Tf, int64_t ldtf) {
  for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < (int64_t)DP * DP; e += (int64_t)gridDim.x * kThreads) {
    const int col = (int)(e / DP), row = (int)(e % DP);
    T v = T(0);
    if (row <= col && col < D) v = U[(int64_t)col * ldu + row];
    else if (row == col) v = T(1);
    Tf[(int64_t)col * ldtf + row] = v;
  }
}
// diagonal prior precision d: w = mw + z / sqrt(d)
template <typename T>
__global__ __launch_bounds__(kThreads) void diag_sample_kernel(const T* mw, const T* d, const T* Z, int64_t ldz, T* W, int64_t ldw,
                                                                int D, int64_t S) {
  for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < (int64_t)D * S; e += (int64_t)gridDim.x * kThreads) {
    const int64_t sidx = e / D;
    const int j = (int)(e % D);
    W[sidx * ldw + j] = mw[j] + Z[sidx * ldz + j] / sqrt(d[j]);
  }
}

// ---- large-D marginal stream ( reference :33, :40-43 ) ------------------------------------------------------------------
// mean_n = x_n'mw is a GEMV stream.  var_n = |L^-1 x_n|^2 + s_n with L = U' needs the triangular solve for every
// input: the inputs are laid out as ROWS below L in one tall matrix  Ybar = [L ; X']  and pushed through the same
// panel machinery as the rhs row of the factorisation (trsm_block_kernel + MFMA trailing updates), giving
// Y = X' L^-T; the variance is the row sum of squares.

// element (d, n) of X -> mean[n]; also (optionally) writes row DP + n of Ybar (transpose fill) -- one pass over X
template <typename T>
struct MeanFillArgs {
  const T* X; int64_t ldx; int layout;
  const T* mw; T* mean;          // mean may be NULL
  T* Ybar; int64_t ldy; int row0;  // Ybar may be NULL
  int D, DP, N;
  int64_t grp_X, grp_mw, grp_ws;  // blockIdx.y = regressor of a group: element strides of X / mw, byte stride of mean / Ybar
};

template <typename T>
__global__ __launch_bounds__(kThreads) void mean_fill_kernel(MeanFillArgs<T> a) {
  if (const int64_t g = blockIdx.y) {
    a.X += g * a.grp_X; a.mw += g * a.grp_mw;
    a.mean = ws_shift(a.mean, g * a.grp_ws); a.Ybar = ws_shift(a.Ybar, g * a.grp_ws);
  }
  constexpr int VEC = Mfma<T>::VEC;
  constexpr int LDT = 68;  // row stride: 16-byte aligned rows, 4-way (not 16-way) conflicts on the transposing writes
  typedef T vecT __attribute__((ext_vector_type(Mfma<T>::VEC)));
  __shared__ __attribute__((aligned(16))) T tile[64][LDT];
  const int tid = threadIdx.x;
  const int n0 = blockIdx.x * 64;
  const int tn = tid & 63, tq = tid >> 6;
  double macc = 0.0;  // thread (tn, tq) accumulates mean of column n0 + tn over d = tq, tq + 4, ...
  // ColVecs with 16-byte aligned columns: 16-byte loads along d, 16-byte stores along n, next tile's loads in flight
  const bool vec = a.layout == LAYOUT_COLVECS && (a.D % VEC) == 0 && (a.ldx % VEC) == 0 && ((uintptr_t)a.X % 16) == 0;
  constexpr int VPR = 64 / VEC;                 // vectors per tile row
  constexpr int NVT = 64 * 64 / (VEC * kThreads);  // vectors per thread per tile
  vecT pre[NVT];
  auto prefetch = [&](int d0) {
#pragma unroll
    for (int u = 0; u < NVT; ++u) {
      const int vi = u * kThreads + tid;
      const int d = d0 + (vi % VPR) * VEC, n = n0 + vi / VPR;
      pre[u] = (d < a.D && n < a.N) ? *reinterpret_cast<const vecT*>(a.X + (int64_t)n * a.ldx + d) : vecT(T(0));
    }
  };
  if (vec) prefetch(0);
  for (int d0 = 0; d0 < a.DP; d0 += 64) {
    __syncthreads();
    if (vec) {
#pragma unroll
      for (int u = 0; u < NVT; ++u) {
        const int vi = u * kThreads + tid;
        const int dd0 = (vi % VPR) * VEC, nn = vi / VPR;
#pragma unroll
        for (int e = 0; e < VEC; ++e) tile[dd0 + e][nn] = pre[u][e];
      }
      if (d0 + 64 < a.DP) prefetch(d0 + 64);
    } else {
      // load a 64 (d) x 64 (n) tile, coalesced along the contiguous axis of the layout
      for (int e = tid; e < 64 * 64; e += kThreads) {
        int dd, nn;
        if (a.layout == LAYOUT_COLVECS) { dd = e & 63; nn = e >> 6; }
        else                            { nn = e & 63; dd = e >> 6; }
        const int d = d0 + dd, n = n0 + nn;
        T v = T(0);
        if (d < a.D && n < a.N) v = (a.layout == LAYOUT_COLVECS) ? a.X[(int64_t)n * a.ldx + d] : a.X[(int64_t)d * a.ldx + n];
        tile[dd][nn] = v;
      }
    }
    __syncthreads();
    if (a.mean) {
      for (int dd = tq; dd < 64; dd += 4) {
        const int d = d0 + dd;
        if (d < a.D) macc += (double)tile[dd][tn] * (double)a.mw[d];
      }
    }
    if (a.Ybar) {  // Ybar[row0 + n, d]: rows contiguous -> 16-byte stores along n (row0, n0, ldy are multiples of 64)
#pragma unroll
      for (int u = 0; u < NVT; ++u) {
        const int vj = u * kThreads + tid;
        const int nn0 = (vj % VPR) * VEC, dd = vj / VPR;
        const int d = d0 + dd;
        if (d < a.DP)  // padding rows/cols are zero
          *reinterpret_cast<vecT*>(a.Ybar + (int64_t)d * a.ldy + a.row0 + n0 + nn0) = *reinterpret_cast<const vecT*>(&tile[dd][nn0]);
      }
    }
  }
  if (a.mean) {
    __syncthreads();
    double* red = reinterpret_cast<double*>(&tile[0][0]);  // 4 x 64 doubles fit easily
    red[tq * 64 + tn] = macc;
    __syncthreads();
    if (tq == 0 && n0 + tn < a.N) a.mean[n0 + tn] = (T)(((red[tn] + red[64 + tn]) + red[128 + tn]) + red[192 + tn]);
  }
}

// mean-only stream for ColVecs with 16-byte aligned columns: two columns per wave per step, every load of a step in
// flight before the first use (the colstats pattern); mean_n = x_n'mw accumulated in double, fixed-order butterfly
template <typename T>
__global__ __launch_bounds__(kThreads) void mean_stream_kernel(const T* X, int64_t ldx, const T* mw, T* mean, int D, int N) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int VEC = Mfma<T>::VEC;
  typedef T vecT __attribute__((ext_vector_type(Mfma<T>::VEC)));
  T* const mwl = reinterpret_cast<T*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int d = tid; d < D; d += kThreads) mwl[d] = mw[d];
  __syncthreads();
  const int DV = D / VEC;
  const vecT* mwv = reinterpret_cast<const vecT*>(mwl);
  const int wid = blockIdx.x * kWaves + wave, nw = gridDim.x * kWaves;
  for (int n = 2 * wid; n < N; n += 2 * nw) {
    const bool two = n + 1 < N;
    const vecT* c0 = reinterpret_cast<const vecT*>(X + (int64_t)n * ldx);
    const vecT* c1 = reinterpret_cast<const vecT*>(X + (int64_t)(two ? n + 1 : n) * ldx);
    double mu0 = 0.0, mu1 = 0.0;
    for (int i0 = 0; i0 < DV; i0 += 256) {
      vecT v0[4], v1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * 64 + lane;
        const bool in = i < DV;
        v0[u] = in ? c0[i] : vecT(T(0));
        v1[u] = in ? c1[i] : vecT(T(0));
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * 64 + lane;
        if (i < DV) {
          const vecT m = mwv[i];
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            mu0 += (double)v0[u][e] * (double)m[e];
            mu1 += (double)v1[u][e] * (double)m[e];
          }
        }
      }
    }
    mu0 = wave_allreduce(mu0);
    mu1 = wave_allreduce(mu1);
    if (lane == 0) mean[n] = (T)mu0;
    if (lane == 1 && two) mean[n + 1] = (T)mu1;
  }
}

// L = U' into the top DP x DP block of Ybar (lower, unit padding); U upper column-major (ldu)
template <typename T>
__global__ __launch_bounds__(kThreads) void factor_transpose_fill_kernel(const T* U, int64_t ldu, int D, int DP, T* Ybar,
                                                                         int64_t ldy) {
  __shared__ T tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;  // bx: row block of L, by: col block of L
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) {
    const int ur = by + tx, uc = bx + k;  // U[ur, uc] = L[uc, ur]; coalesced along ur
    tile[k][tx] = (ur < D && uc < D && ur <= uc) ? U[(int64_t)uc * ldu + ur] : T(0);
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int row = bx + tx, col = by + k;  // L[row, col] = U[col, row] = tile[tx][k]
    if (row < DP && col < DP) {
      T v = tile[tx][k];
      if (row >= D || col >= D) v = (row == col) ? T(1) : T(0);
      if (row >= col) Ybar[(int64_t)col * ldy + row] = v;
    }
  }
}

// diagonal prior: var[n] = sum_d x[d,n]^2 / dprior[d] + s_n  (pure stream)
template <typename T>
__global__ __launch_bounds__(kThreads) void var_diag_prior_kernel(const T* X, int64_t ldx, int layout, const T* dprior, int D,
                                                                  int N, const T* s, int noise_kind, T* var) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (layout == LAYOUT_COLVECS) {
    for (int n = blockIdx.x * kWaves + wave; n < N; n += gridDim.x * kWaves) {
      const T* col = X + (int64_t)n * ldx;
      double acc = 0.0;
      for (int d = lane; d < D; d += 64) { const double x = (double)col[d]; acc += x * x / (double)dprior[d]; }
      acc = wave_allreduce(acc);
      if (lane == 0) var[n] = (T)acc + ((noise_kind == NOISE_DIAGONAL) ? s[n] : s[0]);
    }
  } else {
    for (int n = blockIdx.x * kThreads + threadIdx.x; n < N; n += gridDim.x * kThreads) {
      double acc = 0.0;
      for (int d = 0; d < D; ++d) { const double x = (double)X[(int64_t)d * ldx + n]; acc += x * x / (double)dprior[d]; }
      var[n] = (T)acc + ((noise_kind == NOISE_DIAGONAL) ? s[n] : s[0]);
    }
  }
}

// ---- marginal stream for D <= 128 with an upper factor: the same TRSM core, fused with mean and row sum of squares ----
// A workgroup owns a strided set of input tiles of ONE regressor (RB inputs each: 128 in f32, 64 in f64): L = U' is packed
// and its 16 x 16 inverse blocks are formed once, then per tile: the inputs become the ROWS of an LDS block (the next
// tile's loads are already in flight), Y = X'L^-T by the barrier-free MFMA sweep, var_n = |Y_n|^2 + s_n,
// mean_n = x_n'mw (:33, :40-43).  The first version reloaded and re-inverted L for every tile (one tile per workgroup).
template <typename T>
__global__ __launch_bounds__(kThreads) void marginals_mfma_kernel(MarginalArgs<T> a) {
  using Cfg = TrsmCfg<T>;
  constexpr int VEC = Mfma<T>::VEC;
  typedef T vecT __attribute__((ext_vector_type(Mfma<T>::VEC)));
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // with a factor: [P | Xs | dinv | Linv | mw];  mean-only / diagonal prior: [Xs | dinv | mw] (two workgroups per CU)
  const bool use_factor = a.var != nullptr && a.prior_kind == PRIOR_UPPER_FACTOR;
  const bool diag_prior = a.var != nullptr && a.prior_kind == PRIOR_DIAGONAL;
  constexpr int XS_BYTES = (Cfg::RB * Cfg::LDX * (int)sizeof(T) + 15) & ~15;
  T* const P = reinterpret_cast<T*>(smem);
  T* const Xs = reinterpret_cast<T*>(smem + (use_factor ? Cfg::OFF_X : 0));
  T* const dinv = reinterpret_cast<T*>(smem + (use_factor ? Cfg::OFF_DI : XS_BYTES));
  T* const Linv = reinterpret_cast<T*>(smem + Cfg::OFF_LI);
  T* const mwl = reinterpret_cast<T*>(smem + (use_factor ? Cfg::LDS_BYTES : XS_BYTES + kPB * (int)sizeof(T)));  // [128]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int D = a.D, N = a.N;
  const int reg = a.reg0 + blockIdx.y;  // grid.y is limited to 65535: the host launches in chunks
  if (a.info && a.info[reg] != 0) return;
  const T* X = a.X + (int64_t)reg * a.strideX;
  const T* U = a.U + (int64_t)reg * a.strideU;
  const T* s = a.s + (int64_t)reg * a.strides;
  const T* mw = a.mw + (int64_t)reg * a.stridemw;
  const int nchunks = (D + 15) >> 4, DPc = nchunks * 16;
  const int ntiles = (N + Cfg::RB - 1) / Cfg::RB;

  // ---- the first tile's inputs go in flight before anything else (vector path: ColVecs, whole 16-byte vectors)
  const bool vec = a.layout == LAYOUT_COLVECS && D == kPB && (a.ldx % VEC) == 0 && ((uintptr_t)X % 16) == 0;
  constexpr int VPR = kPB / VEC;                           // vectors per input (row of Xs)
  constexpr int NVT = Cfg::RB * kPB / (VEC * kThreads);    // vectors per thread per tile
  vecT pre[NVT];
  auto prefetch = [&](int tile) {
    const int n0 = tile * Cfg::RB;
#pragma unroll
    for (int u = 0; u < NVT; ++u) {
      const int vi = u * kThreads + tid;
      const int r = vi / VPR, c0 = (vi % VPR) * VEC;
      pre[u] = (n0 + r < N) ? *reinterpret_cast<const vecT*>(X + (int64_t)(n0 + r) * a.ldx + c0) : vecT(T(0));
    }
  };
  if (vec && (int)blockIdx.x < ntiles) prefetch(blockIdx.x);

  // ---- once per workgroup: L = U' packed (padding: unit diagonal), reciprocal pivots, inverse blocks, mw
  const bool uvec = use_factor && D == kPB && (a.ldu % VEC) == 0 && ((uintptr_t)U % 16) == 0;
  if (uvec) load_upper_block_to_packed(P, U, a.ldu, tid);  // every load of the block in flight at once
#pragma unroll 1
  for (int base = 0; use_factor && !uvec && base < DPc * DPc; base += kThreads * 8) {
    T v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * kThreads + tid;
      const int r = idx / DPc, c = idx % DPc;  // L[r][c] = U[c, r]: consecutive threads -> consecutive c
      const bool ok = idx < DPc * DPc && c <= r && r < D;
      v[u] = U[ok ? (int64_t)r * a.ldu + c : 0];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * kThreads + tid;
      const int r = idx / DPc, c = idx % DPc;
      if (idx < DPc * DPc && c <= r) P[pidx(r, c)] = (r < D) ? v[u] : (r == c ? T(1) : T(0));
    }
  }
  if (tid < kPB) mwl[tid] = (tid < D) ? mw[tid] : T(0);
  __syncthreads();
  if (use_factor && tid < DPc) dinv[tid] = T(1) / P[pidx(tid, tid)];
  if (diag_prior && tid < kPB) dinv[tid] = (tid < D) ? T(1) / U[tid] : T(0);  // U = the diagonal of the precision
  __syncthreads();
  if (use_factor) trsm_prepare<T>(P, dinv, Linv, nchunks, tid);

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int n0 = tile * Cfg::RB;
    const int nt = min(Cfg::RB, N - n0);
    __syncthreads();  // the previous tile's readers of Xs are done
    if (vec) {
#pragma unroll
      for (int u = 0; u < NVT; ++u) {
        const int vi = u * kThreads + tid;
        const int r = vi / VPR, c0 = (vi % VPR) * VEC;
#pragma unroll
        for (int e = 0; e < VEC; ++e) Xs[r * Cfg::LDX + c0 + e] = pre[u][e];
      }
      if (tile + (int)gridDim.x < ntiles) prefetch(tile + gridDim.x);  // in flight during this tile's sweep
    } else {
#pragma unroll 1
      for (int base = 0; base < Cfg::RB * DPc; base += kThreads * 8) {
        T v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int idx = base + u * kThreads + tid;
          int r, c;
          if (a.layout == LAYOUT_COLVECS) { c = idx % DPc; r = idx / DPc; }
          else                            { r = idx % Cfg::RB; c = idx / Cfg::RB; }
          const bool ok = idx < Cfg::RB * DPc && r < nt && c < D;
          const int64_t addr = (a.layout == LAYOUT_COLVECS) ? (int64_t)(n0 + r) * a.ldx + c : (int64_t)c * a.ldx + n0 + r;
          v[u] = X[ok ? addr : 0];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int idx = base + u * kThreads + tid;
          int r, c;
          if (a.layout == LAYOUT_COLVECS) { c = idx % DPc; r = idx / DPc; }
          else                            { r = idx % Cfg::RB; c = idx / Cfg::RB; }
          if (idx < Cfg::RB * DPc) Xs[r * Cfg::LDX + c] = (r < nt && c < D) ? v[u] : T(0);
        }
      }
    }
    __syncthreads();
    // TPR threads per input row (all 256 threads busy), partial sums combined by a shuffle inside the 2- or 4-lane group
    constexpr int TPR = kThreads / Cfg::RB;
    const int prow = tid / TPR, ppart = tid % TPR;
    T m = T(0);
    if (a.mean) {
      const T* xr = Xs + prow * Cfg::LDX;
      T m0 = T(0), m1 = T(0);
      for (int c = 2 * ppart; c + 1 < DPc; c += 2 * TPR) { m0 += xr[c] * mwl[c]; m1 += xr[c + 1] * mwl[c + 1]; }
      m = m0 + m1;
#pragma unroll
      for (int o = 1; o < TPR; o <<= 1) m += __shfl_xor(m, o, 64);
    }
    if (use_factor) {
      __syncthreads();  // mean reads rows across the waves' tiles before the sweep rewrites them
      trsm_sweep<T>(Xs, P, Linv, nchunks, lane, wave);
    }
    {
      T v = T(0);
      if (a.var) {
        const T* xr = Xs + prow * Cfg::LDX;
        T v0 = T(0), v1 = T(0);
        if (use_factor) {
          for (int c = 2 * ppart; c + 1 < DPc; c += 2 * TPR) { v0 += xr[c] * xr[c]; v1 += xr[c + 1] * xr[c + 1]; }
        } else {  // diagonal precision: var_n = sum_d x_dn^2 / d_d
          for (int c = 2 * ppart; c + 1 < DPc; c += 2 * TPR) { v0 += xr[c] * xr[c] * dinv[c]; v1 += xr[c + 1] * xr[c + 1] * dinv[c + 1]; }
        }
        v = v0 + v1;
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) v += __shfl_xor(v, o, 64);
      }
      if (ppart == 0 && prow < nt) {
        if (a.mean) a.mean[(int64_t)reg * a.stridemean + n0 + prow] = m;
        if (a.var) a.var[(int64_t)reg * a.stridevar + n0 + prow] = v + ((a.noise_kind == NOISE_DIAGONAL) ? s[n0 + prow] : s[0]);
      }
    }
  }
}

// ---- weight draws for D <= 128 with a factor: W[:, s] = mw + U^-1 Z[:, s]  (:51, sampling_functions.jl:29,35,44) ----------
// U^-1 z = (z' L^-1)' with L = U': the draws are the ROWS of an LDS block and one backward MFMA sweep (trsm_sweep_back) solves
// a whole tile of them -- the per-lane substitution of the first version re-read U from global memory for every multiply
// (266 us for 64 draws at D = 128).
template <typename T>
__global__ __launch_bounds__(kThreads) void sample_weights_mfma_kernel(const T* __restrict__ mw, const T* __restrict__ U, int64_t ldu,
                                                                       const T* __restrict__ Z, int64_t ldz, T* __restrict__ W,
                                                                       int64_t ldw, int D, int64_t S) {
  using Cfg = TrsmCfg<T>;
  constexpr int VEC = Mfma<T>::VEC;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* const P = reinterpret_cast<T*>(smem);
  T* const Xs = reinterpret_cast<T*>(smem + Cfg::OFF_X);
  T* const dinv = reinterpret_cast<T*>(smem + Cfg::OFF_DI);
  T* const Linv = reinterpret_cast<T*>(smem + Cfg::OFF_LI);
  T* const mwl = reinterpret_cast<T*>(smem + Cfg::LDS_BYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int nchunks = (D + 15) >> 4, DPc = nchunks * 16;
  const bool uvec = D == kPB && (ldu % VEC) == 0 && ((uintptr_t)U % 16) == 0;
  if (uvec) load_upper_block_to_packed(P, U, ldu, tid);
#pragma unroll 1
  for (int base = 0; !uvec && base < DPc * DPc; base += kThreads * 8) {
    T v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * kThreads + tid;
      const int r = idx / DPc, c = idx % DPc;
      const bool in = idx < DPc * DPc && c <= r && r < D;
      v[u] = U[in ? (int64_t)r * ldu + c : 0];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * kThreads + tid;
      const int r = idx / DPc, c = idx % DPc;
      if (idx < DPc * DPc && c <= r) P[pidx(r, c)] = (r < D) ? v[u] : (r == c ? T(1) : T(0));
    }
  }
  if (tid < kPB) mwl[tid] = (tid < D) ? mw[tid] : T(0);
  __syncthreads();
  if (tid < DPc) dinv[tid] = T(1) / P[pidx(tid, tid)];
  __syncthreads();
  trsm_prepare<T>(P, dinv, Linv, nchunks, tid);
  const int64_t ntiles = (S + Cfg::RB - 1) / Cfg::RB;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t s0 = tile * Cfg::RB;
    const int nt = (int)min((int64_t)Cfg::RB, S - s0);
    __syncthreads();
    for (int idx = tid; idx < Cfg::RB * DPc; idx += kThreads) {
      const int c = idx % DPc, r = idx / DPc;  // draw r, coordinate c: contiguous along c in Z
      Xs[r * Cfg::LDX + c] = (r < nt && c < D) ? Z[(s0 + r) * ldz + c] : T(0);
    }
    __syncthreads();
    trsm_sweep_back<T>(Xs, P, Linv, nchunks, lane, wave);
    for (int idx = tid; idx < Cfg::RB * DPc; idx += kThreads) {
      const int c = idx % DPc, r = idx / DPc;
      if (r < nt && c < D) W[(s0 + r) * ldw + c] = mwl[c] + Xs[r * Cfg::LDX + c];
    }
  }
}

// ---- gradient of the log marginal likelihood for D <= 128 --------------------------------------
// The reverse-mode rule of logpdf(fx, y) (reference :55-58), which AD of the reference produces and a drop-in behind a
// ccall has to supply itself.  With S = diag(1/s), A = Lw + X S X' = L L', mw' the posterior mean, r = y - X'mw':
//     dL/dy = -S r,   dL/dmw = X S r,   dL/dX = (mw' r' - A^-1 X) S,   dL/ds_n = -(s_n - r_n^2 - |L^-1 x_n|^2) / (2 s_n^2),
//     dL/dLw = -(m m' + A^-1 - Lw^-1) / 2   (assembled by the caller from A^-1 and m = mw' - mw; D x D host work).
// Same tile loop as marginals_mfma_kernel: the inputs of a tile are the ROWS of an LDS block; sweep 1 (X L^-T) gives the
// quadratic forms, sweep 2 (X L^-1 on top of it) gives A^-1 x_n for all of them -- two MFMA sweeps per tile, no extra pass
// over X.  A^-1 itself falls out of the same two sweeps applied to the rows of the identity (pseudo tiles).
// dmw is reduced per workgroup in a fixed order and summed by grad_reduce_kernel (deterministic).
template <typename T>
struct GradArgs {
  const T* X; int64_t ldx, strideX;
  const T* y; int64_t stridey;
  const T* s; int64_t strides;
  const T* mwp; int64_t stridemwp;       // posterior mean
  const T* U; int64_t ldu, strideU;      // upper factor T = chol(A).U
  T* dX; int64_t lddx, stridedX;
  T* dy; int64_t stridedy;
  T* ds; int64_t strideds;
  double* dmw_part;                      // [B][gridDim.x][128]
  T* Ainv; int64_t ldai, strideAi;
  const int32_t* info;
  int layout, noise_kind;
  int D, N, B;
  int reg0;  // first regressor of this launch
};

template <typename T>
__global__ __launch_bounds__(kThreads) void logpdf_grad_kernel(GradArgs<T> a) {
  using Cfg = TrsmCfg<T>;
  constexpr int VEC = Mfma<T>::VEC;
  typedef T vecT __attribute__((ext_vector_type(Mfma<T>::VEC)));
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* const P = reinterpret_cast<T*>(smem);
  T* const Xs = reinterpret_cast<T*>(smem + Cfg::OFF_X);
  T* const dinv = reinterpret_cast<T*>(smem + Cfg::OFF_DI);
  T* const Linv = reinterpret_cast<T*>(smem + Cfg::OFF_LI);
  T* const mwl = reinterpret_cast<T*>(smem + Cfg::LDS_BYTES);  // [128] posterior mean
  T* const wrv = mwl + kPB;                                     // [RB] w_n r_n
  T* const rvv = wrv + Cfg::RB;                                 // [RB] r_n
  T* const wvv = rvv + Cfg::RB;                                 // [RB] w_n
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int D = a.D, N = a.N;
  const int reg = a.reg0 + blockIdx.y;  // grid.y is limited to 65535: the host launches in chunks
  const int nchunks = (D + 15) >> 4, DPc = nchunks * 16;
  double dmw_acc = 0.0;  // thread c < 128: sum over this workgroup's tiles of x_cn w_n r_n
  const bool ok = !(a.info && a.info[reg] != 0);
  if (ok) {
    const T* X = a.X + (int64_t)reg * a.strideX;
    const T* U = a.U + (int64_t)reg * a.strideU;
    const T* s = a.s + (int64_t)reg * a.strides;
    const T* y = a.y + (int64_t)reg * a.stridey;
    const T* mwp = a.mwp + (int64_t)reg * a.stridemwp;
    const int ntiles = (N + Cfg::RB - 1) / Cfg::RB;
    const int npseudo = a.Ainv ? (DPc + Cfg::RB - 1) / Cfg::RB : 0;
    // vector path (ColVecs, D = 128, 16-byte aligned columns): the next tile's inputs are in flight during this tile's sweeps
    const bool vec = a.layout == LAYOUT_COLVECS && D == kPB && (a.ldx % VEC) == 0 && ((uintptr_t)X % 16) == 0;
    const bool vecd = vec && a.dX && (a.lddx % VEC) == 0 && ((uintptr_t)(a.dX + (int64_t)reg * a.stridedX) % 16) == 0;
    constexpr int VPR = kPB / VEC;
    constexpr int NVT = Cfg::RB * kPB / (VEC * kThreads);
    vecT pre[NVT];
    auto prefetch = [&](int tile) {
      const int n0p = tile * Cfg::RB;
#pragma unroll
      for (int u = 0; u < NVT; ++u) {
        const int vi = u * kThreads + tid;
        const int r = vi / VPR, c0 = (vi % VPR) * VEC;
        pre[u] = (n0p + r < N) ? *reinterpret_cast<const vecT*>(X + (int64_t)(n0p + r) * a.ldx + c0) : vecT(T(0));
      }
    };
    if (vec && (int)blockIdx.x < ntiles) prefetch(blockIdx.x);
    // ---- once per workgroup: L = U' packed (padding: unit diagonal), reciprocal pivots, inverse blocks, mw'
    const bool uvec = D == kPB && (a.ldu % VEC) == 0 && ((uintptr_t)U % 16) == 0;
    if (uvec) load_upper_block_to_packed(P, U, a.ldu, tid);
#pragma unroll 1
    for (int base = 0; !uvec && base < DPc * DPc; base += kThreads * 8) {
      T v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u * kThreads + tid;
        const int r = idx / DPc, c = idx % DPc;
        const bool in = idx < DPc * DPc && c <= r && r < D;
        v[u] = U[in ? (int64_t)r * a.ldu + c : 0];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u * kThreads + tid;
        const int r = idx / DPc, c = idx % DPc;
        if (idx < DPc * DPc && c <= r) P[pidx(r, c)] = (r < D) ? v[u] : (r == c ? T(1) : T(0));
      }
    }
    if (tid < kPB) mwl[tid] = (tid < D) ? mwp[tid] : T(0);
    __syncthreads();
    if (tid < DPc) dinv[tid] = T(1) / P[pidx(tid, tid)];
    __syncthreads();
    trsm_prepare<T>(P, dinv, Linv, nchunks, tid);

    for (int tile = blockIdx.x; tile < ntiles + npseudo; tile += gridDim.x) {
      const bool pseudo = tile >= ntiles;
      const int n0 = pseudo ? (tile - ntiles) * Cfg::RB : tile * Cfg::RB;  // pseudo: first identity row
      const int nt = pseudo ? min(Cfg::RB, DPc - n0) : min(Cfg::RB, N - n0);
      __syncthreads();  // the previous tile's readers of Xs are done
      if (vec && !pseudo) {
#pragma unroll
        for (int u = 0; u < NVT; ++u) {
          const int vi = u * kThreads + tid;
          const int r = vi / VPR, c0 = (vi % VPR) * VEC;
#pragma unroll
          for (int e = 0; e < VEC; ++e) Xs[r * Cfg::LDX + c0 + e] = pre[u][e];
        }
        if (tile + (int)gridDim.x < ntiles) prefetch(tile + gridDim.x);
      } else
#pragma unroll 1
      for (int base = 0; base < Cfg::RB * DPc; base += kThreads * 8) {
        T v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int idx = base + u * kThreads + tid;
          int r, c;
          if (a.layout == LAYOUT_COLVECS) { c = idx % DPc; r = idx / DPc; }
          else                            { r = idx % Cfg::RB; c = idx / Cfg::RB; }
          const bool in = !pseudo && idx < Cfg::RB * DPc && r < nt && c < D;
          const int64_t addr = (a.layout == LAYOUT_COLVECS) ? (int64_t)(n0 + r) * a.ldx + c : (int64_t)c * a.ldx + n0 + r;
          v[u] = X[in ? addr : 0];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int idx = base + u * kThreads + tid;
          int r, c;
          if (a.layout == LAYOUT_COLVECS) { c = idx % DPc; r = idx / DPc; }
          else                            { r = idx % Cfg::RB; c = idx / Cfg::RB; }
          if (idx < Cfg::RB * DPc) {
            T val = (r < nt && c < D) ? v[u] : T(0);
            if (pseudo) val = (r < nt && n0 + r == c) ? T(1) : T(0);
            Xs[r * Cfg::LDX + c] = val;
          }
        }
      }
      __syncthreads();
      // TPR threads per input row; the thread with ppart == 0 owns the row's scalars
      constexpr int TPR = kThreads / Cfg::RB;
      const int prow = tid / TPR, ppart = tid % TPR;
      const bool owner = ppart == 0 && prow < nt;
      T rr = T(0), w = T(0), sv = T(1);
      if (!pseudo) {
        {
          const T* xr = Xs + prow * Cfg::LDX;
          T m0 = T(0), m1 = T(0);
          for (int c = 2 * ppart; c + 1 < DPc; c += 2 * TPR) { m0 += xr[c] * mwl[c]; m1 += xr[c + 1] * mwl[c + 1]; }
          T mu = m0 + m1;
#pragma unroll
          for (int o = 1; o < TPR; o <<= 1) mu += __shfl_xor(mu, o, 64);
          if (owner) {
            sv = (a.noise_kind == NOISE_DIAGONAL) ? s[n0 + prow] : s[0];
            w = T(1) / sv;
            rr = y[n0 + prow] - mu;  // posterior residual
          }
          if (ppart == 0) {
            wrv[prow] = w * rr;
            rvv[prow] = rr;
            wvv[prow] = w;
          }
        }
        __syncthreads();
        if (tid < kPB) {  // dmw_c += sum_n x_cn w_n r_n over the tile (fixed order)
          double acc = 0.0;
          for (int r = 0; r < Cfg::RB; ++r) acc += (double)Xs[r * Cfg::LDX + tid] * (double)wrv[r];
          dmw_acc += acc;
        }
      }
      __syncthreads();
      trsm_sweep<T>(Xs, P, Linv, nchunks, lane, wave);       // rows z_n' = x_n' L^-T
      T vq = T(0);
      if (!pseudo) {
        const T* xr = Xs + prow * Cfg::LDX;
        T v0 = T(0), v1 = T(0);
        for (int c = 2 * ppart; c + 1 < DPc; c += 2 * TPR) { v0 += xr[c] * xr[c]; v1 += xr[c + 1] * xr[c + 1]; }
        vq = v0 + v1;  // x_n' A^-1 x_n
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) vq += __shfl_xor(vq, o, 64);
      }
      __syncthreads();
      trsm_sweep_back<T>(Xs, P, Linv, nchunks, lane, wave);  // rows g_n' = x_n' A^-1
      if (pseudo) {
        T* Ai = a.Ainv + (int64_t)reg * a.strideAi;
        for (int idx = tid; idx < Cfg::RB * DPc; idx += kThreads) {
          const int r = idx % Cfg::RB, c = idx / Cfg::RB;  // consecutive threads -> consecutive rows of a column
          if (r < nt && n0 + r < D && c < D) Ai[(int64_t)c * a.ldai + n0 + r] = Xs[r * Cfg::LDX + c];
        }
      } else {
        if (owner) {
          if (a.dy) a.dy[(int64_t)reg * a.stridedy + n0 + prow] = -w * rr;
          if (a.ds) a.ds[(int64_t)reg * a.strideds + n0 + prow] = -(sv - rr * rr - vq) / (T(2) * sv * sv);
        }
        if (vecd) {
          T* dXr = a.dX + (int64_t)reg * a.stridedX;
#pragma unroll 4
          for (int u = 0; u < NVT; ++u) {
            const int vi = u * kThreads + tid;
            const int r = vi / VPR, c0 = (vi % VPR) * VEC;
            if (r < nt) {
              vecT o;
#pragma unroll
              for (int e = 0; e < VEC; ++e) o[e] = wvv[r] * (rvv[r] * mwl[c0 + e] - Xs[r * Cfg::LDX + c0 + e]);
              *reinterpret_cast<vecT*>(dXr + (int64_t)(n0 + r) * a.lddx + c0) = o;
            }
          }
        } else if (a.dX) {
          T* dXr = a.dX + (int64_t)reg * a.stridedX;
          for (int idx = tid; idx < Cfg::RB * DPc; idx += kThreads) {
            int r, c;
            if (a.layout == LAYOUT_COLVECS) { c = idx % DPc; r = idx / DPc; }
            else                            { r = idx % Cfg::RB; c = idx / Cfg::RB; }
            if (r < nt && c < D) {
              const T val = wvv[r] * (rvv[r] * mwl[c] - Xs[r * Cfg::LDX + c]);
              const int64_t addr = (a.layout == LAYOUT_COLVECS) ? (int64_t)(n0 + r) * a.lddx + c : (int64_t)c * a.lddx + n0 + r;
              dXr[addr] = val;
            }
          }
        }
      }
    }
  }
  if (a.dmw_part && tid < kPB) a.dmw_part[((int64_t)reg * gridDim.x + blockIdx.x) * kPB + tid] = dmw_acc;
}

// dmw[reg][c] = sum over the workgroup partials, fixed order
template <typename T>
__global__ __launch_bounds__(kPB) void grad_reduce_kernel(const double* part, int nparts, T* dmw, int64_t stridedmw, int D) {
  const int reg = blockIdx.x, c = threadIdx.x;
  if (c >= D) return;
  double acc = 0.0;
  for (int g = 0; g < nparts; ++g) acc += part[((int64_t)reg * nparts + g) * kPB + c];
  dmw[(int64_t)reg * stridedmw + c] = (T)acc;
}

// ---- gradient for D > 128: the same two sweeps through the panel machinery of the tall matrix ---------------------------
// Tall matrix  [ F ; X' ; I ]  (ld = rows): F = the factor block with BOTH triangles filled (lower: L, upper: T = L'), the
// inputs as rows, and (for A^-1) the rows of the identity.  Forward panels (trsm_block_kernel + MFMA trailing updates, as
// in the marginal stream) turn every row x' into x'L^-T; backward panels (trsm_back_block_kernel + the same trailing
// update kernel reading the UPPER triangle of F as its second operand) turn that into x'L^-T L^-1 = x'A^-1.

// top block: lower triangle L = U', upper triangle U, unit padding
template <typename T>
__global__ __launch_bounds__(kThreads) void factor_sym_fill_kernel(const T* U, int64_t ldu, int D, int DP, T* Ybar, int64_t ldy, int64_t grp_U = 0,
                                                                   int64_t grp_ws = 0) {
  __shared__ T tile[32][33];
  if (const int64_t g = blockIdx.z) { U += g * grp_U; Ybar = ws_shift(Ybar, g * grp_ws); }  // regressor of a group
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) {
    const int ur = by + tx, uc = bx + k;
    tile[k][tx] = (ur < D && uc < D && ur <= uc) ? U[(int64_t)uc * ldu + ur] : T(0);
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int row = bx + tx, col = by + k;  // L[row, col] = U[col, row] = tile[tx][k]
    if (row < DP && col < DP && row >= col) {
      T v = tile[tx][k];
      if (row >= D || col >= D) v = (row == col) ? T(1) : T(0);
      Ybar[(int64_t)col * ldy + row] = v;                  // lower (and diagonal)
      if (row > col) Ybar[(int64_t)row * ldy + col] = v;   // mirrored: element (col, row) of the upper triangle
    }
  }
}

// rows [row0, row0 + DP) of the tall matrix := identity
template <typename T>
__global__ __launch_bounds__(kThreads) void identity_rows_kernel(T* Ybar, int64_t ldy, int row0, int DP, int64_t grp_ws = 0) {
  Ybar = ws_shift(Ybar, (int64_t)blockIdx.y * grp_ws);
  for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < (int64_t)DP * DP; e += (int64_t)gridDim.x * kThreads) {
    const int c = (int)(e / DP), r = (int)(e % DP);
    Ybar[(int64_t)c * ldy + row0 + r] = (r == c) ? T(1) : T(0);
  }
}

// X <- X L_pp^-1 for one block of RB rows (the backward panel step)
template <typename T>
__global__ __launch_bounds__(kThreads) void trsm_back_block_kernel(T* Abar, int64_t lda, int p, int row_begin, int nrows_total,
                                                                   const int32_t* info, int64_t grp_ws = 0) {
  if (const int64_t g = blockIdx.y) { Abar = ws_shift(Abar, g * grp_ws); info += g; }  // regressor of a group
  using Cfg = TrsmCfg<T>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* const P = reinterpret_cast<T*>(smem);
  T* const Xs = reinterpret_cast<T*>(smem + Cfg::OFF_X);
  T* const dinv = reinterpret_cast<T*>(smem + Cfg::OFF_DI);
  T* const Linv = reinterpret_cast<T*>(smem + Cfg::OFF_LI);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int r0 = row_begin + blockIdx.x * Cfg::RB;
  const int nr = min(Cfg::RB, nrows_total - r0);
  const T* Lpp = Abar + (int64_t)p * kPB * lda + (int64_t)p * kPB;
  T* Xg = Abar + (int64_t)p * kPB * lda + r0;
  {
    BlockVec<T, kPB> lb;
    BlockVec<T, Cfg::RB> xb;
    lb.load(Lpp, lda, tid);
    xb.load(Xg, lda, tid);
    if (*info != 0) return;
    lb.to_packed_lower(P, tid);
    xb.to_rows(Xs, Cfg::LDX, nr, tid);
  }
  __syncthreads();
  if (tid < kPB) dinv[tid] = T(1) / P[pidx(tid, tid)];
  __syncthreads();
  trsm_prepare<T>(P, dinv, Linv, 8, tid);
  trsm_sweep_back<T>(Xs, P, Linv, 8, lane, wave);
  {
    using BV = BlockVec<T, Cfg::RB>;
#pragma unroll 4
    for (int u = 0; u < BV::NV; ++u) {
      const int vi = u * kThreads + tid;
      const int c = vi / BV::VPC, rr = (vi % BV::VPC) * BV::VEC;
      if (rr < nr) {
        typename BV::vecT o;
#pragma unroll
        for (int e = 0; e < BV::VEC; ++e) o[e] = Xs[(rr + e) * Cfg::LDX + c];
        *reinterpret_cast<typename BV::vecT*>(Xg + (int64_t)c * lda + rr) = o;
      }
    }
  }
}

// per observation: r_n = y_n - mean_n, w_n = 1/s_n; wr_n = w_n r_n; dy_n = -w_n r_n; ds_n = -(s_n - r_n^2 - v_n)/(2 s_n^2)
// (var_n = v_n + s_n comes from the forward panels' fused row sums of squares)
struct GradObsGroup {  // blockIdx.y = regressor of a group: element strides of the caller's arrays, byte stride of mean / var / rvec / wvec
  int64_t y, s, dy, ds, ws;
};
template <typename T>
__global__ __launch_bounds__(kThreads) void grad_obs_kernel(const T* y, const T* mean, const T* var, const T* s, int noise_kind,
                                                            int N, T* rvec, T* wvec, T* dy, T* ds, GradObsGroup grp = GradObsGroup()) {
  const int n = blockIdx.x * kThreads + threadIdx.x;
  if (n >= N) return;
  if (const int64_t g = blockIdx.y) {
    y += g * grp.y; s += g * grp.s;
    if (dy) dy += g * grp.dy;
    if (ds) ds += g * grp.ds;
    mean = ws_shift(mean, g * grp.ws); var = ws_shift(var, g * grp.ws); rvec = ws_shift(rvec, g * grp.ws); wvec = ws_shift(wvec, g * grp.ws);
  }
  const T sv = (noise_kind == NOISE_DIAGONAL) ? s[n] : s[0];
  const T w = T(1) / sv, rr = y[n] - mean[n];
  rvec[n] = rr;
  wvec[n] = w;
  if (dy) dy[n] = -w * rr;
  if (ds) ds[n] = -(sv - rr * rr - (var[n] - sv)) / (T(2) * sv * sv);
}

// dX = (mw' r' - A^-1 X) S in the caller's layout from the rows g_n' = x_n'A^-1 of the tall matrix; the same 64 x 64
// tiles also give the partial sums of  dmw_d = sum_n x_dn w_n r_n  (fixed order inside a tile; tiles summed by
// grad_reduce_large_kernel)
template <typename T>
struct GradOutArgs {
  const T* Ybar; int64_t ldy; int row0;   // rows g_n
  const T* X; int64_t ldx; int layout;    // original inputs (for dmw)
  const T* rvec; const T* wvec; const T* mwp;
  T* dX; int64_t lddx;
  double* dmw_part;                        // [gridDim.x][DP]
  int D, DP, N;
  int64_t grp_X, grp_mwp, grp_dX, grp_ws;  // blockIdx.y = regressor of a group: element strides of X / mwp / dX, byte stride of the rest
};
template <typename T>
__global__ __launch_bounds__(kThreads) void grad_out_large_kernel(GradOutArgs<T> a) {
  __shared__ T tile[64][65];
  __shared__ T wr[64], rr[64], ww[64];
  if (const int64_t g = blockIdx.y) {
    a.X += g * a.grp_X; a.mwp += g * a.grp_mwp;
    if (a.dX) a.dX += g * a.grp_dX;
    a.Ybar = ws_shift(a.Ybar, g * a.grp_ws); a.rvec = ws_shift(a.rvec, g * a.grp_ws); a.wvec = ws_shift(a.wvec, g * a.grp_ws);
    a.dmw_part = ws_shift(a.dmw_part, g * a.grp_ws);
  }
  const int tid = threadIdx.x;
  const int n0 = blockIdx.x * 64;
  if (tid < 64) {
    const int n = n0 + tid;
    const T r = n < a.N ? a.rvec[n] : T(0), w = n < a.N ? a.wvec[n] : T(0);
    rr[tid] = r; ww[tid] = w; wr[tid] = w * r;
  }
  for (int d0 = 0; d0 < a.DP; d0 += 64) {
    __syncthreads();
    // g tile: rows n (contiguous in the tall matrix), columns d
    for (int e = tid; e < 64 * 64; e += kThreads) {
      const int nn = e & 63, dd = e >> 6;
      tile[dd][nn] = a.Ybar[(int64_t)(d0 + dd) * a.ldy + a.row0 + n0 + nn];
    }
    __syncthreads();
    if (a.dX) {
      for (int e = tid; e < 64 * 64; e += kThreads) {
        int dd, nn;
        if (a.layout == LAYOUT_COLVECS) { dd = e & 63; nn = e >> 6; }
        else                            { nn = e & 63; dd = e >> 6; }
        const int d = d0 + dd, n = n0 + nn;
        if (d < a.D && n < a.N) {
          const T val = ww[nn] * (rr[nn] * a.mwp[d] - tile[dd][nn]);
          a.dX[(a.layout == LAYOUT_COLVECS) ? (int64_t)n * a.lddx + d : (int64_t)d * a.lddx + n] = val;
        }
      }
    }
    __syncthreads();
    if (a.dmw_part) {  // x tile, then dmw partial for these 64 d's
      for (int e = tid; e < 64 * 64; e += kThreads) {
        int dd, nn;
        if (a.layout == LAYOUT_COLVECS) { dd = e & 63; nn = e >> 6; }
        else                            { nn = e & 63; dd = e >> 6; }
        const int d = d0 + dd, n = n0 + nn;
        T v = T(0);
        if (d < a.D && n < a.N) v = (a.layout == LAYOUT_COLVECS) ? a.X[(int64_t)n * a.ldx + d] : a.X[(int64_t)d * a.ldx + n];
        tile[dd][nn] = v;
      }
      __syncthreads();
      if (tid < 64) {
        double acc = 0.0;
        for (int nn = 0; nn < 64; ++nn) acc += (double)tile[tid][nn] * (double)wr[nn];
        a.dmw_part[(int64_t)blockIdx.x * a.DP + d0 + tid] = acc;
      }
    }
  }
}
template <typename T>
__global__ __launch_bounds__(kThreads) void grad_reduce_large_kernel(const double* part, int nparts, int DP, int D, T* dmw, int64_t grp_ws = 0,
                                                                     int64_t grp_dmw = 0) {
  const int d = blockIdx.x * kThreads + threadIdx.x;
  if (d >= D) return;
  part = ws_shift(part, (int64_t)blockIdx.y * grp_ws); dmw += (int64_t)blockIdx.y * grp_dmw;  // regressor of a group
  double acc = 0.0;
  for (int g = 0; g < nparts; ++g) acc += part[(int64_t)g * DP + d];
  dmw[d] = (T)acc;
}
// A^-1 from the identity rows of the tall matrix
template <typename T>
__global__ __launch_bounds__(kThreads) void ainv_copy_kernel(const T* Ybar, int64_t ldy, int row0, int D, T* Ainv, int64_t ldai, int64_t grp_ws = 0,
                                                             int64_t grp_Ai = 0) {
  Ybar = ws_shift(Ybar, (int64_t)blockIdx.y * grp_ws); Ainv += (int64_t)blockIdx.y * grp_Ai;  // regressor of a group
  for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < (int64_t)D * D; e += (int64_t)gridDim.x * kThreads) {
    const int c = (int)(e / D), r = (int)(e % D);
    Ainv[(int64_t)c * ldai + r] = Ybar[(int64_t)c * ldy + row0 + r];
  }
}

// ---- shared-X multi-output evidence: logpdf(fx, Y::Matrix) ---------------------------------------
// All S columns of Y share X, hence A = Lw + X S X' and its factor: ONE Gram + Cholesky (the ordinary posterior call on
// column 0), then per column only  q_s = delta_s' S delta_s  and  |L^-1 b_s|^2  with  b_s = X S delta_s:
//     logpdf_s = logpdf_0 + (q_0 - |u_0|^2)/2 - (q_s - |u_s|^2)/2.
// B = X (S Delta) is one D x N x S GEMM (gram_tile_kernel with the residual matrix as its first operand and X as its
// second), and the solves are the tall-matrix panels of the large path applied to the rows b_s'.

// R = S (Y - mu 1') in the layout of X (ColVecs: R' stored SP x N, s contiguous; RowVecs: N x SP), q partials per 64 rows
template <typename T>
struct MultiPrepArgs {
  const T* Y; int64_t ldY;      // N x S column-major
  const T* mu; const T* s; int noise_kind;
  T* R; int64_t ldr; int layout;
  double* qpart;                 // [gridDim.x][SP]
  int N, S, SP;
};
template <typename T>
__global__ __launch_bounds__(kThreads) void multi_prep_kernel(MultiPrepArgs<T> a) {
  __shared__ T tile[64][65];
  __shared__ double qt[4][64];
  const int tid = threadIdx.x;
  const int n0 = blockIdx.x * 64;
  const int tn = tid & 63, tq = tid >> 6;
  T w = T(0), m = T(0);
  if (n0 + tn < a.N) {
    w = T(1) / ((a.noise_kind == NOISE_DIAGONAL) ? a.s[n0 + tn] : a.s[0]);
    m = a.mu[n0 + tn];
  }
  for (int s0 = 0; s0 < a.SP; s0 += 64) {
    __syncthreads();
    for (int k = tq; k < 64; k += 4) {  // column s0 + k, rows n0 + tn: coalesced along n
      const int sidx = s0 + k, n = n0 + tn;
      T r = T(0);
      double q = 0.0;
      if (sidx < a.S && n < a.N) {
        const T d = a.Y[(int64_t)sidx * a.ldY + n] - m;
        r = w * d;
        q = (double)d * (double)r;
      }
      tile[k][tn] = r;
      q = wave_allreduce(q);            // the 64 rows of this tile, fixed butterfly
      if (tn == 0) qt[0][k] = q;
    }
    __syncthreads();
    if (tid < 64) a.qpart[(int64_t)blockIdx.x * a.SP + s0 + tid] = qt[0][tid];
    for (int e = tid; e < 64 * 64; e += kThreads) {
      if (a.layout == LAYOUT_COLVECS) {  // R'[s + n * ldr]
        const int ss = e & 63, nn = e >> 6;
        if (n0 + nn < a.N) a.R[(int64_t)(n0 + nn) * a.ldr + s0 + ss] = tile[ss][nn];
      } else {                           // R[n + s * ldr]
        const int nn = e & 63, ss = e >> 6;
        if (n0 + nn < a.N) a.R[(int64_t)(s0 + ss) * a.ldr + n0 + nn] = tile[ss][nn];
      }
    }
  }
}

// split-K partial tiles of B' = R'X' (rows s, columns d) -> rows [row0, row0 + SP) of the tall matrix, fixed order
template <typename T>
__global__ __launch_bounds__(kThreads) void multi_reduce_kernel(const T* Gpart, int nsplit, int ntiles, int ntile_rows, T* Ybar,
                                                                int64_t ldy, int row0) {
  const int t = blockIdx.x;
  const int I = t % ntile_rows, J = t / ntile_rows;  // tri == 3 enumeration of gram_tile_kernel
  constexpr int kChunk = kPB * kPB / 16;
  const int e_begin = blockIdx.y * kChunk, e_end = e_begin + kChunk;
  for (int e = e_begin + threadIdx.x; e < e_end; e += kThreads) {
    const int rl = e % kPB, cl = e / kPB;
    T sum = T(0);
    int sp = 0;
    for (; sp + 8 <= nsplit; sp += 8) {  // eight partials in flight, added in split order (one load and a dependent add per split
      T v[8];                            // was a latency chain of nsplit global loads)
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = Gpart[((int64_t)(sp + u) * ntiles + t) * (kPB * kPB) + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) sum += v[u];
    }
    for (; sp < nsplit; ++sp) sum += Gpart[((int64_t)sp * ntiles + t) * (kPB * kPB) + e];
    Ybar[(int64_t)(J * kPB + cl) * ldy + row0 + I * kPB + rl] = sum;
  }
}

// logpdf_s from column 0's value and the per-column scalars; optional posterior means from the rows m_s
template <typename T>
__global__ __launch_bounds__(kThreads) void multi_finish_kernel(const double* lp0, const double* qpart, int nqparts, int SP,
                                                                const T* uu, int S, double* logpdf) {
  // one workgroup per column s: q_s and q_0 are sums of nqparts partials each (a single thread walking them was 1024 dependent
  // global loads: 0.75 ms of a 3 ms call) -- thread t takes the partials t, t + 256, ..., then a fixed-order tree: deterministic
  __shared__ double red[2][kThreads];
  const int sidx = blockIdx.x, tid = threadIdx.x;
  if (sidx >= S) return;
  double q = 0.0, q0 = 0.0;
  for (int g = tid; g < nqparts; g += kThreads) {
    q += qpart[(int64_t)g * SP + sidx];
    q0 += qpart[(int64_t)g * SP];
  }
  red[0][tid] = q;
  red[1][tid] = q0;
  __syncthreads();
  for (int m = kThreads / 2; m >= 1; m >>= 1) {
    if (tid < m) {
      red[0][tid] += red[0][tid + m];
      red[1][tid] += red[1][tid + m];
    }
    __syncthreads();
  }
  if (tid == 0) logpdf[sidx] = *lp0 + 0.5 * (red[1][0] - (double)uu[0]) - 0.5 * (red[0][0] - (double)uu[sidx]);
}
// The same from the rows the blocked factorisation carried along (multi-output call on the planes route): u_s' = row DP + s of the
// factored Abar, q_s from the planes pass's partial sums (isotropic noise: still without the 1 / s).  One workgroup per column.
template <typename T>
__global__ __launch_bounds__(kThreads) void multi_rows_finish_kernel(const double* lp0, const double* qsp, int nq, const T* s_iso, const T* Abar,
                                                                     int64_t lda, int DP, int D, int S, double* logpdf) {
  __shared__ double red[4][kThreads];
  const int sidx = blockIdx.x, tid = threadIdx.x;
  if (sidx >= S) return;
  double q = 0.0, q0 = 0.0, uu = 0.0, uu0 = 0.0;
  for (int g = tid; g < nq; g += kThreads) {
    q += qsp[(int64_t)g * kPB + sidx];
    q0 += qsp[(int64_t)g * kPB];
  }
  for (int d = tid; d < D; d += kThreads) {
    const double u = (double)Abar[(int64_t)d * lda + DP + sidx], u0 = (double)Abar[(int64_t)d * lda + DP];
    uu += u * u;
    uu0 += u0 * u0;
  }
  red[0][tid] = q; red[1][tid] = q0; red[2][tid] = uu; red[3][tid] = uu0;
  __syncthreads();
  for (int m = kThreads / 2; m >= 1; m >>= 1) {
    if (tid < m) {
#pragma unroll
      for (int k = 0; k < 4; ++k) red[k][tid] += red[k][tid + m];
    }
    __syncthreads();
  }
  if (tid == 0) {
    const double sc = s_iso ? 1.0 / (double)s_iso[0] : 1.0;
    logpdf[sidx] = sidx == 0 ? *lp0 : *lp0 + 0.5 * (sc * red[1][0] - red[3][0]) - 0.5 * (sc * red[0][0] - red[2][0]);
  }
}
template <typename T>
__global__ __launch_bounds__(kThreads) void multi_means_kernel(const T* Ybar, int64_t ldy, int row0, const T* mw, int D, int S,
                                                               T* mw_post, int64_t ldmp) {
  for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < (int64_t)D * S; e += (int64_t)gridDim.x * kThreads) {
    const int sidx = (int)(e / D), d = (int)(e % D);
    mw_post[(int64_t)sidx * ldmp + d] = mw[d] + Ybar[(int64_t)d * ldy + row0 + sidx];
  }
}

// ---- N-sharded single regressor: sufficient statistics of a column block, summed by the host's collective ----
// stats = the augmented matrix without the prior: lower triangle of G_r = X_r S_r X_r' in rows [0, DP), row DP = b_r';
// scal = {delta' S delta, logdet Sigma_y} of the block.  All of it is additive over column blocks.
template <typename T>
__global__ __launch_bounds__(kThreads) void stats_scalars_kernel(const double* qpart, const double* lpart, int nparts, int noise_kind,
                                                                 const T* s, int N, double* scal) {
  __shared__ double scr[8];
  double q = 0.0, l = 0.0;
  for (int i = threadIdx.x; i < nparts; i += kThreads) { q += qpart[i]; l += lpart[i]; }
  q = block_allreduce(q, scr, threadIdx.x);
  l = block_allreduce(l, scr, threadIdx.x);
  if (threadIdx.x == 0) {
    scal[0] = q;
    scal[1] = (noise_kind == NOISE_DIAGONAL) ? l : (double)N * log((double)s[0]);
  }
}
// summed statistics + prior precision -> the augmented matrix the factorisation expects (padding: unit diagonal, zero rows
// below b'); optional full symmetric copy of A for the caller
template <typename T>
__global__ __launch_bounds__(kThreads) void stats_add_prior_kernel(T* Abar, int64_t lda, int D, int DP, const T* Lw, int64_t ldl,
                                                                   int prior_kind, T* Lw_post, int64_t ldlp) {
  for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < (int64_t)(DP + kPB) * DP; e += (int64_t)gridDim.x * kThreads) {
    const int col = (int)(e / (DP + kPB)), row = (int)(e % (DP + kPB));
    T* p = Abar + (int64_t)col * lda + row;
    if (row < DP) {
      if (row < col) continue;  // upper triangle: unused
      if (row >= D) { *p = (row == col) ? T(1) : T(0); continue; }
      T v = *p;
      if (prior_kind == PRIOR_DENSE) v += Lw[(int64_t)row * ldl + col];  // upper entry (col, row)
      else if (prior_kind == PRIOR_DIAGONAL && row == col) v += Lw[row];
      *p = v;
      if (Lw_post) { Lw_post[(int64_t)col * ldlp + row] = v; Lw_post[(int64_t)row * ldlp + col] = v; }
    } else if (row == DP) {
      if (col >= D) *p = T(0);
    } else {
      *p = T(0);
    }
  }
}

}  // namespace blr
