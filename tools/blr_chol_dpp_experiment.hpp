// EXPERIMENT (round 5) -- not part of the library; built only by tools/chol_bench.hip (profiles/r05_microbench_chol_bench.txt; LOG.md,
// late round 5).  Per 128 x 128 fp64 factorisation: phase_chol 40.3 us; chol128_dpp 38.7; chol128_cw with 4 waves 37.0, with 8 waves
// (one chain wave, six update waves, its SIMD partner idle) 32.1 us -- most of that a higher clock in the microbenchmark: in cycles
// 63 k against 68 k.  chol128_cw<.., 8> was also built INTO fused_i8_kernel (waves 4 - 7 kept alive through the four-wave prior code by
// counting barriers, the block inverses handed to the back substitution): -4.3 k cycles of factorisation and -5.5 k of substitution
// per regressor, and 4.403 -> 4.398 ms per 4096 updates sustained (in-kernel clock 1.812 -> 1.802 GHz): the kernel runs at the power
// limit and gave the cycles back as clock.  Not shipped.
//
// Blocked Cholesky of the packed 128 x 128 lower triangle in LDS, fp64, for the int8 route's kernel (reference:
// `cholesky(Symmetric(...))`, bayesian_linear_regression.jl:86, and the forward substitution of :57 / :68 riding along).
//
// phase_chol (blr_fused_small.hpp) eliminates a 16-column panel one row per lane over the WHOLE panel height: every multiplier of a
// column reaches the other lanes through v_readlane -> SGPR -> v_fma, 16 x (reciprocal + 15 - c broadcasts) in series per panel --
// 317 cycles per column, 40 k of the 68 k cycles of a factorisation (tools/chol_bench), while the matrix pipe idles.  Here the serial
// part is ONE 16 x 16 tile per panel:
//   (1) every wave factors the diagonal tile, one row per lane, each 16-lane DPP row on its own copy, every broadcast a DPP
//       `row_newbcast` operand of the multiply-add itself, and gets the tile's INVERSE from the same pass -- DPP row q carries columns
//       {q, 4 + q, 8 + q, 12 + q} of it, exactly the four B fragments of v_mfma_f64_16x16x4 (tile_factor_invert, blr_panel.hpp: the
//       chain wave of the D > 128 panel kernel);
//   (2) the sub-diagonal tiles are solved on the matrix pipe, L_IJ = A_IJ L_JJ^-T: 4 MFMAs per tile, tiles dealt over the waves;
//       u_J = L_JJ^-1 r_J (the right-hand side's forward substitution) is a 16 x 16 product against the same inverse;
//   (3) after one barrier: r_K -= L_KJ u_J for the rows below, and the trailing update A_IK -= L_IJ L_KJ' of the tiles each wave keeps
//       in its accumulators, as in phase_chol.
// The inverses W_J = L_JJ^-1 also go to LDS for the blocked back substitution (i8_backsolve_blocked), which then no longer forms them.
// Not bit-identical to phase_chol: u_J is a product with the inverse, not a substitution, and L_IJ a product with L_JJ^-T.
#pragma once
#include "blr_fused_small.hpp"
#include "blr_panel.hpp"

namespace blr {

// 64-bit lane exchange across the 16-lane DPP rows (lane ^ 16, lane ^ 32)
__device__ __forceinline__ double xor_lanes(double x, int mask) {
  const int lo = __shfl_xor(__double2loint(x), mask, 64), hi = __shfl_xor(__double2hiint(x), mask, 64);
  return __hiloint2double(hi, lo);
}

// ---- the 16 x 16 tile in fp64, instruction stream in OUR order -----------------------------------------------------------------------------
// tile_factor_invert's generic form leaves the order to hipcc, which keeps a column's dependent chain BEHIND its independent multiply-adds:
// 3.0 k cycles per tile (tools/chol_bench).  The chain of a column is
//   v_fmac_f64_dpp a[C+1] -> (2 wait states) -> v_rsq_f64_dpp of the pivot -> h = (x/2) y0 -> e = 1/2 - h y0 -> p = 1 + 3/2 e -> q = e p
//   -> y = y0 + y0 q -> l = a[C+1] y, -l -> (2 wait states) -> next column
// (one CUBIC step on the hardware's 2^-24 seed: y0 (1 + e + 3/2 e^2) has relative error delta^3, one operation shorter than two Newton
// steps), a dependent fp64 result costing ~10 cycles; the other 14 - C multiply-adds of the column, the inverse's multiply-adds, the
// mask of the rows below and the lane's own 1 / L(r, r) are dealt into those latency slots as volatile single-instruction statements.
template <int K>
__device__ __forceinline__ void vfmac_bc16(double& acc, double b, double m) {
  BLR_VA("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(m), "n"(K));
}
template <int C, int K0, int N>
__device__ __forceinline__ void col_fill64(double (&a)[16], double ln) {
  if constexpr (N > 0 && K0 < 16) {
    vfmac_bc16<K0>(a[K0], a[C], ln);
    col_fill64<C, K0 + 1, N - 1>(a, ln);
  }
}
template <int C>
__device__ __forceinline__ void tile_factor_col_f64h(double (&a)[16], double (&y)[4], double ln, double rs, double& rsr, double c15, int r) {
  // on entry: a[C] = L(r, C) past its wait states, ln = -a[C], rs = 1 / L(C, C)
  double t;
  if constexpr (C < 15) {
    constexpr int n = C + 1;
    vfmac_bc16<n>(a[n], a[C], ln);                                                                        // chain
    BLR_VA("v_mul_f64 %0, %1, %2" : "=v"(t) : "v"(ln), "v"(rs));                                          // (wait state) -l / L(C, C)
    if constexpr (C + 2 < 16) col_fill64<C, C + 2, 1>(a, ln);                                             // (wait state)
    else BLR_VA("s_nop 0");
    double y0, x, hx, h, e, p, q2, yn, lnn;
    BLR_VA("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(x) : "v"(a[n]), "n"(n));         // chain
    BLR_VA("v_rsq_f64 %0, %1" : "=v"(y0) : "v"(x));                                                       // chain
    t = (r > C) ? t : 0.0;                                                                                // rows below C only
    BLR_VA("v_mul_f64 %0, %1, 0.5" : "=v"(hx) : "v"(x));
    col_fill64<C, C + 3, 1>(a, ln);
    BLR_VA("v_mul_f64 %0, %1, %2" : "=v"(h) : "v"(hx), "v"(y0));                                          // chain
    col_fill64<C, C + 4, 1>(a, ln);
    if constexpr (0 <= C) vfmac_bc16<C>(y[0], y[0], t);
    BLR_VA("v_fma_f64 %0, -%1, %2, 0.5" : "=v"(e) : "v"(h), "v"(y0));                                     // chain
    col_fill64<C, C + 5, 1>(a, ln);
    if constexpr (4 <= C) vfmac_bc16<C>(y[1], y[1], t);
    BLR_VA("v_fma_f64 %0, %1, %2, 1.0" : "=v"(p) : "v"(e), "v"(c15));                                     // chain
    col_fill64<C, C + 6, 1>(a, ln);
    if constexpr (8 <= C) vfmac_bc16<C>(y[2], y[2], t);
    BLR_VA("v_mul_f64 %0, %1, %2" : "=v"(q2) : "v"(e), "v"(p));                                           // chain
    col_fill64<C, C + 7, 1>(a, ln);
    if constexpr (12 <= C) vfmac_bc16<C>(y[3], y[3], t);
    BLR_VA("v_fma_f64 %0, %1, %2, %1" : "=v"(yn) : "v"(y0), "v"(q2));                                     // chain: 1 / sqrt(pivot)
    col_fill64<C, C + 8, 2>(a, ln);
    BLR_VA("v_mul_f64 %0, -%1, %2" : "=v"(lnn) : "v"(a[n]), "v"(yn));                                     // chain: -l of column C + 1
    BLR_VA("v_mul_f64 %0, %0, %1" : "+v"(a[n]) : "v"(yn));                                                // chain:  l of column C + 1
    rsr = (r == n) ? yn : rsr;                                                                            // the lane's own 1 / L(r, r)
    // the remaining fillers (at least two instructions: the wait states before a[n]'s DPP readers)
    col_fill64<C, C + 10, 16>(a, ln);
    if constexpr (C + 11 >= 16) BLR_VA("s_nop 1");  // (fewer than two fillers left)
    tile_factor_col_f64h<C + 1>(a, y, lnn, yn, rsr, c15, r);
  } else {
    BLR_VA("v_mul_f64 %0, %1, %2" : "=v"(t) : "v"(ln), "v"(rs));
    t = (r > C) ? t : 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) vfmac_bc16<C>(y[j], y[j], t);
  }
}
// a[c] in: A(r, c), out: L(r, c); y[j] out: L(r, r) Linv(r, 4 j + q) (as tile_factor_invert); rsr out: 1 / L(r, r) -- +inf or NaN from the
// first non-positive pivot on
__device__ __forceinline__ void tile_factor_invert_f64h(double (&a)[16], double (&y)[4], double& rsr, int lane) {
  const int r = lane & 15, q = lane >> 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) y[j] = (r == 4 * j + q) ? 1.0 : 0.0;
  double c15 = 1.5;
  asm volatile("" : "+v"(c15));
  double y0, x, hx, h, e, p, q2, yn, ln;
  BLR_VA("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "=v"(x) : "v"(a[0]));
  BLR_VA("v_rsq_f64 %0, %1" : "=v"(y0) : "v"(x));
  BLR_VA("v_mul_f64 %0, %1, 0.5" : "=v"(hx) : "v"(x));
  BLR_VA("v_mul_f64 %0, %1, %2" : "=v"(h) : "v"(hx), "v"(y0));
  BLR_VA("v_fma_f64 %0, -%1, %2, 0.5" : "=v"(e) : "v"(h), "v"(y0));
  BLR_VA("v_fma_f64 %0, %1, %2, 1.0" : "=v"(p) : "v"(e), "v"(c15));
  BLR_VA("v_mul_f64 %0, %1, %2" : "=v"(q2) : "v"(e), "v"(p));
  BLR_VA("v_fma_f64 %0, %1, %2, %1" : "=v"(yn) : "v"(y0), "v"(q2));
  BLR_VA("v_mul_f64 %0, -%1, %2" : "=v"(ln) : "v"(a[0]), "v"(yn));
  BLR_VA("v_mul_f64 %0, %0, %1\n\ts_nop 1" : "+v"(a[0]) : "v"(yn));
  rsr = yn;  // (lane 0's; the others pick theirs up in their column)
  tile_factor_col_f64h<0>(a, y, ln, yn, rsr, c15, r);
}

// On entry: P (LDS offset 0) = packed lower triangle of A, bvec (SmallCfg<double, 8>::OFF_B) = b.  On exit: P = L, bvec = u = L^-1 b,
// [OFF_W, + 16 KiB) = W_J[i][c] = (L_JJ^-1)(i, c) as [8][16][16] doubles.  [OFF_U, + 1 KiB) is scratch (u while b is still being read).
// Returns 0 or the LAPACK-style 1-based index of the failing leading minor.  Four-wave code (256 threads), one instruction stream.
template <int OFF_W, int OFF_U>
__device__ __attribute__((noinline)) int chol128_dpp(char* smem) {
  using T = double;
  using C = SmallCfg<double, 8>;
  using acc4 = typename Mfma<T>::acc4;
  constexpr int TPW = C::TPW;
  T* const P = reinterpret_cast<T*>(smem);
  T* const bvec = reinterpret_cast<T*>(smem + C::OFF_B);
  T* const Wst = reinterpret_cast<T*>(smem + OFF_W);
  T* const ust = reinterpret_cast<T*>(smem + OFF_U);
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int pr = (r * (r + 1)) >> 1;  // pidx(r, 0)
  T* const dummy = reinterpret_cast<T*>(smem + C::OFF_DINV) + r;
  int cr[4], pcr[4];  // C-layout rows of this lane inside a tile, and pidx(cr, 0)
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    cr[v] = Mfma<T>::crow(lane, v);
    pcr[v] = (cr[v] * (cr[v] + 1)) >> 1;
  }
  // trailing matrix: the tiles of block rows `wave` and 7 - `wave` in accumulators (diagonal tiles: mirrored to full symmetric tiles)
  acc4 acc[TPW];
  int tI[TPW], tK[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    int I, K;
    wave_tile(8, wave, i, I, K);
    tI[i] = uni(I);
    tK[i] = uni(K);
    const int col = 16 * K + r;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int row = 16 * I + cr[v];
      acc[i][v] = P[pidx(max(row, col), min(row, col))];
    }
  }

  int info = 0;
  BLR_STAMP_INIT;
  BLR_STAMP(0);
#pragma unroll 1
  for (int J = 0; J < 8; ++J) {
    __syncthreads();  // block column J is final in P (stored by the owners of its tiles right after update J - 1)
    BLR_STAMP(1);
    // (1) the diagonal tile, row r of it in every lane (entries right of the diagonal: dead values, see tile_factor_col)
    T* const rowp = P + ((128 * J * J + 8 * J) + 16 * J * r + pr + 16 * J);  // pidx(16 J + r, 16 J)
    T a[16], y[4];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = rowp[c];
    T rsr;
    tile_factor_invert_f64h(a, y, rsr, lane);
    const uint64_t badm = __ballot(!(rsr > T(0) && rsr < __builtin_inf()));  // a non-positive pivot: +inf or NaN from there on
    if (badm != 0) {  // (uniform, and the same in every wave)
      info = 16 * J + __builtin_ctzll(badm) + 1;
      break;
    }
    T w[4];  // (L_JJ^-1)(r, 4 j + q)
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = y[j] * rsr;
    BLR_STAMP(2);
    // u_J = L_JJ^-1 r_J
    const T* const bj = bvec + 16 * J + q;
    T up = T(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) up = __builtin_fma(w[j], bj[4 * j], up);
    up += xor_lanes(up, 16);
    up += xor_lanes(up, 32);
    // the tile's L, its inverse and u_J go to LDS, one wave each
    if (wave == (J & 3)) {
      if (q == 0) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          T* dst = (c <= r) ? rowp + c : dummy;
          *dst = a[c];
        }
      }
    } else if (wave == ((J + 1) & 3)) {
#pragma unroll
      for (int j = 0; j < 4; ++j) Wst[(16 * J + r) * 16 + 4 * j + q] = w[j];
      if (q == 0) ust[16 * J + r] = up;
    }
    // (2) L_IJ = A_IJ L_JJ^-T: X(m, n) = sum_k A_IJ(m, k) W(n, k) -- A fragment from P, B fragment = w
#pragma unroll 1
    for (int I = J + 1 + wave; I < 8; I += 4) {
      const int tb = (128 * I * I + 8 * I) + 16 * J;   // pidx(16 I, 16 J)
      const T* pa = P + (tb + 16 * I * r + pr + q);    // pidx(16 I + r, 16 J + q)
      T fa[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) fa[ks] = pa[4 * ks];
      acc4 x = {T(0), T(0), T(0), T(0)};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) x = Mfma<T>::mma(fa[ks], w[ks], x);
#pragma unroll
      for (int v = 0; v < 4; ++v) P[tb + 16 * I * cr[v] + pcr[v] + r] = x[v];  // L(16 I + cr, 16 J + r)
    }
    BLR_STAMP(3);
    __syncthreads();
    BLR_STAMP(4);
    if (J == 7) break;
    // (3a) r_K -= L_KJ u_J for the block rows below (dealt like the solves)
    {
      T uq[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) uq[j] = ust[16 * J + 4 * j + q];
#pragma unroll 1
      for (int K = J + 1 + wave; K < 8; K += 4) {
        const T* pa = P + ((128 * K * K + 8 * K) + 16 * J + 16 * K * r + pr + q);
        T s = T(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) s = __builtin_fma(pa[4 * j], uq[j], s);
        s += xor_lanes(s, 16);
        s += xor_lanes(s, 32);
        if (q == 0) bvec[16 * K + r] -= s;
      }
    }
    // (3b) trailing update from the finished panel; block column J + 1 is final after it and goes straight back to P
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
  if (info == 0 && tid < 128) bvec[tid] = ust[tid];
  __syncthreads();
  BLR_STAMP_FLUSH;
  return info;
}

// ---- the same factorisation with the serial part on ONE wave ---------------------------------------------------------------------------------
// chol128_dpp lets every wave factor the diagonal tile: 2.8 k cycles per tile that no wave spends on anything else (the tile is bound by
// the ISSUE of its ~350 fp64 instructions on one wave, not by their latency), then solves and updates -- 60 k cycles per factorisation,
// no better than phase_chol.  Here wave 0 is the CHAIN wave, it owns block row J + 1 at step J and never waits for the bulk of a trailing
// update; the trailing matrix stays in LDS (any wave takes any tile).  Per block column J, two barriers:
//   B2  chain wave: solve tile (J+1, J), r_{J+1} -= L u_J, update tile (J+1, J+1)   |  waves 1 - 3: solve the tiles (I, J), I > J + 1,
//                                                                                   |  and take r_I -= L_IJ u_J along
//   B3  chain wave: factor + invert tile (J+1, J+1); L, W, u to LDS                 |  waves 1 - 3: trailing update J of every other tile
template <int OFF_W, int OFF_U, int NW = 4>
__device__ __attribute__((noinline)) int chol128_cw(char* smem) {
  // update waves: with 8 waves, wave 4 -- the chain wave's partner on SIMD 0 -- only keeps the barriers company (as an update wave it
  // took issue slots from the chain: 33.1 us against 3x.x with the SIMD left to the chain wave)
  constexpr int NU = NW == 8 ? 6 : NW - 1;
  using T = double;
  using C = SmallCfg<double, 8>;
  using acc4 = typename Mfma<T>::acc4;
  T* const P = reinterpret_cast<T*>(smem);
  T* const bvec = reinterpret_cast<T*>(smem + C::OFF_B);
  T* const Wst = reinterpret_cast<T*>(smem + OFF_W);
  T* const ust = reinterpret_cast<T*>(smem + OFF_U);
  int* const iscr = reinterpret_cast<int*>(smem + C::OFF_SCR + 64);
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int uw = (NW == 8) ? (wave < 4 ? wave - 1 : wave - 2) : wave - 1;  // update-wave index; wave 4 of 8: 2 -> see NU (it gets none)
  const bool idle = (NW == 8) && wave == 4;
  const int r = lane & 15, q = lane >> 4;
  const int pr = (r * (r + 1)) >> 1;  // pidx(r, 0)
  T* const dummy = reinterpret_cast<T*>(smem + C::OFF_DINV) + r;
  int cr[4], pcr[4];  // C-layout rows of this lane inside a tile, and pidx(cr, 0)
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    cr[v] = Mfma<T>::crow(lane, v);
    pcr[v] = (cr[v] * (cr[v] + 1)) >> 1;
  }
  // P_IK -= L_IJ L_KJ' for TWO tiles of the trailing matrix at a time, in place (diagonal tiles: the lower half).  `two` false: the
  // second tile is absent (its reads repeat the first tile's, its writes go to the dump word).
  auto update_tiles = [&](int I0, int K0, int I1, int K1, bool two, int J) {
    const int bI0 = 128 * I0 * I0 + 8 * I0, bK0 = 128 * K0 * K0 + 8 * K0;  // pidx(16 I, 0)
    const int bI1 = 128 * I1 * I1 + 8 * I1, bK1 = 128 * K1 * K1 + 8 * K1;
    const int fo = pr + 16 * J + q;
    const T* pI0 = P + (bI0 + 16 * I0 * r + fo);  // pidx(16 I + r, 16 J + q)
    const T* pK0 = P + (bK0 + 16 * K0 * r + fo);
    const T* pI1 = P + (bI1 + 16 * I1 * r + fo);
    const T* pK1 = P + (bK1 + 16 * K1 * r + fo);
    T fa0[4], fb0[4], fa1[4], fb1[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { fa0[ks] = pI0[4 * ks]; fb0[ks] = pK0[4 * ks]; fa1[ks] = pI1[4 * ks]; fb1[ks] = pK1[4 * ks]; }
    T *pc0[4], *pc1[4];
    acc4 acc0, acc1;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      pc0[v] = (I0 != K0 || r <= cr[v]) ? P + (bI0 + 16 * I0 * cr[v] + pcr[v] + 16 * K0 + r) : dummy;
      pc1[v] = (two && (I1 != K1 || r <= cr[v])) ? P + (bI1 + 16 * I1 * cr[v] + pcr[v] + 16 * K1 + r) : dummy;
      acc0[v] = *pc0[v];
      acc1[v] = *pc1[v];
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      acc0 = Mfma<T>::mma(-fa0[ks], fb0[ks], acc0);
      acc1 = Mfma<T>::mma(-fa1[ks], fb1[ks], acc1);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) { *pc0[v] = acc0[v]; *pc1[v] = acc1[v]; }
  };
  auto update_one = [&](int I, int K, int J) {
    const int bI = 128 * I * I + 8 * I, bK = 128 * K * K + 8 * K;
    const int fo = pr + 16 * J + q;
    const T* pI = P + (bI + 16 * I * r + fo);
    const T* pK = P + (bK + 16 * K * r + fo);
    T fa[4], fb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { fa[ks] = pI[4 * ks]; fb[ks] = pK[4 * ks]; }
    T* pc[4];
    acc4 acc;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      pc[v] = (I != K || r <= cr[v]) ? P + (bI + 16 * I * cr[v] + pcr[v] + 16 * K + r) : dummy;
      acc[v] = *pc[v];
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) acc = Mfma<T>::mma(-fa[ks], fb[ks], acc);
#pragma unroll
    for (int v = 0; v < 4; ++v) *pc[v] = acc[v];
  };
  // L_IJ = A_IJ L_JJ^-T (B fragment = W_J) for one tile, then r_I -= L_IJ u_J from the solved tile's own fragments
  auto solve_row = [&](int I, int J, const T (&w)[4], const T (&uq)[4]) {
    const int tb = (128 * I * I + 8 * I) + 16 * J;   // pidx(16 I, 16 J)
    const T* pa = P + (tb + 16 * I * r + pr + q);    // pidx(16 I + r, 16 J + q)
    T fa[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fa[ks] = pa[4 * ks];
    acc4 x = {T(0), T(0), T(0), T(0)};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) x = Mfma<T>::mma(fa[ks], w[ks], x);
#pragma unroll
    for (int v = 0; v < 4; ++v) P[tb + 16 * I * cr[v] + pcr[v] + r] = x[v];  // L(16 I + cr, 16 J + r)
    T sacc = T(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) sacc = __builtin_fma(pa[4 * j], uq[j], sacc);  // (the wave's own stores: in order)
    sacc += xor_lanes(sacc, 16);
    sacc += xor_lanes(sacc, 32);
    if (q == 0) bvec[16 * I + r] -= sacc;
  };
  // the chain wave's tile: factor + invert (J, J); L_JJ, W_J, u_J = W_J r_J and the verdict on the pivots to LDS
  auto factor_tile = [&](int J) {
    T* const rowp = P + ((128 * J * J + 8 * J) + 16 * J * r + pr + 16 * J);  // pidx(16 J + r, 16 J)
    T a[16], y[4], rsr, rj[4];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = rowp[c];
#pragma unroll
    for (int j = 0; j < 4; ++j) rj[j] = bvec[16 * J + 4 * j + q];
    tile_factor_invert_f64h(a, y, rsr, lane);
    const uint64_t badm = __ballot(!(rsr > T(0) && rsr < __builtin_inf()));  // a non-positive pivot: +inf or NaN from there on
    if (lane == 0) iscr[7] = (badm != 0) ? 16 * J + __builtin_ctzll(badm) + 1 : 0;
    T w[4], up = T(0);  // w = (L_JJ^-1)(r, 4 j + q)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      w[j] = y[j] * rsr;
      up = __builtin_fma(w[j], rj[j], up);
    }
    up += xor_lanes(up, 16);
    up += xor_lanes(up, 32);
#pragma unroll
    for (int j = 0; j < 4; ++j) Wst[(16 * J + r) * 16 + 4 * j + q] = w[j];
    if (q == 0) {
      ust[16 * J + r] = up;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        T* dst = (c <= r) ? rowp + c : dummy;
        *dst = a[c];
      }
    }
  };

  int info = 0;
  BLR_STAMP_INIT;
  __syncthreads();
  BLR_STAMP(0);
  if (wave == 0) factor_tile(0);
  BLR_STAMP(1);
#pragma unroll 1
  for (int J = 0; J < 8; ++J) {
    __syncthreads();  // B2: L_JJ, W_J, u_J and the verdict on the pivots are in LDS; trailing update J - 1 is complete
    BLR_STAMP(2);
    info = iscr[7];
    if (info != 0 || J == 7) break;  // (uniform)
    {
      T w[4], uq[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { w[j] = Wst[(16 * J + r) * 16 + 4 * j + q]; uq[j] = ust[16 * J + 4 * j + q]; }
      if (wave == 0) {
        solve_row(J + 1, J, w, uq);
        update_one(J + 1, J + 1, J);
      } else if (!idle) {
#pragma unroll 1
        for (int I = J + 2 + uw; I < 8; I += NU) solve_row(I, J, w, uq);
      }
    }
    BLR_STAMP(3);
    __syncthreads();  // B3: block column J of L is in P, tile (J + 1, J + 1) is final
    BLR_STAMP(4);
    if (wave == 0) {
      factor_tile(J + 1);
    } else if (!idle) {
      // trailing update J of the tiles (I, K), J + 1 <= K <= I <= 7, without (J + 1, J + 1): dealt over the three update waves in pairs
      const int n1 = 7 - J;  // tiles per edge
      const int nt = n1 * (n1 + 1) / 2;
      auto tile_of = [&](int t, int& I, int& K) {
        int i = 0;
#pragma unroll
        for (int k = 1; k < 7; ++k) i += (t >= k * (k + 1) / 2) ? 1 : 0;
        I = J + 1 + i;
        K = J + 1 + t - i * (i + 1) / 2;
      };
#pragma unroll 1
      for (int t = 1 + uw; t < nt; t += 2 * NU) {  // (t = 0 is the chain wave's tile)
        int I0, K0, I1, K1;
        const bool two = t + NU < nt;
        tile_of(t, I0, K0);
        tile_of(two ? t + NU : t, I1, K1);
        update_tiles(I0, K0, I1, K1, two, J);
      }
    }
    BLR_STAMP(5);
  }
  __syncthreads();
  if (info == 0 && tid < 128) bvec[tid] = ust[tid];
  __syncthreads();
  BLR_STAMP_FLUSH;
  return info;
}

}  // namespace blr
