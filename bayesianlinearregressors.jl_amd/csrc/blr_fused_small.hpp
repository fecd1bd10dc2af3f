// Fused per-regressor inference kernel for D <= 128 (one 256-thread workgroup per regressor).
//
// Replaces reference src/bayesian_linear_regression.jl:72-89 (__compute_inference_quantities),
// :55-58 (logpdf), :60-69 (posterior) with the direct Gram form of SURVEY.md 0.1:
//   phase 0  (dense prior only) Cholesky of Lw in LDS -> logdet Lw, SPD check          (:78)
//   phase 1  one streaming pass over X: MFMA SYRK  G += x_n w_n x_n'  (lower 16x16 tiles only),
//            per column mu_n = x_n'mw, delta_n, b += x_n delta_n w_n, q += delta_n^2 w_n,
//            l += log s_n                                                  (:79-84, :86, :57)
//   phase 2  A = Lw + G into packed LDS; in-place Cholesky A = L L' (T = L')              (:86, :67)
//   phase 3  u = L^-1 b, m = L^-T u, mw' = mw + m, evidence                              (:57, :64, :68)
// X is read from HBM exactly once; nothing intermediate touches HBM.
//
// LDS image of a stage (4*KS columns): [k-step j][row block I][lane l] holds
//   X[16I + (l&15), n0 + 4j + (l>>4)]  -- i.e. already in MFMA operand order, so every fragment
// read is one conflict-free ds_read of 64 consecutive elements.
#pragma once
#include <utility>

#include "blr_common.hpp"

#ifndef BLR_JUNROLL
#define BLR_JUNROLL 2
#endif

namespace blr {

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
};

template <typename T, int NB>
struct SmallCfg {
  static constexpr int DP = 16 * NB;
  // k-steps (of 4 columns) per stage; chosen so PER is a multiple of the 16-byte vector width
  static constexpr int KS = (NB < 3 || ((NB & 1) && sizeof(T) == 4)) ? 16 : 8;
  static constexpr int NSC = 4 * KS;                    // columns per stage
  static constexpr int SLOT = KS * NB * 64;             // elements per LDS slot
  static constexpr int PER = SLOT / kThreads;           // elements per thread per stage
  static constexpr int NT = NB * (NB + 1) / 2;          // lower-triangular 16x16 tiles
  static constexpr int TPW = (NT + kWaves - 1) / kWaves;  // tiles per wave
  static constexpr int PACKED = DP * (DP + 1) / 2;
  static constexpr int RED_BYTES = 16 * DP * 8;          // b partials: 4 waves x 4 lane-rows x DP doubles
  static constexpr int REGION0_A = 2 * SLOT * (int)sizeof(T);
  static constexpr int REGION0_B = PACKED * (int)sizeof(T);
  static constexpr int REGION0_C = RED_BYTES;
  static constexpr int REGION0 =
      ((REGION0_A > REGION0_B ? (REGION0_A > REGION0_C ? REGION0_A : REGION0_C)
                              : (REGION0_B > REGION0_C ? REGION0_B : REGION0_C)) + 15) & ~15;
  // after region 0: ybuf[2][NSC], wbuf[2][NSC], bvec[DP], dinv[DP] (T); scratch doubles/ints
  static constexpr int OFF_Y = REGION0;
  static constexpr int OFF_W = OFF_Y + 2 * NSC * (int)sizeof(T);
  static constexpr int OFF_B = OFF_W + 2 * NSC * (int)sizeof(T);
  static constexpr int OFF_DINV = OFF_B + DP * (int)sizeof(T);
  static constexpr int OFF_MW = OFF_DINV + DP * (int)sizeof(T);
  static constexpr int OFF_SCR = (OFF_MW + DP * (int)sizeof(T) + 15) & ~15;
  static constexpr int LDS_BYTES = OFF_SCR + 64;
};

// ---- stage loader ------------------------------------------------------------------------------
// MODE 0: ColVecs data, generic (any D, ldx, alignment)   MODE 1: RowVecs data
// MODE 2: prior factor U as pseudo-observations: element (d, j) = U[j + d*ld] for j <= d, else 0
// MODE 3: ColVecs data, 16-byte vectors (D % VEC == 0, 16-byte aligned columns)
template <typename T, int NB>
struct StageRegs {
  T x[SmallCfg<T, NB>::PER];
  T yv, wv;
};

__device__ __forceinline__ int frag_off(int NB, int d, int nl) {
  return ((((nl >> 2) * NB + (d >> 4)) * 4 + (nl & 3)) << 4) + (d & 15);
}

template <typename T, int NB, int MODE>
__device__ __forceinline__ void stage_load(StageRegs<T, NB>& r, const T* __restrict__ base, int64_t ld, int D, int ncols,
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
      vecT v = *reinterpret_cast<const vecT*>(base + addr);
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

// ---- one stage of MFMA + vector work for wave W -------------------------------------------------
template <typename T, int NB>
using AccArr = typename Mfma<T>::acc4[SmallCfg<T, NB>::TPW];

template <typename T, int NB, int t, int slot_i>
__device__ __forceinline__ void mma_one(AccArr<T, NB>& acc, const T (&fa)[NB],
                                        const T (&f)[NB]) {
  if constexpr (t < SmallCfg<T, NB>::NT) {
    constexpr int I = tile_I(t), J = tile_J(t);
    acc[slot_i] = Mfma<T>::mma(fa[I], f[J], acc[slot_i]);
  }
}

template <typename T, int NB, int W, int... Is>
__device__ __forceinline__ void mma_all(AccArr<T, NB>& acc, const T (&fa)[NB],
                                        const T (&f)[NB], std::integer_sequence<int, Is...>) {
  (mma_one<T, NB, W + kWaves * Is, Is>(acc, fa, f), ...);
}

template <typename T, int NB, int W>
__device__ __forceinline__ void compute_stage(const T* __restrict__ slot, const T* __restrict__ ybuf,
                                              const T* __restrict__ wbuf,
                                              AccArr<T, NB>& acc, double (&bacc)[NB],
                                              double& qacc, const T* __restrict__ mwl, int lane, bool is_data) {
  using C = SmallCfg<T, NB>;
#pragma unroll BLR_JUNROLL
  for (int j = 0; j < C::KS; ++j) {
    T f[NB], fa[NB];
#pragma unroll
    for (int I = 0; I < NB; ++I) f[I] = slot[(j * NB + I) * 64 + lane];
    const T w = wbuf[4 * j + (lane >> 4)];
#pragma unroll
    for (int I = 0; I < NB; ++I) fa[I] = f[I] * w;
    mma_all<T, NB, W>(acc, fa, f, std::make_integer_sequence<int, C::TPW>{});
    if ((j & 3) == W && is_data) {
      // column vector work for the 4 columns of this k-step: lane (r, q) holds rows 16I + r of column q
      T mu = T(0);
#pragma unroll
      for (int I = 0; I < NB; ++I) mu += f[I] * mwl[16 * I + (lane & 15)];
      mu = row16_allreduce(mu);
      const T delta = ybuf[4 * j + (lane >> 4)] - mu;  // :82  y - mean(fx)
      const T rn = delta * w;
      if ((lane & 15) == 0) qacc += (double)delta * (double)rn;
#pragma unroll
      for (int I = 0; I < NB; ++I) bacc[I] += (double)f[I] * (double)rn;
    }
  }
}

// ---- in-LDS Cholesky on the packed lower triangle (row i at i(i+1)/2) ----------------------------
// Right-looking with deferred column scaling: one barrier per column.  On exit P holds L (A = L L'),
// dinv[j] = L[j][j].  Returns 0 or the LAPACK-style index (1-based) of the failing leading minor.
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
  if (tid < D) dinv[tid] = sqrt(P[pidx(tid, tid)]);  // dinv holds the DIAGONAL of L (true divisions below)
  __syncthreads();
  for (int i = ti; i < D; i += 16) {
    T* row = P + pidx(i, 0);
    for (int k = tk; k < i; k += 16) row[k] /= dinv[k];
  }
  if (tid < D) P[pidx(tid, tid)] = dinv[tid];
  __syncthreads();
  return 0;
}

// ---- blocked Cholesky with the trailing matrix in MFMA accumulators -------------------------------
// On entry the lower 16x16 tiles of A live in `acc` (tile t = wave + 4 i at acc[i], MFMA C layout).
// For each block column J:
//   (a) the owners of tiles (I, J) store them into the packed LDS triangle P (row i at i(i+1)/2);
//   (b) every wave loads panel rows, ONE ROW PER LANE (lanes 0-15: the 16 diagonal-block rows, held
//       redundantly by all four waves so no cross-wave traffic is needed; lanes 16-63: 48 rows below),
//       and eliminates the 16 columns in registers -- pivots and multipliers travel by v_readlane;
//       the right-hand side b rides along (forward substitution u = L^-1 b for free);
//   (c) the finished panel goes back to P and all waves apply  A_IK -= L_IJ L_KJ'  to their
//       remaining tiles with 4 MFMAs per tile, reading both operands from P in fragment order.
// Three barriers per block column instead of one (or two) per scalar column, and no LDS round trip
// for the trailing matrix.  On exit P holds L (A = L L'), bvec holds u.  Returns LAPACK-style info.
template <typename T, int NB>
__device__ __forceinline__ int chol_blocked(AccArr<T, NB>& acc, T* __restrict__ P, T* __restrict__ bvec, int D,
                                            int wave, int lane, bool with_rhs) {
  using C = SmallCfg<T, NB>;
  const int nblk = (D + 15) >> 4;
  int info = 0;
  for (int J = 0; J < nblk; ++J) {
    __syncthreads();
    // (a) panel tiles -> packed LDS
#pragma unroll
    for (int i = 0; i < C::TPW; ++i) {
      const int t = wave + kWaves * i;
      if (t < C::NT) {
        int I = 0;
        while ((I + 1) * (I + 2) / 2 <= t) ++I;
        const int K = t - I * (I + 1) / 2;
        if (K == J) {
          const int col = 16 * J + (lane & 15);
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int row = 16 * I + Mfma<T>::crow(lane, v);
            if (col <= row) P[pidx(row, col)] = acc[i][v];
          }
        }
      }
    }
    __syncthreads();
    // (b) one row per lane
    const bool is_diag = lane < 16;
    const int ri = is_diag ? 16 * J + lane : 16 * (J + 1) + 48 * wave + (lane - 16);
    const bool active = ri < D;
    const int ncols = min(16, D - 16 * J);
    T arow[16];
    T bl = T(0);
    {
      const T* src = P + pidx(active ? ri : 0, 16 * J);
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const bool ok = active && (!is_diag || c <= lane);
        arow[c] = ok ? src[c] : T(0);
      }
      if (with_rhs && active) bl = bvec[ri];
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      if (c < ncols) {
        const T d2 = readlane(arow[c], c);
        if (!(d2 > T(0))) {  // wave-uniform (SGPR) and identical in all four waves
          if (info == 0) info = 16 * J + c + 1;
        }
        const T d = sqrt(d2);
        const T rinv = T(1) / d;
        const T piv_b = readlane(bl, c);
        const T uc = piv_b * rinv;
        const T lc = arow[c] * rinv;  // column c of L for this lane's row
        arow[c] = (lane == c) ? d : lc;
        if (lane == c) bl = uc;
        else if (lane > c) bl -= lc * uc;
#pragma unroll
        for (int k = c + 1; k < 16; ++k) {
          const T lk = readlane(arow[c], k);  // L[16J + k][16J + c]
          arow[k] -= arow[c] * lk;
        }
      }
    }
    if (info != 0) break;  // uniform across the block: every wave factors the same diagonal rows
    if (active) {
      T* dst = P + pidx(ri, 16 * J);
#pragma unroll
      for (int c = 0; c < 16; ++c)
        if (!is_diag || c <= lane) dst[c] = arow[c];
      if (with_rhs && (!is_diag || wave == 0)) bvec[ri] = bl;
    }
    __syncthreads();
    // (c) trailing update from the finished panel
    const int r = lane & 15, q = lane >> 4;
#pragma unroll
    for (int i = 0; i < C::TPW; ++i) {
      const int t = wave + kWaves * i;
      if (t < C::NT) {
        int I = 0;
        while ((I + 1) * (I + 2) / 2 <= t) ++I;
        const int K = t - I * (I + 1) / 2;
        if (K > J && I < nblk) {
          const int rowI = 16 * I + r, rowK = 16 * K + r;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const T fI = P[pidx(rowI, 16 * J + 4 * ks + q)];
            const T fK = P[pidx(rowK, 16 * J + 4 * ks + q)];
            acc[i] = Mfma<T>::mma(-fI, fK, acc[i]);
          }
        }
      }
    }
  }
  __syncthreads();
  return info;
}

// ---- the kernel -----------------------------------------------------------------------------------
template <typename T, int NB, int MODE /* data loader: 0 ColVecs generic, 1 RowVecs, 3 ColVecs vector */>
__global__ __launch_bounds__(kThreads, 2) void fused_small_kernel(PosteriorArgs<T> a) {
  using C = SmallCfg<T, NB>;
  using acc4 = typename Mfma<T>::acc4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* const slot0 = reinterpret_cast<T*>(smem);
  T* const P = reinterpret_cast<T*>(smem);
  double* const red = reinterpret_cast<double*>(smem);
  T* const ybuf = reinterpret_cast<T*>(smem + C::OFF_Y);
  T* const wbuf = reinterpret_cast<T*>(smem + C::OFF_W);
  T* const bvec = reinterpret_cast<T*>(smem + C::OFF_B);
  T* const dinv = reinterpret_cast<T*>(smem + C::OFF_DINV);
  T* const mwl = reinterpret_cast<T*>(smem + C::OFF_MW);
  double* const scr = reinterpret_cast<double*>(smem + C::OFF_SCR);
  int* const iscr = reinterpret_cast<int*>(smem + C::OFF_SCR + 32);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: the per-wave switch below is a real branch
  const int D = a.D, N = a.N;
  const double kNaN = __longlong_as_double(0x7ff8000000000000LL);

  for (int reg = blockIdx.x; reg < a.B; reg += gridDim.x) {
    const T* X = a.X + (int64_t)reg * a.strideX;
    const T* y = a.y + (int64_t)reg * a.stridey;
    const T* s = a.s + (int64_t)reg * a.strides;
    const T* mw = a.mw + (int64_t)reg * a.stridemw;
    const T* Lw = a.Lw + (int64_t)reg * a.strideLw;
    int info = 0;
    double logdet_Lw = 0.0;
    acc4 acc[C::TPW];

    // Loads the lower tiles of the dense prior precision (UPPER triangle of the caller's matrix is read,
    // as LAPACK 'U' does) into the accumulators; diagonal tiles get both halves so they stay symmetric.
    auto load_dense_prior = [&]() {
#pragma unroll
      for (int i = 0; i < C::TPW; ++i) {
        const int t = wave + kWaves * i;
        int I = 0;
        while ((I + 1) * (I + 2) / 2 <= t) ++I;
        const int K = t - I * (I + 1) / 2;
        const int col = 16 * K + (lane & 15);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int row = 16 * I + Mfma<T>::crow(lane, v);
          T val = T(0);
          if (t < C::NT && row < D && col < D) {
            const int lo = min(row, col), hi = max(row, col);
            val = Lw[(int64_t)hi * a.ldl + lo];  // upper entry (lo, hi)
          }
          acc[i][v] = val;
        }
      }
    };

    // ---- phase 0: prior -----------------------------------------------------------------------
    if (a.prior_kind == PRIOR_DENSE) {
      load_dense_prior();
      info = chol_blocked<T, NB>(acc, P, bvec, D, wave, lane, false);  // :78
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
      continue;
    }

    // ---- phase 1: streaming Gram, accumulators start from the prior precision ----------------------
    if (a.prior_kind == PRIOR_DENSE) {
      load_dense_prior();
    } else {
#pragma unroll
      for (int i = 0; i < C::TPW; ++i) {
        const int t = wave + kWaves * i;
        int I = 0;
        while ((I + 1) * (I + 2) / 2 <= t) ++I;
        const int K = t - I * (I + 1) / 2;
        const int col = 16 * K + (lane & 15);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int row = 16 * I + Mfma<T>::crow(lane, v);
          acc[i][v] = (a.prior_kind == PRIOR_DIAGONAL && t < C::NT && row == col && row < D) ? Lw[row] : T(0);
        }
      }
    }
    double bacc[NB];
#pragma unroll
    for (int I = 0; I < NB; ++I) bacc[I] = 0.0;
    if (tid < C::DP) mwl[tid] = tid < D ? mw[tid] : T(0);  // visible after the first stage barrier
    double qacc = 0.0, lacc = 0.0;

    const bool prior_cols = (a.prior_kind == PRIOR_UPPER_FACTOR);
    const int nprior_stages = prior_cols ? (D + C::NSC - 1) / C::NSC : 0;
    const int ndata_stages = (N + C::NSC - 1) / C::NSC;
    const int nstages = nprior_stages + ndata_stages;
    const bool diag_noise = (a.noise_kind == NOISE_DIAGONAL);
    const T s_iso = diag_noise ? T(1) : s[0];

    StageRegs<T, NB> regs;
    auto issue = [&](int td) {  // prefetch data stage td into registers
      const int n0 = td * C::NSC;
      stage_load<T, NB, MODE>(regs, X, a.ldx, D, N, n0, tid);
      regs.yv = T(0);
      regs.wv = T(0);
      if (tid < C::NSC && n0 + tid < N) {
        regs.yv = y[n0 + tid];
        T sv = diag_noise ? s[n0 + tid] : s_iso;
        regs.wv = T(1) / sv;                       // :79/:81  Sigma_y^-1 on the diagonal
        if (diag_noise) lacc += log((double)sv);   // :84  logdet(Sigma_y)
      }
    };

    __syncthreads();  // region 0 is free (previous regressor / phase 0 done)
    if (ndata_stages > 0) issue(0);
    for (int t = 0; t < nstages; ++t) {
      const int sl = t & 1;
      T* slot = slot0 + sl * C::SLOT;
      const bool is_data = t >= nprior_stages;
      if (!is_data) {
        // prior pseudo-observations: a handful of stages, loaded synchronously (registers die here)
        StageRegs<T, NB> pr;
        const int n0 = t * C::NSC;
        stage_load<T, NB, 2>(pr, Lw, a.ldl, D, D, n0, tid);
        pr.yv = T(0);
        pr.wv = (tid < C::NSC && n0 + tid < D) ? T(1) : T(0);
        stage_store<T, NB, 2>(pr, slot, ybuf + sl * C::NSC, wbuf + sl * C::NSC, tid);
      } else {
        stage_store<T, NB, MODE>(regs, slot, ybuf + sl * C::NSC, wbuf + sl * C::NSC, tid);
      }
      __syncthreads();
      const int tdn = t + 1 - nprior_stages;  // next data stage
      if (tdn > 0 && tdn < ndata_stages) issue(tdn);  // in flight while this stage computes
      const T* yb = ybuf + sl * C::NSC;
      const T* wb = wbuf + sl * C::NSC;
      switch (wave) {
        case 0: compute_stage<T, NB, 0>(slot, yb, wb, acc, bacc, qacc, mwl, lane, is_data); break;
        case 1: compute_stage<T, NB, 1>(slot, yb, wb, acc, bacc, qacc, mwl, lane, is_data); break;
        case 2: compute_stage<T, NB, 2>(slot, yb, wb, acc, bacc, qacc, mwl, lane, is_data); break;
        default: compute_stage<T, NB, 3>(slot, yb, wb, acc, bacc, qacc, mwl, lane, is_data); break;
      }
    }

    // ---- phase 2: b -> LDS, A (registers) -> blocked Cholesky -----------------------------------------
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

    if (a.Lw_post) {  // posterior precision Lw' = A, full symmetric (:92), straight from the accumulators
      T* out = a.Lw_post + (int64_t)reg * a.strideLp;
#pragma unroll
      for (int i = 0; i < C::TPW; ++i) {
        const int t = wave + kWaves * i;
        if (t < C::NT) {
          int I = 0;
          while ((I + 1) * (I + 2) / 2 <= t) ++I;
          const int K = t - I * (I + 1) / 2;
          const int col = 16 * K + (lane & 15);
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int row = 16 * I + Mfma<T>::crow(lane, v);
            if (col <= row && row < D) {
              out[(int64_t)row * a.ldlp + col] = acc[i][v];
              out[(int64_t)col * a.ldlp + row] = acc[i][v];
            }
          }
        }
      }
    }

    info = chol_blocked<T, NB>(acc, P, bvec, D, wave, lane, true);  // :86; T = L' is chol(Lw + G).U (:67)
    if (info != 0) {
      if (tid == 0) {
        a.info[reg] = info;
        if (a.logpdf) a.logpdf[reg] = kNaN;
      }
      continue;
    }

    if (a.T_post) {
      T* out = a.T_post + (int64_t)reg * a.strideT;
      for (int idx = tid; idx < D * D; idx += kThreads) {
        int c = idx / D, r = idx % D;
        out[(int64_t)c * a.ldt + r] = (r <= c) ? P[pidx(c, r)] : T(0);
      }
    }
    if (tid < D) dinv[tid] = T(1) / P[pidx(tid, tid)];
    __syncthreads();

    // ---- phase 3: back substitution + evidence (wave 0); bvec already holds u = L^-1 b (:57) -----------
    if (wave == 0) {
      const int i0 = lane, i1 = lane + 64;
      T b0 = i0 < D ? bvec[i0] : T(0);
      T b1 = i1 < D ? bvec[i1] : T(0);
      double uu = (double)b0 * (double)b0 + (double)b1 * (double)b1;
      uu = wave_allreduce(uu);
      // backward: m = L^-T u                                    (:64, :68)
      for (int k = D - 1; k >= 0; --k) {
        T src = (k < 64) ? b0 : b1;
        T mk = readlane(src, k & 63) * dinv[k];
        if (lane == (k & 63)) { if (k < 64) b0 = mk; else b1 = mk; }
        const T* row = P + pidx(k, 0);
        if (i0 < k) b0 -= row[i0] * mk;
        if (i1 < k) b1 -= row[i1] * mk;
      }
      if (a.mw_post) {
        T* out = a.mw_post + (int64_t)reg * a.stride_mwpost;
        if (i0 < D) out[i0] = mw[i0] + b0;  // :68  mw + Uw \ m_eps
        if (i1 < D) out[i1] = mw[i1] + b1;
      }
      double ld = 0.0;
      if (i0 < D) ld += log((double)P[pidx(i0, i0)]);
      if (i1 < D) ld += log((double)P[pidx(i1, i1)]);
      ld = 2.0 * wave_allreduce(ld);  // logdet A
      if (lane == 0) {
        a.info[reg] = 0;
        if (a.logpdf) {
          const double LOG2PI = 1.8378770664093454835606594728112;
          a.logpdf[reg] = -0.5 * ((double)N * LOG2PI + logdet_Sy + quad + ld - logdet_Lw - uu);  // :84 + :57
        }
      }
    }
  }
}

}  // namespace blr
