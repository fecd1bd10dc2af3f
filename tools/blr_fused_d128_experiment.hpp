// EXPERIMENT (round 3), not part of the library: D = 128 exactly, ColVecs with 16-byte aligned columns, isotropic noise,
// diagonal prior precision -- the headline shape of BASELINE config 2 at THREE workgroups per CU.  Parity-green when it was
// wired into the dispatcher (37 cases against the oracle and the general kernel), and NOT faster: 0.684 M updates/s against
// 0.698 for the two-workgroup kernel in steady state.  The premise below -- a third workgroup fills the matrix pipe while one
// factorises -- is right in CYCLES and irrelevant in TIME: on random operands streamed from HBM the part holds 1.7-1.8 GHz
// under this load (2.15 GHz with the operands cache-resident, 2.3-2.4 GHz on zeros), the ring loop ALONE sustains 50 TF
// (tools/ring_probe 8192 1), which is what the whole fused kernel delivers -- the kernel is bound by the power the f64 matrix
// pipe plus the HBM stream draw, and a third workgroup only lowers the clock further (1.67 GHz).  Kept for the record:
//   tools/b.sh fused_bench _d128 -DFB_D128 ; ./fused_bench_d128 3840 4096 0 10
//
//   reference src/bayesian_linear_regression.jl:55-69, :72-89 -- the same direct Gram form and the same four phases as
//   fused_small_kernel (blr_fused_small.hpp), whose general path keeps every other case.
//
// Why a second kernel.  fused_small_kernel keeps the packed triangle P of A (66 KB in f64) in LDS from the end of the Gram loop
// to the end of the back substitution, which caps it at two workgroups per CU.  Measured (tools/ring_probe.hip):
// the Gram ring loop alone sustains 67.6 TF with two workgroups per CU and 69.2 TF with three -- the f64 matrix pipe is
// saturated in CYCLES either way (the clock the part holds under this load is what is left) -- but the whole kernel reaches
// 50 TF, because while one workgroup factorises and substitutes (13 % of its time, dependent f64 vector chains that
// occupy the shared DP datapath) the other one alone cannot fill the matrix pipe.  With three workgroups two are left to
// do so.  To fit three (<= 53 KB of LDS each, <= 168 registers):
//   * the ring has three half-stages (48 KB) instead of four (gram_iso_ring<.., NH = 3>);
//   * A never lives in LDS: the Gram phase hands its accumulator tiles to the factorisation through a per-workgroup
//     scratch in global memory (73.7 KB; each wave reads back exactly the lanes it wrote, so the hand-off needs no
//     barrier and stays in L2) -- the phases remain separate NOINLINE functions with their own register allocation;
//   * the factorisation keeps the trailing matrix in the MFMA accumulators as before and exchanges ONE 128 x 16 panel at a
//     time through LDS (two 17 KB buffers, row stride 17: conflict-free for the C-layout store, the row-per-lane load and
//     the fragment reads); a finished panel of L goes straight to T = L' in global memory (T_post, or scratch when the
//     caller wants no factor), 128 contiguous bytes per lane;
//   * the back substitution m = L^-T u runs block-wise on all four waves with T read back from L2.
// T_post and mw_post of a regressor whose info != 0 are undefined.
#pragma once
#include "blr_fused_small.hpp"

namespace blr {

template <typename T>
struct D128Cfg {
  static constexpr int NB = 8, DP = 128, NH = 3;
  static constexpr int HALF = 4 * NB * 64;                                  // elements of a ring half (4 k-steps = 16 columns)
  static constexpr int RING_BYTES = NH * HALF * (int)sizeof(T);             // 48 KB (f64)
  static constexpr int PLD = 17;                                            // row stride of a panel buffer
  static constexpr int PANEL = DP * PLD;                                    // elements of one panel buffer
  static_assert(2 * PANEL * (int)sizeof(T) <= RING_BYTES, "the two panel buffers alias the ring");
  static_assert(16 * DP * 8 <= RING_BYTES, "the b-partial scratch aliases the ring");
  static constexpr int OFF_Y = RING_BYTES;                                  // ybuf[NH][16]   (LDS-DMA target)
  static constexpr int OFF_B = OFF_Y + NH * 16 * (int)sizeof(T);            // bvec[DP]: b, then u, then m
  static constexpr int OFF_MW = OFF_B + DP * (int)sizeof(T);                // mwl[DP]
  static constexpr int OFF_DG = OFF_MW + DP * (int)sizeof(T);               // diagL[DP]
  static constexpr int OFF_SCR = (OFF_DG + DP * (int)sizeof(T) + 15) & ~15;  // 8 doubles + 8 ints
  static constexpr int OFF_CTX = OFF_SCR + 96;
  static constexpr int LDS_BYTES = (OFF_CTX + 96 + 15) & ~15;
  static_assert(3 * LDS_BYTES <= 160 * 1024, "three workgroups per CU");
  static constexpr int TILE_SCRATCH = 36 * 256;                             // elements: the 36 lower tiles of A, C layout
  static constexpr int WG_SCRATCH = TILE_SCRATCH + DP * DP;                 // + T when the caller passes no T_post
};

template <typename T>
struct D128Ctx {  // per-regressor, uniform; handed to the phases through LDS
  const T* X; const T* y; const T* mw; const T* dprior;
  T* Lw_post; T* Ascr; T* Tst;
  int64_t ldx, ldlp, ldt;
  int N;
  T s_iso;
};

// ---- phase 1: streaming Gram.  Out: the tiles of A = diag(d) + X X' / s in the scratch (C layout, slot [wave][i][v][lane]),
//      bvec = b = X (y - X'mw) / s, scr[4] = quadratic form, optionally Lw_post = A (full symmetric).
template <typename T>
BLR_PHASE void d128_gram(char* smem) {
  using C = D128Cfg<T>;
  using acc4 = typename Mfma<T>::acc4;
  T* const ring = reinterpret_cast<T*>(smem);
  double* const red = reinterpret_cast<double*>(smem);
  T* const ybuf = reinterpret_cast<T*>(smem + C::OFF_Y);
  T* const bvec = reinterpret_cast<T*>(smem + C::OFF_B);
  T* const mwl = reinterpret_cast<T*>(smem + C::OFF_MW);
  double* const scr = reinterpret_cast<double*>(smem + C::OFF_SCR);
  const D128Ctx<T>* ctx = reinterpret_cast<const D128Ctx<T>*>(smem + C::OFF_CTX);
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int wave = uni(tid >> 6);
  const BLR_GLOBAL T* X = as_global(uni(ctx->X));
  const BLR_GLOBAL T* y = as_global(uni(ctx->y));
  const BLR_GLOBAL T* mw = as_global(uni(ctx->mw));
  const BLR_GLOBAL T* dpr = as_global(uni(ctx->dprior));
  BLR_GLOBAL T* Lw_post = as_global(uni(ctx->Lw_post));
  BLR_GLOBAL T* Ascr = as_global(uni(ctx->Ascr));
  const int64_t ldx = uni(ctx->ldx), ldlp = uni(ctx->ldlp);
  const int N = uni(ctx->N);
  const T s_iso = ctx->s_iso;

  acc4 acc[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) acc[i] = acc4{T(0), T(0), T(0), T(0)};
  double bacc[8];
#pragma unroll
  for (int I = 0; I < 8; ++I) bacc[I] = 0.0;
  double qacc = 0.0;
  if (tid < C::DP) mwl[tid] = mw[tid];
  __syncthreads();
  T mwf[8];
#pragma unroll
  for (int I = 0; I < 8; ++I) mwf[I] = mwl[16 * I + (lane & 15)];
  bool mwz = true;
#pragma unroll
  for (int I = 0; I < 8; ++I) mwz = mwz && (mwf[I] == T(0));
  mwz = __all(mwz);  // the 64 lanes of a wave hold all 128 entries of mw: wave-uniform, and the same in every wave
  const T wiso = T(1) / s_iso;
  const unsigned voff = glds_lane_offset<T>(ldx, lane);
  auto ring_run = [&](auto ws, auto mz) {
    gram_iso_ring<T, decltype(ws)::value, decltype(mz)::value, C::NH>(ring, ybuf, X, y, ldx, N, voff, lane, acc, bacc, qacc, mwf, wiso);
  };
  auto ring_w = [&](auto mz) {
    switch (wave) {
      case 0: ring_run(std::integral_constant<int, 0>{}, mz); break;
      case 1: ring_run(std::integral_constant<int, 1>{}, mz); break;
      case 2: ring_run(std::integral_constant<int, 2>{}, mz); break;
      default: ring_run(std::integral_constant<int, 3>{}, mz); break;
    }
  };
  if (mwz) ring_w(std::true_type{});
  else ring_w(std::false_type{});

  // A = diag(d) + (1 / s) X X'  -> scratch (and Lw_post); wave W owns block rows W and 7 - W (wave_tile)
  const int r = lane & 15;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    int I, K;
    wave_tile(8, wave, i, I, K);
    const int col = 16 * K + r;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int row = 16 * I + Mfma<T>::crow(lane, v);
      T val = acc[i][v] * wiso;
      if (row == col) val += dpr[row];  // only on the diagonal tiles (I == K is wave-uniform, the compare is cheap)
      acc[i][v] = val;
      Ascr[((wave * 9 + i) * 4 + v) * 64 + lane] = val;
      if (Lw_post != nullptr) {  // posterior precision Lw' = A, full symmetric (:92): the tile and its mirror image
        Lw_post[(int64_t)col * ldlp + row] = val;
        if (I != K) Lw_post[(int64_t)row * ldlp + col] = val;
      }
    }
  }

  // b partials -> LDS -> fixed-order sum (the ring is free: the loop ended with every half consumed)
  __syncthreads();
  {
    const int q = lane >> 4;
#pragma unroll
    for (int I = 0; I < 8; ++I) red[(wave * 4 + q) * C::DP + 16 * I + r] = bacc[I];
  }
  __syncthreads();
  if (tid < C::DP) {
    double sum = 0.0;
#pragma unroll
    for (int p = 0; p < 16; ++p) sum += red[p * C::DP + tid];  // fixed order
    bvec[tid] = (T)sum;  // b = X S (y - X'mw)   (:57 Bt'dy, unwhitened; the ring loop applied 1 / s to r_n = delta_n / s already)
  }
  const double quad = block_allreduce(qacc, scr, tid);  // (y-m)' S (y-m)
  if (tid == 0) scr[4] = quad;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tile stores have left this wave before the next phase reads them back
  __syncthreads();
}

// ---- phase 2: blocked Cholesky, trailing matrix in the accumulators, one panel at a time through LDS; u = L^-1 b rides along.
//      Out: T = L' in Tst (global, column-major, ldt; the strictly-lower part INSIDE the diagonal blocks zeroed), bvec = u,
//      diagL = diag(L).  Returns 0 or the LAPACK-style index of the failing leading minor.
template <typename T>
BLR_PHASE int d128_chol(char* smem) {
  using C = D128Cfg<T>;
  using acc4 = typename Mfma<T>::acc4;
  constexpr int PLD = C::PLD;
  T* const pan0 = reinterpret_cast<T*>(smem);
  T* const bvec = reinterpret_cast<T*>(smem + C::OFF_B);
  T* const diagL = reinterpret_cast<T*>(smem + C::OFF_DG);
  const D128Ctx<T>* ctx = reinterpret_cast<const D128Ctx<T>*>(smem + C::OFF_CTX);
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int wave = uni(tid >> 6);
  const BLR_GLOBAL T* Ascr = as_global(uni(ctx->Ascr));
  BLR_GLOBAL T* Tst = as_global(uni(ctx->Tst));
  const int64_t ldt = uni(ctx->ldt);
  const int r = lane & 15, q = lane >> 4;
  // per-lane parts of the panel-buffer addresses; the tile-dependent part (16 PLD I) stays SCALAR and is kept out of reach of
  // loop-invariant code motion (36 hoisted address registers were what pushed this phase over 168 registers)
  int cst[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) cst[v] = Mfma<T>::crow(lane, v) * PLD + r;
  const int cfr = r * PLD + q;
  auto opaque_s = [](int v) { asm volatile("" : "+s"(v)); return v; };

  acc4 acc[9];
  int tI[9], tK[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    int I, K;
    wave_tile(8, wave, i, I, K);
    tI[i] = uni(I);
    tK[i] = uni(K);
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[i][v] = Ascr[((wave * 9 + i) * 4 + v) * 64 + lane];  // the lanes this wave wrote itself
  }

  int info = 0;
  for (int J = 0; J < 8; ++J) {
    T* const pan = pan0 + (J & 1) * C::PANEL;
    // (a) block column J -> panel buffer (C layout -> rows); buffer J & 1 was last read in iteration J - 2
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      if (tK[i] == J) {  // scalar branch
        T* const pt = pan + opaque_s(16 * PLD * tI[i]);
#pragma unroll
        for (int v = 0; v < 4; ++v) pt[cst[v]] = acc[i][v];
      }
    }
    __syncthreads();
    // (b) one row per lane: lanes 0-15 the diagonal-block rows (redundantly in all four waves), lanes 16-63 rows below.
    //     Entries to the right of the diagonal of a diagonal-block row are dead values (see phase_chol)
    const bool is_diag = lane < 16;
    const int ri = is_diag ? 16 * J + lane : 16 * (J + 1) + 48 * wave + (lane - 16);
    const bool active = ri < C::DP;
    const int ria = active ? ri : 0;
    T* const rowp = pan + ria * PLD;
    T arow[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) arow[c] = rowp[c];
    T bl = bvec[ria];
    T own_rsq = T(1), own_diag = T(1);
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const T d2 = readlane(arow[c], c);
      if (!(d2 > T(0))) {  // wave-uniform (SGPR) and identical in all four waves
        if (info == 0) info = 16 * J + c + 1;
      }
      const T t = arow[c] * fast_rcp(d2);
      const T bc = readlane(bl, c);
      if (lane > c) bl -= t * bc;
#pragma unroll
      for (int k = c + 1; k < 16; ++k) {
        const T akc = readlane(arow[c], k);  // unscaled A[16J + k][16J + c]
        arow[k] -= t * akc;
      }
      const T rsq = fast_rsqrt(d2);
      if (lane == c) { own_rsq = rsq; own_diag = d2 * rsq; }
      arow[c] = (lane == c) ? d2 * rsq : arow[c] * rsq;  // L[i][c] = a_ic / sqrt(d2); diagonal = sqrt(d2)
    }
    if (is_diag) bl *= own_rsq;  // u_c = b_c / L_cc for the diagonal-block rows
    if (info != 0) break;        // uniform across the block: every wave factors the same diagonal rows
    // (c) the finished panel: back to the buffer for the trailing update (rows below the diagonal block are what it reads)
    //     and out to T = L': row ri of the panel is rows 16J .. 16J+15 of column ri of T, 128 contiguous bytes
    if (active) {
#pragma unroll
      for (int c = 0; c < 16; ++c) rowp[c] = arow[c];
      if (!is_diag || wave == 0) {
        typedef T vecT __attribute__((ext_vector_type(Mfma<T>::VEC)));
        constexpr int VEC = Mfma<T>::VEC;
        BLR_GLOBAL T* out = Tst + (int64_t)ri * ldt + 16 * J;
#pragma unroll
        for (int c = 0; c < 16; c += VEC) {
          vecT o;
#pragma unroll
          for (int e = 0; e < VEC; ++e) o[e] = (!is_diag || c + e <= lane) ? arow[c + e] : T(0);
          *reinterpret_cast<BLR_GLOBAL vecT*>(out + c) = o;
        }
        bvec[ri] = bl;
        if (is_diag) diagL[ri] = own_diag;
      }
    }
    __syncthreads();
    // (d) trailing update A_IK -= L_IJ L_KJ' for the tiles to the right of block column J
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      if (tK[i] > J) {  // scalar branch
        const T* pI = pan + opaque_s(16 * PLD * tI[i]) + cfr;
        const T* pK = pan + opaque_s(16 * PLD * tK[i]) + cfr;
        T fa[4], fb[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { fa[ks] = pI[4 * ks]; fb[ks] = pK[4 * ks]; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc[i] = Mfma<T>::mma(-fa[ks], fb[ks], acc[i]);
      }
    }
  }
  __syncthreads();
  return info;
}

// ---- phase 3: m = L^-T u block-wise on all four waves (T read back from L2), |u|^2, logdet A; the strictly-lower part of T
//      below the diagonal blocks is zeroed.  Out: bvec = m, scr[6] = |u|^2, scr[7] = logdet A.
template <typename T>
BLR_PHASE void d128_backsolve(char* smem, int zero_fill_in) {
  using C = D128Cfg<T>;
  T* const bvec = reinterpret_cast<T*>(smem + C::OFF_B);
  T* const diagL = reinterpret_cast<T*>(smem + C::OFF_DG);
  T* const part = reinterpret_cast<T*>(smem);  // [16 groups][16 rows] partial sums (the ring / panel area is free by now)
  double* const scr = reinterpret_cast<double*>(smem + C::OFF_SCR);
  const D128Ctx<T>* ctx = reinterpret_cast<const D128Ctx<T>*>(smem + C::OFF_CTX);
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int wave = uni(tid >> 6);
  BLR_GLOBAL T* Tst = as_global(uni(ctx->Tst));
  const int64_t ldt = uni(ctx->ldt);
  const bool zero_fill = uni(zero_fill_in) != 0;
  // every store of the factorisation to T has to be visible to the loads below: same CU, so draining this wave's stores and a
  // workgroup barrier are enough (the L1 is write-through)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  {
    const double u = tid < C::DP ? (double)bvec[tid] : 0.0;
    const double ld = tid < C::DP ? log((double)diagL[tid]) : 0.0;
    const double uu = block_allreduce(u * u, scr, tid);
    const double lds = block_allreduce(ld, scr, tid);
    if (tid == 0) { scr[6] = uu; scr[7] = 2.0 * lds; }
  }
  if (zero_fill) {
    // T is upper triangular: rows below the diagonal block of every column, 16 bytes per store (D x D / 2 elements in all)
    typedef T vecT __attribute__((ext_vector_type(Mfma<T>::VEC)));
    constexpr int VEC = Mfma<T>::VEC, VPC = C::DP / VEC;
    for (int e = tid; e < C::DP * VPC; e += kThreads) {
      const int c = e / VPC, r0 = (e - c * VPC) * VEC;
      if (r0 >= 16 * (c / 16 + 1)) *reinterpret_cast<BLR_GLOBAL vecT*>(Tst + (int64_t)c * ldt + r0) = vecT(T(0));
    }
  }
  const int j = tid & 15, g = tid >> 4;
  for (int J = 7; J >= 0; --J) {
    // s_j = sum over the finished entries c >= 16 (J + 1) of T[16J + j, c] m_c; group g takes c = 16 (J + 1) + g + 16 t
    T sum = T(0);
    for (int c = 16 * (J + 1) + g; c < C::DP; c += 16) sum += Tst[(int64_t)c * ldt + 16 * J + j] * bvec[c];
    part[g * 16 + j] = sum;
    __syncthreads();
    if (wave == 0) {
      // lanes 0-15: row j of the diagonal block T_JJ (upper triangular), solved from the last column to the first
      T trow[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) trow[k] = (lane < 16) ? Tst[(int64_t)(16 * J + k) * ldt + 16 * J + (lane & 15)] : T(0);
      T rhs = T(0);
      if (lane < 16) {
        rhs = bvec[16 * J + lane];
#pragma unroll
        for (int gg = 0; gg < 16; ++gg) rhs -= part[gg * 16 + lane];  // fixed order
      }
      const T dinv = (lane < 16) ? T(1) / diagL[16 * J + (lane & 15)] : T(0);
#pragma unroll
      for (int k = 15; k >= 0; --k) {
        const T mk = readlane(rhs, k) * readlane(dinv, k);
        if (lane == k) rhs = mk;
        else if (lane < k) rhs -= trow[k] * mk;
      }
      if (lane < 16) bvec[16 * J + lane] = rhs;
    }
    __syncthreads();
  }
}

// ---- the kernel: a persistent grid of 3 x (number of CUs) workgroups strides over the batch ------------------------------
template <typename T>
__global__ __launch_bounds__(kThreads, 3) void fused_d128_kernel(PosteriorArgs<T> a, T* scratch) {
  using C = D128Cfg<T>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* const bvec = reinterpret_cast<T*>(smem + C::OFF_B);
  double* const scr = reinterpret_cast<double*>(smem + C::OFF_SCR);
  int* const iscr = reinterpret_cast<int*>(smem + C::OFF_SCR + 64);
  D128Ctx<T>* ctx = reinterpret_cast<D128Ctx<T>*>(smem + C::OFF_CTX);
  const int tid = threadIdx.x;
  const int N = a.N;
  const double kNaN = __longlong_as_double(0x7ff8000000000000LL);
  T* const my_scratch = scratch + (int64_t)blockIdx.x * C::WG_SCRATCH;

  for (int64_t reg = blockIdx.x; reg < a.B; reg += gridDim.x) {
    const T* mw = a.mw + reg * a.stridemw;
    const T* dpr = a.Lw + reg * a.strideLw;
    const T s_iso = a.s[reg * a.strides];
    __syncthreads();  // previous regressor fully done with LDS
    if (tid == 0) {
      ctx->X = a.X + reg * a.strideX;
      ctx->y = a.y + reg * a.stridey;
      ctx->mw = mw;
      ctx->dprior = dpr;
      ctx->Lw_post = a.Lw_post ? a.Lw_post + reg * a.strideLp : nullptr;
      ctx->Ascr = my_scratch;
      ctx->Tst = a.T_post ? a.T_post + reg * a.strideT : my_scratch + C::TILE_SCRATCH;
      ctx->ldx = a.ldx;
      ctx->ldlp = a.ldlp;
      ctx->ldt = a.T_post ? a.ldt : C::DP;
      ctx->N = N;
      ctx->s_iso = s_iso;
    }
    // ---- phase 0: prior (reference :78): entries of the diagonal precision must be positive; logdet
    int info = 0;
    double logdet_Lw;
    {
      double v = 0.0;
      int bad = 0x7fffffff;
      if (tid < C::DP) {
        const T dv = dpr[tid];
        if (dv > T(0)) v = log((double)dv);
        else bad = tid + 1;
      }
      bad = block_min_int(bad, iscr, tid);
      if (bad != 0x7fffffff) info = bad;
      logdet_Lw = block_allreduce(v, scr, tid);
    }
    if (info == 0 && !(s_iso > T(0))) info = 1;  // reference :79: cholesky(Sigma_y) throws at the first variance
    if (info != 0) {  // block-uniform
      if (tid == 0) {
        a.info[reg] = info;
        if (a.logpdf) a.logpdf[reg] = kNaN;
      }
      continue;
    }
    __syncthreads();
    BLR_PSTAMP_INIT;
    d128_gram<T>(smem);
    BLR_PSTAMP(1);
    const double quad = scr[4];
    const double logdet_Sy = (double)N * log((double)s_iso);
    info = d128_chol<T>(smem);
    BLR_PSTAMP(3);
    if (info != 0) {
      if (tid == 0) {
        a.info[reg] = info;
        if (a.logpdf) a.logpdf[reg] = kNaN;
      }
      continue;
    }
    d128_backsolve<T>(smem, a.T_post != nullptr ? 1 : 0);
    BLR_PSTAMP(5);
#ifdef BLR_GRAM_STAMPS
#ifdef BLR_PSTAMP_ALL
    if (tid == 0) atomicAdd(&g_pstamps[15], 1ull);
#else
    if (blockIdx.x == 0 && tid == 0) g_pstamps[15] += 1;
#endif
#endif
    if (a.mw_post && tid < C::DP) a.mw_post[reg * a.stride_mwpost + tid] = mw[tid] + bvec[tid];  // :68
    if (tid == 0) {
      a.info[reg] = 0;
      if (a.logpdf) {
        const double LOG2PI = 1.8378770664093454835606594728112;
        a.logpdf[reg] = -0.5 * ((double)N * LOG2PI + logdet_Sy + quad + scr[7] - logdet_Lw - scr[6]);  // :84 + :57
      }
    }
  }
}

}  // namespace blr
