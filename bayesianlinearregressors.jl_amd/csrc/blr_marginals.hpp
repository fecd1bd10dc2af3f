// Marginal stream for D = 128 with a factor (PDMat / posterior / dense prior after its Cholesky): var_n = |L^-1 x_n|^2 + s_n,
// mean_n = x_n'mw, L = U' (reference src/bayesian_linear_regression.jl:33, :40-43).
//
// marginals_mfma_kernel (blr_large.hpp) runs the triangular solve as a blocked SWEEP over an LDS image of the inputs: eight
// dependent chunk steps per tile, every one a round trip accumulator -> LDS -> fragment, on one workgroup per CU (150 KB of
// LDS in fp64): 0.25 of the matrix peak.  The same D^2 N flops have no dependency at all once the triangular INVERSE is
// formed: z_n' = x_n' L^-T is a plain product of the inputs with the upper-triangular M = L^-T -- 144 MFMAs per 16 inputs,
// the same count as the sweep's (a triangular inverse is triangular), every one independent of the others' results.
//   marg_image_kernel    once per regressor: M = I L^-T by the existing sweep (two tiles of 64 unit rows, one workgroup each), written
//                        out in MFMA B-fragment order (73.7 KB in fp64);
//   marginals_gemm_kernel  the stream: the image in LDS (two workgroups per CU), a wave per 16-input tile: the tile's 128 x 16
//                        entries go from HBM straight into registers as 16-byte loads (the next tile's are in flight), are
//                        the A operands of all 144 MFMAs (column block J of M needs rows d < 16 (J + 1) only), the squares of
//                        the accumulators give var, mean rides on the same registers.
// The explicit inverse costs forward accuracy cond(L) eps -- the bound of the substitution itself; the 16 x 16 diagonal blocks
// were inverted explicitly already.  Routed for D = 128, aligned ColVecs or RowVecs (scalar loads); everything else stays on the sweep kernel.
#pragma once
#include "blr_large.hpp"

namespace blr {

template <typename T>
struct MargGemmCfg {
  static constexpr int VEC = Mfma<T>::VEC;             // consecutive d per 16-byte load = MFMAs fed by one load
  static constexpr int NLOAD = kPB / (4 * VEC);        // loads per lane and tile: 16 (f64) / 8 (f32)
  static constexpr int NFRAG = 4 * 36;                 // B fragments of the image: sum_J 4 (J + 1)
  static constexpr int IMG_ELEMS = NFRAG * 64;
  static constexpr int OFF_MW = IMG_ELEMS * (int)sizeof(T);
  static constexpr int LDS_BYTES = OFF_MW + kPB * (int)sizeof(T);
  // contraction index of MFMA m (of a column block), lane group g = lane >> 4:  one 16-byte load covers VEC consecutive d
  __host__ __device__ static constexpr int d_of(int m, int g) { return 4 * VEC * (m / VEC) + VEC * g + (m % VEC); }
  __host__ __device__ static constexpr int frag0(int J) { return 2 * J * (J + 1); }  // first fragment of column block J
  // second image (gradient): groups of four fragments (Jc, J), J = Jc .. 7, in this order; first group of output block Jc
  __host__ __device__ static constexpr int frag2_0(int Jc) { return 8 * Jc - Jc * (Jc - 1) / 2; }
};

// ---- M = L^-T in B-fragment order, two workgroups per regressor (blockIdx.y: rows 0..63 / 64..127 of M) ----------------------------
template <typename T>
__global__ __launch_bounds__(kThreads) void marg_image_kernel(const T* __restrict__ U, int64_t ldu, int64_t strideU, int D, T* __restrict__ img,
                                                              const int32_t* __restrict__ info, int reg0, int Dtotal = 0, int nblk = 1,
                                                              int64_t strideB = 0, T* __restrict__ img2 = nullptr) {
  using Cfg = TrsmCfg<T>;
  using G = MargGemmCfg<T>;
  constexpr int VEC = Mfma<T>::VEC;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* const P = reinterpret_cast<T*>(smem);
  T* const Xs = reinterpret_cast<T*>(smem + Cfg::OFF_X);
  T* const dinv = reinterpret_cast<T*>(smem + Cfg::OFF_DI);
  T* const Linv = reinterpret_cast<T*>(smem + Cfg::OFF_LI);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int reg = reg0 + blockIdx.x;
  if (Dtotal) {  // the "regressors" are the nblk diagonal 128-blocks of factors of order Dtotal (marg_blocksub_kernel): one status
    const int b = reg / nblk, blk = reg % nblk;  // word, one factor (strideB elements apart) per nblk of them
    if (info && info[b] != 0) return;
    D = min(kPB, Dtotal - kPB * blk);
    U += (int64_t)b * strideB + (int64_t)blk * strideU;
  } else {
    if (info && info[reg] != 0) return;
    U += (int64_t)reg * strideU;
  }
  img += (int64_t)reg * G::IMG_ELEMS;
  if (img2) img2 += (int64_t)reg * G::IMG_ELEMS;
  const int nchunks = kPB / 16;
#ifdef BLR_IMG_STAMPS
  unsigned long long ist[6]; int isn = 0;
#define IMG_T() do { ist[isn++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define IMG_T() do {} while (0)
#endif
  IMG_T();
  const bool uvec = D == kPB && (ldu % VEC) == 0 && ((uintptr_t)U % 16) == 0;
  if (uvec) load_upper_block_to_packed(P, U, ldu, tid);
#pragma unroll 1
  for (int base = 0; !uvec && base < kPB * kPB; base += kThreads * 8) {
    T v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * kThreads + tid;
      const int r = idx / kPB, c = idx % kPB;  // L[r][c] = U[c, r]
      const bool ok = c <= r && r < D;
      v[u] = U[ok ? (int64_t)r * ldu + c : 0];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * kThreads + tid;
      const int r = idx / kPB, c = idx % kPB;
      if (c <= r) P[pidx(r, c)] = (r < D) ? v[u] : (r == c ? T(1) : T(0));  // (padding: unit diagonal)
    }
  }
  __syncthreads();
  IMG_T();
  if (tid < kPB) dinv[tid] = T(1) / P[pidx(tid, tid)];
  __syncthreads();
  IMG_T();
  trsm_prepare<T>(P, dinv, Linv, nchunks, tid);
  IMG_T();
  {
    const int half = blockIdx.y;
    __syncthreads();
    for (int e = tid; e < Cfg::RB * kPB; e += kThreads) {
      const int r = e / kPB, c = e % kPB;
      Xs[r * Cfg::LDX + c] = (c == 64 * half + r) ? T(1) : T(0);
    }
    __syncthreads();
    trsm_sweep<T>(Xs, P, Linv, nchunks, lane, wave);  // rows 64 half .. + 63 of L^-T (ends with a barrier)
    IMG_T();
    // image entries whose contraction index d falls into this half: fragment m of column block J to wave m % 4 (a flat loop over
    // the 9216 entries with a search for J per entry was half of this kernel's time)
    const int g4 = lane >> 4;
#pragma unroll 1
    for (int J = 0; J < 8; ++J) {
      for (int m = wave; m < 4 * (J + 1); m += kWaves) {
        const int f = G::frag0(J) + m;
        const int d = G::d_of(m, g4);
        // (diagonal blocks of a large factor: the VEC fragments one 16-byte load of the inputs feeds sit next to each other per lane)
        const int at = Dtotal ? ((f / VEC) * 64 + lane) * VEC + (f % VEC) : f * 64 + lane;
        if ((d >> 6) == half) img[at] = Xs[(d & 63) * Cfg::LDX + 16 * J + (lane & 15)];
      }
    }
    // the second image (grad_gemm_kernel): B fragments of M' for g = z M' -- fragment (Jc, J, v), J >= Jc, holds M[16 Jc + col][k]
    // with the contraction index k = 16 J + crow(lane, v): the order in which the accumulators of z' = M'x leave the first product
    if (img2) {
#pragma unroll 1
      for (int Jc = 4 * half; Jc < 4 * half + 4; ++Jc)
        for (int J = Jc; J < 8; ++J) {
          const int f2 = (G::frag2_0(Jc) + (J - Jc)) * 4 + wave;  // v = wave
          img2[f2 * 64 + lane] = Xs[(16 * (Jc & 3) + (lane & 15)) * Cfg::LDX + 16 * J + Mfma<T>::crow(lane, wave)];
        }
    }
    IMG_T();
#ifdef BLR_IMG_STAMPS
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0)
      printf("image kernel, cycles: load %llu | 1/diag %llu | 16x16 inverses %llu | identity + sweep %llu | image out %llu\n", ist[1] - ist[0],
             ist[2] - ist[1], ist[3] - ist[2], ist[4] - ist[3], ist[5] - ist[4]);
#endif
  }
}

// ---- the stream ------------------------------------------------------------------------------------------------------------------
// ROWV: RowVecs inputs (N x D column-major): the same registers filled by scalar loads -- for a fixed d the sixteen inputs of a tile
// are consecutive (128 bytes in fp64), so the loads stay whole cache lines; twice (fp64) / four times (fp32) the load instructions.
template <typename T, bool ROWV = false>
__global__ __launch_bounds__(kThreads, 2) void marginals_gemm_kernel(MarginalArgs<T> a, const T* __restrict__ img_all) {
  using G = MargGemmCfg<T>;
  using acc4 = typename Mfma<T>::acc4;
  constexpr int VEC = G::VEC, NL = G::NLOAD;
  typedef T vecT __attribute__((ext_vector_type(Mfma<T>::VEC)));
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* const img = reinterpret_cast<T*>(smem);
  T* const mwl = reinterpret_cast<T*>(smem + G::OFF_MW);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int N = a.N;
  const int reg = a.reg0 + blockIdx.y;
  if (a.info && a.info[reg] != 0) return;
  const BLR_GLOBAL T* X = as_global(a.X + (int64_t)reg * a.strideX);
  const BLR_GLOBAL T* s = as_global(a.s + (int64_t)reg * a.strides);
  const BLR_GLOBAL T* mw = as_global(a.mw + (int64_t)reg * a.stridemw);
  const int ntiles = (N + 15) >> 4;
  const int t0 = blockIdx.x * kWaves + wave, tstep = gridDim.x * kWaves;
  const int g = lane >> 4, li = lane & 15;
  // this lane's slice of a tile: input n0 + li, entries d = 4 VEC m' + VEC g .. + VEC of it, m' = 0 .. NL - 1
  vecT av[NL], an[NL];
  // the noise variances of the inputs whose var this lane stores travel with the tile's loads (one value per call when the noise
  // is isotropic): loaded at the store they were a global-memory round trip at the END of every tile, with nothing left to hide it
  const bool diag_noise = a.noise_kind == NOISE_DIAGONAL;
  const T s_iso = diag_noise ? T(0) : s[0];
  // (the product is computed transposed, z' = M'x with the image as the A operand -- grad_gemm_kernel's trick: lane (li, g) then holds
  //  entries of ITS input's z, so the row sum needs two shuffles instead of sixteen DPP adds, and noise value and store are one
  //  coalesced access of the sixteen g = 0 lanes instead of four scattered ones)
  T sv, sn;
  auto fetch = [&](int tile, vecT (&dst)[NL], T& sd) {
    const int n = min(tile * 16 + li, N - 1);  // (inputs past the end re-read the last one; never stored)
    if constexpr (ROWV) {
      const BLR_GLOBAL T* p = X + n + (int64_t)(VEC * g) * a.ldx;
#pragma unroll
      for (int u = 0; u < NL; ++u)
#pragma unroll
        for (int e = 0; e < VEC; ++e) dst[u][e] = p[(int64_t)(4 * VEC * u + e) * a.ldx];
    } else {
      const BLR_GLOBAL vecT* p = reinterpret_cast<const BLR_GLOBAL vecT*>(X + (int64_t)n * a.ldx + VEC * g);
#pragma unroll
      for (int u = 0; u < NL; ++u) dst[u] = p[4 * u];  // (4 VEC elements = 4 vectors apart)
    }
    sd = (diag_noise && a.var) ? s[n] : s_iso;
  };
  if (t0 < ntiles) fetch(t0, av, sv);
  // the image and the prior mean: once per workgroup
  {
    const BLR_GLOBAL vecT* src = reinterpret_cast<const BLR_GLOBAL vecT*>(as_global(img_all + (int64_t)reg * G::IMG_ELEMS));
    vecT* dst = reinterpret_cast<vecT*>(img);
    for (int e = tid; e < G::IMG_ELEMS / VEC; e += kThreads) dst[e] = src[e];
    if (tid < kPB) mwl[tid] = mw[tid];
  }
  __syncthreads();
  for (int tile = t0; tile < ntiles; tile += tstep) {
    const bool more = tile + tstep < ntiles;
    if (more) fetch(tile + tstep, an, sn);  // in flight during this tile's 144 MFMAs
    // mean_n = x_n'mw (:33): this lane's 4 VEC NL / ... entries, then over the four lane groups of an input
    T macc = T(0);
    if (a.mean) {
#pragma unroll
      for (int u = 0; u < NL; ++u)
#pragma unroll
        for (int e = 0; e < VEC; ++e) macc += av[u][e] * mwl[4 * VEC * u + VEC * g + e];
      macc += __shfl_xor(macc, 16, 64);
      macc += __shfl_xor(macc, 32, 64);
    }
    // z = x'M column block by column block; var_n = |z_n|^2
    T sq = T(0);
    if (a.var) {
#pragma unroll
      for (int J = 0; J < 8; ++J) {
        acc4 acc = {T(0), T(0), T(0), T(0)};
        const T* fb = img + G::frag0(J) * 64 + lane;
#pragma unroll
        for (int m = 0; m < 4 * (J + 1); ++m) acc = Mfma<T>::mma(fb[m * 64], av[m / VEC][m % VEC], acc);
#pragma unroll
        for (int v = 0; v < 4; ++v) sq += acc[v] * acc[v];
      }
      sq += __shfl_xor(sq, 16, 64);
      sq += __shfl_xor(sq, 32, 64);
    }
    const int n0 = tile * 16;
    if (g == 0 && n0 + li < N) {
      if (a.mean) a.mean[(int64_t)reg * a.stridemean + n0 + li] = macc;
      if (a.var) a.var[(int64_t)reg * a.stridevar + n0 + li] = sq + sv;
    }
    if (more) {
#pragma unroll
      for (int u = 0; u < NL; ++u) av[u] = an[u];
      sv = sn;
    }
  }
}

// ---- the evidence gradient at D = 128 as two products with the same inverse (reference :55-58 through its reverse rule) ------------------
// logpdf_grad_kernel (blr_large.hpp) sweeps every tile of inputs forward (x'L^-T: the quadratic form) and backward (x'A^-1: dX)
// through LDS: two dependent chains of eight steps per tile, 0.25 of the matrix peak.  With M = L^-T explicit both are plain
// products, and the first one's accumulators ARE the second one's A operands if the first is computed transposed:
//   z' = M'x with the IMAGE as the A operand and the tile's registers as B: lane (n, g) ends up holding z_n[16 J + crow(g, v)] --
//   four contraction indices per lane group, exactly what an A fragment of  g = z M'  needs once the B fragments of M' are laid out in
//   that order (marg_image_kernel's second image).  G = X'A^-1 leaves in C layout: sixteen consecutive d per store.
// One workgroup of eight waves per CU (both images in LDS: 144 KB in fp64), a wave per 16-input tile, the next tile's loads in
// flight during the second product; dmw partials per lane, reduced once at the end in a fixed order.
template <typename T>
struct GradGemmCfg {
  using G = MargGemmCfg<T>;
#ifndef BLR_GG_WAVES64
#define BLR_GG_WAVES64 4
#endif
  // fp64: av, z and the dmw partials are 64 registers each -- four waves (512 registers per lane) instead of eight with 100+ spilled
  static constexpr int WAVES = sizeof(T) == 8 ? BLR_GG_WAVES64 : 8, THREADS = 64 * WAVES;
  static constexpr int OFF_IMG2 = G::IMG_ELEMS * (int)sizeof(T);
  static constexpr int OFF_MW = 2 * OFF_IMG2;
  static constexpr int OFF_RED = OFF_MW + kPB * (int)sizeof(T);          // [WAVES][128] doubles: dmw partials of the waves
  static constexpr int LDS_BYTES = OFF_RED + WAVES * kPB * 8;
};

template <typename T, bool ROWV = false>
__global__ __launch_bounds__(GradGemmCfg<T>::THREADS, 1) void grad_gemm_kernel(GradArgs<T> a, const T* __restrict__ img_all,
                                                                                const T* __restrict__ img2_all) {
  using G = MargGemmCfg<T>;
  using C = GradGemmCfg<T>;
  using acc4 = typename Mfma<T>::acc4;
  constexpr int VEC = G::VEC, NL = G::NLOAD;
  typedef T vecT __attribute__((ext_vector_type(Mfma<T>::VEC)));
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* const img = reinterpret_cast<T*>(smem);
  T* const img2 = reinterpret_cast<T*>(smem + C::OFF_IMG2);
  T* const mwl = reinterpret_cast<T*>(smem + C::OFF_MW);
  double* const red = reinterpret_cast<double*>(smem + C::OFF_RED);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int N = a.N;
  const int reg = a.reg0 + blockIdx.y;
  const bool ok = !(a.info && a.info[reg] != 0);
  T dacc[NL][VEC];  // this lane's share of dmw_d = sum_n x_dn w_n r_n: d = d_of(VEC u + e, g), inputs n = li (mod 16)
#pragma unroll
  for (int u = 0; u < NL; ++u)
#pragma unroll
    for (int e = 0; e < VEC; ++e) dacc[u][e] = T(0);
  const int g = lane >> 4, li = lane & 15;
  if (ok) {
    const BLR_GLOBAL T* X = as_global(a.X + (int64_t)reg * a.strideX);
    const BLR_GLOBAL T* s = as_global(a.s + (int64_t)reg * a.strides);
    const BLR_GLOBAL T* y = as_global(a.y + (int64_t)reg * a.stridey);
    const BLR_GLOBAL T* mwp = as_global(a.mwp + (int64_t)reg * a.stridemwp);
    T* const dXr = a.dX ? a.dX + (int64_t)reg * a.stridedX : nullptr;
    const int ntiles = (N + 15) >> 4;
    const int t0 = blockIdx.x * C::WAVES + wave, tstep = gridDim.x * C::WAVES;
    const bool diag_noise = a.noise_kind == NOISE_DIAGONAL;
    const T s_iso = diag_noise ? T(1) : s[0];
    vecT av[NL];  // (fp64: 64 registers -- and so are z and the dmw partials: the next tile's loads reuse av once the first product is done)
    T yv = T(0), svv = s_iso;
    auto fetch = [&](int tile, vecT (&dst)[NL], T& yd, T& sd) {
      const int n = min(tile * 16 + li, N - 1);  // (inputs past the end re-read the last one; never stored, weight 0)
      if constexpr (ROWV) {  // RowVecs: scalar loads, sixteen consecutive inputs per d (marginals_gemm_kernel)
        const BLR_GLOBAL T* p = X + n + (int64_t)(VEC * g) * a.ldx;
#pragma unroll
        for (int u = 0; u < NL; ++u)
#pragma unroll
          for (int e = 0; e < VEC; ++e) dst[u][e] = p[(int64_t)(4 * VEC * u + e) * a.ldx];
      } else {
        const BLR_GLOBAL vecT* p = reinterpret_cast<const BLR_GLOBAL vecT*>(X + (int64_t)n * a.ldx + VEC * g);
#pragma unroll
        for (int u = 0; u < NL; ++u) dst[u] = p[4 * u];
      }
      yd = y[n];
      sd = diag_noise ? s[n] : s_iso;
    };
    if (t0 < ntiles) fetch(t0, av, yv, svv);
    {
      const BLR_GLOBAL vecT* src = reinterpret_cast<const BLR_GLOBAL vecT*>(as_global(img_all + (int64_t)reg * G::IMG_ELEMS));
      const BLR_GLOBAL vecT* src2 = reinterpret_cast<const BLR_GLOBAL vecT*>(as_global(img2_all + (int64_t)reg * G::IMG_ELEMS));
      vecT* dst = reinterpret_cast<vecT*>(img);
      vecT* dst2 = reinterpret_cast<vecT*>(img2);
      for (int e = tid; e < G::IMG_ELEMS / VEC; e += C::THREADS) { dst[e] = src[e]; dst2[e] = src2[e]; }
      if (tid < kPB) mwl[tid] = mwp[tid];
    }
    __syncthreads();
    for (int tile = t0; tile < ntiles; tile += tstep) {
      const bool more = tile + tstep < ntiles;
      const int n0 = tile * 16;
      const bool valid = n0 + li < N;
      // posterior mean of the input and its residual (lane (li, g): the same value in all four lane groups after the reduction)
      T macc = T(0);
#pragma unroll
      for (int u = 0; u < NL; ++u)
#pragma unroll
        for (int e = 0; e < VEC; ++e) macc += av[u][e] * mwl[4 * VEC * u + VEC * g + e];
      macc += __shfl_xor(macc, 16, 64);
      macc += __shfl_xor(macc, 32, 64);
      const T w = valid ? T(1) / svv : T(0);
      const T rr = valid ? yv - macc : T(0);
      const T wr = w * rr;
#pragma unroll
      for (int u = 0; u < NL; ++u)
#pragma unroll
        for (int e = 0; e < VEC; ++e) dacc[u][e] += av[u][e] * wr;
      // first product, transposed: z[J][v] on lane (li, g) = z_n[16 J + crow(lane, v)], n = n0 + li
      acc4 z[8];
      T vq = T(0);
#pragma unroll
      for (int J = 0; J < 8; ++J) {
        acc4 acc = {T(0), T(0), T(0), T(0)};
        const T* fb = img + G::frag0(J) * 64 + lane;
#pragma unroll
        for (int m = 0; m < 4 * (J + 1); ++m) acc = Mfma<T>::mma(fb[m * 64], av[m / VEC][m % VEC], acc);
        z[J] = acc;
#pragma unroll
        for (int v = 0; v < 4; ++v) vq += acc[v] * acc[v];
        __builtin_amdgcn_sched_barrier(0);  // (without these the scheduler hoists the fragment reads of later blocks: 450 spilled registers)
      }
      vq += __shfl_xor(vq, 16, 64);
      vq += __shfl_xor(vq, 32, 64);  // x_n'A^-1 x_n
      if (g == 0 && valid) {
        if (a.dy) a.dy[(int64_t)reg * a.stridedy + n0 + li] = -wr;
        if (a.ds) a.ds[(int64_t)reg * a.strideds + n0 + li] = -(svv - rr * rr - vq) / (T(2) * svv * svv);
      }
      if (more) fetch(tile + tstep, av, yv, svv);  // in flight during the second product (av is dead since the first)
      // the rows this lane stores: n0 + crow(lane, v)
      T wrow[4], rrow[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int src = Mfma<T>::crow(lane, v);
        wrow[v] = __shfl(w, src, 64);
        rrow[v] = __shfl(rr, src, 64);
      }
      // second product: G[n][16 Jc + c] = sum_{k >= 16 Jc} z_n[k] M[16 Jc + c][k]
      __builtin_amdgcn_sched_barrier(0);
      if (dXr) {
#pragma unroll
        for (int Jc = 0; Jc < 8; ++Jc) {
          acc4 acc = {T(0), T(0), T(0), T(0)};
          const T* fb2 = img2 + G::frag2_0(Jc) * 4 * 64 + lane;
#pragma unroll
          for (int J = Jc; J < 8; ++J)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc = Mfma<T>::mma(z[J][v], fb2[((J - Jc) * 4 + v) * 64], acc);
          const T mwc = mwl[16 * Jc + li];
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int n = n0 + Mfma<T>::crow(lane, v);
            // (RowVecs: the four stores of a block fill sixteen 128-byte lines between them)
            if (n < N) dXr[ROWV ? (int64_t)(16 * Jc + li) * a.lddx + n : (int64_t)n * a.lddx + 16 * Jc + li] = wrow[v] * (rrow[v] * mwc - acc[v]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    // A^-1 = M M' (for dL/dLw) from the second image: its fragments (Jn, J, v) are at once the A operand "rows 16 Jn .. of M as
    // inputs" and the B operand of output block Jc -- 64 blocks of at most 32 MFMAs, dealt over the waves of the regressor's first
    // workgroup; stored through the symmetry so that sixteen lanes write consecutive addresses
    if (a.Ainv && blockIdx.x == 0) {
      T* const Ai = a.Ainv + (int64_t)reg * a.strideAi;
#pragma unroll 1
      for (int p = wave; p < 64; p += C::WAVES) {
        const int Jn = p >> 3, Jc = p & 7;
        acc4 acc = {T(0), T(0), T(0), T(0)};
        const T* fa = img2 + (G::frag2_0(Jn) - Jn) * 4 * 64 + lane;
        const T* fb = img2 + (G::frag2_0(Jc) - Jc) * 4 * 64 + lane;
#pragma unroll 1
        for (int J = max(Jn, Jc); J < 8; ++J)
#pragma unroll
          for (int v = 0; v < 4; ++v) acc = Mfma<T>::mma(fa[(J * 4 + v) * 64], fb[(J * 4 + v) * 64], acc);
#pragma unroll
        for (int v = 0; v < 4; ++v) Ai[(int64_t)(16 * Jn + Mfma<T>::crow(lane, v)) * a.ldai + 16 * Jc + li] = acc[v];
      }
    }
  }
  // dmw partial of this workgroup: lanes of a group hold the same d for 16 different inputs
  if (a.dmw_part) {
#pragma unroll
    for (int u = 0; u < NL; ++u)
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        double v = (double)dacc[u][e];
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
        if (li == 0) red[wave * kPB + 4 * VEC * u + VEC * g + e] = v;
      }
    __syncthreads();
    if (tid < kPB) {
      double acc = 0.0;
#pragma unroll
      for (int w8 = 0; w8 < C::WAVES; ++w8) acc += red[w8 * kPB + tid];
      a.dmw_part[((int64_t)reg * gridDim.x + blockIdx.x) * kPB + tid] = acc;
    }
  }
}

// =====================================================================================================================================
// D > 128: block forward substitution on an LDS-resident tile of inputs
// =====================================================================================================================================
// The tall-matrix route (mean_fill_kernel + one trsm_block_kernel + trailing gram_tile_kernel launch per 128-column panel) writes
// the inputs out as rows of a tall matrix and reads / rewrites the remaining columns once per panel: 4.15 GB of traffic for 277 MB of
// inputs at D = 1024, N = 65536 (fp32).  Rows are independent, so the whole substitution of a tile of inputs can stay on one CU:
//   tile of 32 inputs x D in LDS (128 KB at D = 1024, fp32), overwritten in place block by block:
//     Z_J = (X_J - sum_{K<J} Z_K L_JK') L_JJ^-T ,   L = U'
//   - the sum is a plain product: A operands = finished blocks Z_K from LDS, B operands = 16-byte loads straight from the caller's
//     U (L_JK'[d][j] = U[d + j ldu]: the contraction index is contiguous) -- 2 MB of L2-resident factor per tile, each fragment
//     feeding both 16-row halves of the tile;
//   - L_JJ^-T comes from marg_image_kernel (the D = 128 image, one per diagonal block), 144 MFMAs per 16 rows;
//   - var_n = |z_n|^2 + s_n from the accumulators, mean_n = x_n'mw from the tile: X is read ONCE, nothing is written but the outputs.
// Eight waves: in the product wave w owns column tile w of block J for both row halves; in the diagonal step (4 (j + 1) MFMAs for
// column tile j) the jobs are dealt (rows 0..15, tile w) + (rows 16..31, tile 7 - w): 36 MFMAs each.  Three barriers per block.
#ifdef BLR_MB_STAMPS
__device__ unsigned long long g_mbstamps[8][8];
#define MB_T0 unsigned long long mbt_prev = __builtin_amdgcn_s_memtime(), mbt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define MB_T(slot) do { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); mbt_acc[slot] += t__ - mbt_prev; mbt_prev = t__; } while (0)
#define MB_TFLUSH() do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) for (int q__ = 0; q__ < 8; ++q__) g_mbstamps[threadIdx.x >> 6][q__] = mbt_acc[q__]; } while (0)
#else
#define MB_T0 do {} while (0)
#define MB_T(slot) do {} while (0)
#define MB_TFLUSH() do {} while (0)
#endif
#if defined(BLR_MB_EXP) && (BLR_MB_EXP & 16)
#define MB_SYNC() do {} while (0)
#else
#define MB_SYNC() __syncthreads()
#endif
template <typename T, int RT_ = 32>
struct MargBlockCfg {
  // RT = 32: eight waves, one workgroup per CU -- in the product a wave owns one column tile for both 16-row halves (a factor
  // fragment feeds two MFMAs).  RT = 16: four waves, TWO workgroups per CU -- a wave owns two column tiles of the one row tile
  // (an input fragment feeds two MFMAs; twice the factor traffic per input), and one workgroup's diagonal step / hand-overs
  // run under the other's product.
  static constexpr int RT = RT_, WAVES = RT_ / 4, THREADS = 64 * WAVES, WGS = 32 / RT_;  // (WGS: workgroups per CU)
  static_assert(RT_ == 32 || RT_ == 16, "tile height");
  static constexpr int VEC = Mfma<T>::VEC;
  static constexpr int CH = 4 * VEC;               // contraction indices per 16-byte load of the four lane groups: 16 (f32) / 8 (f64)
  static constexpr int NCH = kPB / CH;             // such chunks per 128-block
  // row stride = 32 bytes mod 256: ds_read_b128 serves lanes {0-3, 12-15, 20-27} (rows 0-3, 12-15 of lane group g, rows 4-11 of
  // g + 1; MI355X_MICROARCH.md, LDS) in one cycle if their 16-byte slots differ: slot = 2 row + g -- evens and odds
  static constexpr int PAD = 32 / (int)sizeof(T);
  __host__ __device__ static constexpr int ld(int DP) { return DP + PAD; }
  __host__ __device__ static constexpr int off_red(int DP) { return RT * ld(DP) * (int)sizeof(T); }
  __host__ __device__ static constexpr int off_mw(int DP) { return off_red(DP) + WAVES * RT * (int)sizeof(double); }
  __host__ __device__ static constexpr int lds_bytes(int DP) { return off_mw(DP) + DP * (int)sizeof(T); }
  static constexpr int kMaxLds = 156 * 1024;
};

template <typename T>
struct MargBlockArgs {
  const T* X; int64_t ldx;  // ColVecs, 16-byte aligned columns
  const T* U; int64_t ldu;  // upper factor, column-major, 16-byte aligned columns
  const T* img;             // L_JJ^-T images of the diagonal blocks (MargGemmCfg<T>::IMG_ELEMS each)
  const T* mw; const T* s; int noise_kind;
  T* mean; T* var;
  const int32_t* info;
  int D, DP, N;  // D: order of the factor (a dense prior's padded Cholesky factor: DP)
  int Dx;        // features of an input (<= D; the tile is zero beyond)
  // blockIdx.y = regressor of a batch: element strides (img: DP / 128 images apart; info: one word each)
  int64_t strideX, strideU, stridemw, strides, stridemean, stridevar;
};

// 16-byte global loads the compiler does not see, and the waits that go with them.  hipcc retires in-order memory counters
// conservatively: at the head of a loop it waits for EVERYTHING (vmcnt(0)) -- here that is a wait for the inputs' next block from HBM
// at the start of every product.  Issued from inline asm the loads are invisible to its wait insertion; the waits are counted by hand
// (the counter retires in order: "at most N outstanding" = everything but the N youngest has arrived) and carry the loaded registers
// as operands, so no use can be scheduled ahead of them.
template <typename V>
__device__ __forceinline__ void mb_load16(V& dst, const BLR_GLOBAL void* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
template <typename V>
__device__ __forceinline__ void mb_load16_nt(V& dst, const BLR_GLOBAL void* p) {
  asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void mb_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <typename V, int K>
__device__ __forceinline__ void mb_pin(V (&r)[K]) {  // (after an mb_wait: uses of r stay behind it)
#pragma unroll
  for (int k = 0; k < K; ++k) asm volatile("" : "+v"(r[k]));
}

template <typename T, int RT>
__global__ __launch_bounds__((MargBlockCfg<T, RT>::THREADS), (MargBlockCfg<T, RT>::WGS)) void marg_blocksub_kernel(MargBlockArgs<T> a) {
  using C = MargBlockCfg<T, RT>;
  constexpr bool kTwoCols = RT == 16;  // the product's two accumulators: two column tiles of one row tile (else two row tiles of one)
  using G = MargGemmCfg<T>;
  using acc4 = typename Mfma<T>::acc4;
  constexpr int VEC = C::VEC, CH = C::CH, NCH = C::NCH;
  constexpr int RL = 8;                          // chunks of the factor per group (one wait per group): 128 (f32) / 64 (f64) rows

  constexpr int BV = kPB / VEC;                  // 16-byte vectors per row of a 128-column block: 32 (f32) / 64 (f64)
  constexpr int XV = C::RT * BV / C::THREADS;    // ... per thread: 2 / 4
  // loads per wave and phase, in issue order: R factor chunks for the next product | I image vectors for the next diagonal step |
  // XV input vectors for the block after next  (always that many: out-of-range ones re-read a valid address)
  constexpr int NI = 9 * (16 / CH);              // chunks of the two jobs of a wave together: (jA + 1) + (jB + 1) = 9 column-tile heights
  constexpr int R = kTwoCols ? 2 * RL : RL, I = NI;
  typedef T vecT __attribute__((ext_vector_type(Mfma<T>::VEC)));
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* const tile = reinterpret_cast<T*>(smem);
  double* const red = reinterpret_cast<double*>(smem + C::off_red(a.DP));
  T* const mws = reinterpret_cast<T*>(smem + C::off_mw(a.DP));  // the prior mean (no compiler-visible global load inside the phases)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  // A batch: workgroups are dealt round-robin over the 8 XCDs in dispatch order (x fastest), so a regressor's workgroups would
  // land on all eight and every L2 would hold every factor.  Renumbered so that an XCD gets whole regressors (bijective when
  // the grid is a multiple of 8; otherwise left as dispatched).
  int bid = blockIdx.x, breg = blockIdx.y;
  const int nbx = gridDim.x;
  if (gridDim.y > 1 && ((nbx * gridDim.y) & 7) == 0) {
    const int w = blockIdx.x + blockIdx.y * nbx, per = (nbx * (int)gridDim.y) >> 3;
    const int w2 = (w & 7) * per + (w >> 3);
    breg = w2 / nbx;
    bid = w2 - breg * nbx;
  }
  const int64_t reg = breg;
  if (a.info && a.info[reg] != 0) return;
  const int D = a.D, N = a.N, NC = a.DP / kPB, LD = C::ld(a.DP);
  const BLR_GLOBAL T* X = as_global(a.X) + reg * a.strideX;
  const BLR_GLOBAL T* U = as_global(a.U) + reg * a.strideU;
  const BLR_GLOBAL T* img = as_global(a.img) + reg * NC * G::IMG_ELEMS;
  const BLR_GLOBAL T* mw = as_global(a.mw) + reg * a.stridemw;
  const BLR_GLOBAL T* s = as_global(a.s) + reg * a.strides;
  T* const mean_out = a.mean ? a.mean + reg * a.stridemean : nullptr;
  T* const var_out = a.var + reg * a.stridevar;
  const int ntiles = (N + C::RT - 1) / C::RT;
  const int jA = wave, jB = 7 - wave;  // diagonal-step jobs: rows 0..15 x column tile jA, rows 16..31 x column tile jB
  // The inputs arrive one 128-column block at a time: thread -> row tid / 16 of the tile, vectors tid % 16 + 16 k of the block (a
  // thread keeps its row: mean_n = x_n'mw is accumulated from the same registers).  Requested at the end of a phase, stored at
  // the end of the next one; non-temporal: 256 MB of inputs must not push the 2 MB of factor out of the L2.
  const int xrow = tid >> 4, xv0 = tid & 15;
  vecT xbuf[XV];
  unsigned xvalid = 0;  // bit k: xbuf[k] holds data (not a re-read of something else)
  auto fetch_block = [&](int t, int Jb) {
    const bool tile_ok = t < ntiles;
    const int n = min((tile_ok ? t : 0) * C::RT + xrow, N - 1);  // (inputs past the end repeat the last one; never stored)
    const BLR_GLOBAL vecT* row = reinterpret_cast<const BLR_GLOBAL vecT*>(X + (int64_t)n * a.ldx);
    xvalid = 0;
#pragma unroll
    for (int k = 0; k < XV; ++k) {
      const int d0 = kPB * Jb + VEC * (xv0 + 16 * k);
      const bool ok = tile_ok && d0 < a.Dx;
      if (ok) xvalid |= 1u << k;
#if defined(BLR_MB_EXP) && (BLR_MB_EXP & 1)
      mb_load16_nt(xbuf[k], row);
#else
      mb_load16_nt(xbuf[k], row + (ok ? d0 / VEC : 0));
#endif
    }
  };
  double macc = 0.0;
  auto store_block = [&](int Jb) {  // (the caller has waited for xbuf)
    vecT* dst = reinterpret_cast<vecT*>(tile + xrow * LD + kPB * Jb);
#pragma unroll
    for (int k = 0; k < XV; ++k) {
      vecT v = xbuf[k];
      if (!((xvalid >> k) & 1u)) {
#pragma unroll
        for (int c = 0; c < VEC; ++c) v[c] = T(0);
      }
      dst[xv0 + 16 * k] = v;
      if (a.mean) {
        const int d0 = kPB * Jb + VEC * (xv0 + 16 * k);
        if (d0 < a.Dx) {
#pragma unroll
          for (int c = 0; c < VEC; ++c) macc += (double)v[c] * (double)mws[d0 + c];
        }
      }
    }
  };
  // fragments of L_JJ^-T for this wave's two jobs, one 16-byte load per chunk of R_J (the image's vector layout): job A's
  // nuA = (jA + 1) 16 / CH chunks first, then job B's NI - nuA  (a job whose column tile lies beyond D re-reads the image's
  // first vector: the count of loads stays NI)
  vecT f[NI];
  const int nuA = (jA + 1) * (16 / CH);
  auto fetch_image = [&](int J) {
    const int ncolt = min(8, (D - kPB * J) / 16);
    const BLR_GLOBAL vecT* im = reinterpret_cast<const BLR_GLOBAL vecT*>(img + (int64_t)J * G::IMG_ELEMS) + lane;
    const BLR_GLOBAL vecT* pa = jA < ncolt ? im + (G::frag0(jA) / VEC) * 64 : im;
    const BLR_GLOBAL vecT* pb = jB < ncolt ? im + (G::frag0(jB) / VEC) * 64 : im;
    const int sa = jA < ncolt ? 64 : 0, sb = jB < ncolt ? 64 : 0;
#pragma unroll
    for (int u = 0; u < NI; ++u) mb_load16(f[u], u < nuA ? pa + u * sa : pb + (u - nuA) * sb);
  };
  vecT b[RL], b2[kTwoCols ? RL : 1];
  auto fetch_factor = [&](const BLR_GLOBAL T* p, const BLR_GLOBAL T* p2) {
#pragma unroll
    for (int u = 0; u < RL; ++u) mb_load16(b[u], p + CH * u);
    if constexpr (kTwoCols) {
#pragma unroll
      for (int u = 0; u < RL; ++u) mb_load16(b2[u], p2 + CH * u);
    }
  };
  // rows of U' this wave multiplies with in block J: column tile `wave` (and, two-column form, 7 - wave; one that lies beyond D
  // re-reads the first: the count of loads stays R)
  auto factor_rows = [&](int J, int which) {
    const int nct = min(8, (D - kPB * J) / 16);
    const int ct = (which == 1 && 7 - wave < nct) ? 7 - wave : wave;
    return U + (int64_t)(kPB * J + 16 * ct + li) * a.ldu + VEC * g;
  };
  const int tstride = nbx;
  if (a.mean) {
    for (int e = tid; e < a.Dx; e += C::THREADS) mws[e] = mw[e];  // (visible after the first barrier of the tile loop)
    __syncthreads();
  }
  MB_T0;
  {
    fetch_block(bid, 0);
    mb_wait<0>();
    mb_pin(xbuf);
    store_block(0);
    fetch_image(0);              // (issue order of a phase: image, then inputs)
    fetch_block(bid, 1);  // (D > 128: at least two blocks)
  }
  for (int t = bid; t < ntiles; t += tstride) {
    const int n0 = t * C::RT;
    __syncthreads();  // block 0 of this tile is in place (stored at the end of the previous tile / above)
    // ---- z = L^-1 x block by block (:41-43) -------------------------------------------------------------------------------------------
    T sqA[4] = {T(0), T(0), T(0), T(0)}, sqB[4] = {T(0), T(0), T(0), T(0)};
#pragma unroll 1
    for (int J = 0; J < NC; ++J) {
      const int ncolt = min(8, (D - kPB * J) / 16);
      if (J > 0) {  // (the residual of block 0 is X_0 itself)
        acc4 c0 = {T(0), T(0), T(0), T(0)}, c1 = {T(0), T(0), T(0), T(0)};
        if (wave < ncolt) {
          const BLR_GLOBAL T* ub = factor_rows(J, 0);
          const BLR_GLOBAL T* ub2 = factor_rows(J, 1);
          const bool two = kTwoCols && 7 - wave < ncolt;  // (the second column tile exists in this block)
          const T* a0 = tile + li * LD + VEC * g;
          const T* a1 = kTwoCols ? a0 : a0 + 16 * LD;
          const int ngrp = J * (NCH / RL);
          // The factor arrives a GROUP of RL chunks at a time, one group ahead (group 0 was requested before the previous block's
          // diagonal step): one wait per group, at its start.  Group 0 lets the I + XV loads issued after its own stay in flight;
          // the later groups wait for everything -- their own loads are the youngest, and what was requested before them (the
          // inputs' block from HBM) has had a whole group of 64 MFMAs.  Inside a group one pinned stream per chunk: the NEXT
          // chunk's fragment reads, then the eight MFMAs of this chunk on operands that arrived during the previous chunk's.
          vecT x0 = *reinterpret_cast<const vecT*>(a0);
          vecT x1 = *reinterpret_cast<const vecT*>(a1);
          MB_T(7);
          mb_wait<I + XV>();
          MB_T(1);
#pragma unroll 1
          for (int grp = 0; grp < ngrp; ++grp) {
            const bool more = grp + 1 < ngrp;
            mb_pin(b);
            if constexpr (kTwoCols) mb_pin(b2);
            vecT bc[RL], bc2[kTwoCols ? RL : 1];
#pragma unroll
            for (int u = 0; u < RL; ++u) {
              bc[u] = b[u];
              if constexpr (kTwoCols) bc2[u] = b2[u];
            }
            mb_pin(bc);  // (copies made before the next group's loads land in b)
            if constexpr (kTwoCols) mb_pin(bc2);
#if !(defined(BLR_MB_EXP) && (BLR_MB_EXP & 4))
            if (more) fetch_factor(ub + CH * RL * (grp + 1), ub2 + CH * RL * (grp + 1));
#endif
#pragma unroll
            for (int u = 0; u < RL; ++u) {
              const int d = CH * (RL * grp + u);
              const int dn = (more || u + 1 < RL) ? d + CH : d;  // (the last chunk re-reads itself: no branch in the stream)
              const vecT nx0 = *reinterpret_cast<const vecT*>(a0 + dn);
              vecT nx1 = nx0;
              if constexpr (!kTwoCols) nx1 = *reinterpret_cast<const vecT*>(a1 + dn);
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int e = 0; e < VEC; ++e) {
                c0 = Mfma<T>::mma(x0[e], bc[u][e], c0);
                if constexpr (kTwoCols) c1 = Mfma<T>::mma(x0[e], bc2[u][e], c1);
                else c1 = Mfma<T>::mma(x1[e], bc[u][e], c1);
              }
              __builtin_amdgcn_sched_barrier(0);
              x0 = nx0;
              x1 = nx1;
            }
            MB_T(0);
#if !(defined(BLR_MB_EXP) && (BLR_MB_EXP & 8))
            if (more) mb_wait<0>();
#endif
            MB_T(1);
          }
#pragma unroll
          for (int v = 0; v < 4; ++v) {  // R_J = X_J - sum, in place (these columns belong to this wave alone)
            T* p = tile + Mfma<T>::crow(lane, v) * LD + kPB * J + 16 * wave + li;
            p[0] -= c0[v];
            if constexpr (kTwoCols) {
              if (two) p[16 * (7 - 2 * wave)] -= c1[v];  // (column tile 7 - wave of the same rows)
            } else {
              p[16 * LD] -= c1[v];
            }
          }
        }
        MB_T(2);
        MB_SYNC();
        MB_T(3);
      }
      // the next block's product starts on these (nothing here depends on the tile)
      const bool next_prod = J + 1 < NC && wave < min(8, (D - kPB * (J + 1)) / 16);
      if (next_prod) fetch_factor(factor_rows(J + 1, 0), factor_rows(J + 1, 1));
      // Z_J = R_J L_JJ^-T: job A = rows 0..15 x column tile jA, job B = rows 16..31 x column tile jB (two independent accumulators).
      // The fragments were requested in the previous phase; younger than them: XV input vectors, the R factor chunks above.
      MB_T(7);
      if (next_prod) mb_wait<XV + R>();
      else mb_wait<XV>();
      MB_T(4);
      mb_pin(f);
      acc4 zA = {T(0), T(0), T(0), T(0)}, zB = {T(0), T(0), T(0), T(0)};
      {
        const T* apA = tile + li * LD + kPB * J + VEC * g;
        const T* apB = tile + ((kTwoCols ? 0 : 16) + li) * LD + kPB * J + VEC * g;
        const bool doA = jA < ncolt, doB = jB < ncolt;
        vecT x = *reinterpret_cast<const vecT*>(apA);
#pragma unroll
        for (int u = 0; u < NI; ++u) {
          // (the next chunk's read goes out before this chunk's MFMAs; the list is A's chunks, then B's)
          const int un = u + 1 < NI ? u + 1 : u;
          const vecT nx = *reinterpret_cast<const vecT*>(un < nuA ? apA + CH * un : apB + CH * (un - nuA));
          if (u < nuA) {
            if (doA) {
#pragma unroll
              for (int e = 0; e < VEC; ++e) zA = Mfma<T>::mma(x[e], f[u][e], zA);
            }
          } else if (doB) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) zB = Mfma<T>::mma(x[e], f[u][e], zB);
          }
          x = nx;
        }
      }
      // (the MFMAs above have read f: pinned behind them through the accumulators)
      asm volatile("" : "+v"(zA), "+v"(zB));
#if !(defined(BLR_MB_EXP) && (BLR_MB_EXP & 2))
      fetch_image(J + 1 < NC ? J + 1 : 0);  // for the next diagonal step: a whole product away
#endif
      MB_T(5);
      MB_SYNC();  // everybody has read R_J
      MB_T(3);
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int r = Mfma<T>::crow(lane, v);
        if (jA < ncolt) {
          tile[r * LD + kPB * J + 16 * jA + li] = zA[v];
          sqA[v] += zA[v] * zA[v];
        }
        if (jB < ncolt) {
          tile[((kTwoCols ? 0 : 16) + r) * LD + kPB * J + 16 * jB + li] = zB[v];
          sqB[v] += zB[v] * zB[v];
        }
      }
      // the inputs' next block goes into its place (after the last block: block 0 of the next tile -- nobody reads Z_0 any more),
      // and the one after it is requested.  It was requested at the end of the previous phase; younger: R (if any) and I.
      if (J == NC - 1) {  // ... this tile's mean is complete: the 16 threads of a row
        if (a.mean) {
          const double m = row16_allreduce(macc);
          if (xv0 == 0 && n0 + xrow < N) mean_out[n0 + xrow] = (T)m;
        }
        macc = 0.0;
      }
      MB_T(7);
      if (next_prod) mb_wait<R + I>();
      else mb_wait<I>();
      MB_T(6);
      mb_pin(xbuf);
      store_block(J + 1 < NC ? J + 1 : 0);
      {
        const int Jn = J + 2;
        fetch_block(Jn < NC ? t : t + tstride, Jn < NC ? Jn : Jn - NC);
      }
      MB_T(7);
      if (J + 1 < NC) MB_SYNC();  // Z_J and X_(J+1) visible to the next block's product
      MB_T(3);
    }
    // ---- var_n = |z_n|^2 + s_n: over the 16 columns a lane group holds, then over the waves in wave order ------------------------------
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const double ra = (double)row16_allreduce(sqA[v]);
      const double rb = (double)row16_allreduce(sqB[v]);
      if (li == 0) {
        if constexpr (kTwoCols) {
          red[wave * C::RT + Mfma<T>::crow(lane, v)] = ra + rb;
        } else {
          red[wave * C::RT + Mfma<T>::crow(lane, v)] = ra;
          red[wave * C::RT + 16 + Mfma<T>::crow(lane, v)] = rb;
        }
      }
    }
    __syncthreads();
    if (tid < C::RT && n0 + tid < N) {
      double sum = 0.0;
#pragma unroll
      for (int w = 0; w < C::WAVES; ++w) sum += red[w * C::RT + tid];
      var_out[n0 + tid] = (T)sum + ((a.noise_kind == NOISE_DIAGONAL) ? s[n0 + tid] : s[0]);
    }
    MB_T(7);
  }
  MB_TFLUSH();
}

}  // namespace blr
