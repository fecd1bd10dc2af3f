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
// were inverted explicitly already.  Routed for D = 128, aligned ColVecs; everything else stays on the sweep kernel.
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
};

// ---- M = L^-T in B-fragment order, two workgroups per regressor (blockIdx.y: rows 0..63 / 64..127 of M) ----------------------------
template <typename T>
__global__ __launch_bounds__(kThreads) void marg_image_kernel(const T* __restrict__ U, int64_t ldu, int64_t strideU, int D, T* __restrict__ img,
                                                              const int32_t* __restrict__ info, int reg0) {
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
  if (info && info[reg] != 0) return;
  U += (int64_t)reg * strideU;
  img += (int64_t)reg * G::IMG_ELEMS;
  const int nchunks = kPB / 16;
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
  if (tid < kPB) dinv[tid] = T(1) / P[pidx(tid, tid)];
  __syncthreads();
  trsm_prepare<T>(P, dinv, Linv, nchunks, tid);
  {
    const int half = blockIdx.y;
    __syncthreads();
    for (int e = tid; e < Cfg::RB * kPB; e += kThreads) {
      const int r = e / kPB, c = e % kPB;
      Xs[r * Cfg::LDX + c] = (c == 64 * half + r) ? T(1) : T(0);
    }
    __syncthreads();
    trsm_sweep<T>(Xs, P, Linv, nchunks, lane, wave);  // rows 64 half .. + 63 of L^-T (ends with a barrier)
    // image entries whose contraction index d falls into this half
    for (int e = tid; e < G::IMG_ELEMS; e += kThreads) {
      const int f = e >> 6, l = e & 63;
      int J = 0;
      while (G::frag0(J + 1) <= f) ++J;
      const int m = f - G::frag0(J);
      const int d = G::d_of(m, l >> 4);
      if ((d >> 6) == half) img[e] = Xs[(d & 63) * Cfg::LDX + 16 * J + (l & 15)];
    }
  }
}

// ---- the stream ------------------------------------------------------------------------------------------------------------------
template <typename T>
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
  auto fetch = [&](int tile, vecT (&dst)[NL]) {
    const int n = min(tile * 16 + li, N - 1);  // (inputs past the end re-read the last one; never stored)
    const BLR_GLOBAL vecT* p = reinterpret_cast<const BLR_GLOBAL vecT*>(X + (int64_t)n * a.ldx + VEC * g);
#pragma unroll
    for (int u = 0; u < NL; ++u) dst[u] = p[4 * u];  // (4 VEC elements = 4 vectors apart)
  };
  if (t0 < ntiles) fetch(t0, av);
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
    if (more) fetch(tile + tstep, an);  // in flight during this tile's 144 MFMAs
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
    T sq[4] = {T(0), T(0), T(0), T(0)};
    if (a.var) {
#pragma unroll
      for (int J = 0; J < 8; ++J) {
        acc4 acc = {T(0), T(0), T(0), T(0)};
        const T* fb = img + G::frag0(J) * 64 + lane;
#pragma unroll
        for (int m = 0; m < 4 * (J + 1); ++m) acc = Mfma<T>::mma(av[m / VEC][m % VEC], fb[m * 64], acc);
#pragma unroll
        for (int v = 0; v < 4; ++v) sq[v] += acc[v] * acc[v];
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) sq[v] = row16_allreduce(sq[v]);  // over the 16 columns of the block a lane group holds
    }
    const int n0 = tile * 16;
    if (a.mean && g == 0 && n0 + li < N) a.mean[(int64_t)reg * a.stridemean + n0 + li] = macc;
    if (a.var && li == 0) {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int n = n0 + Mfma<T>::crow(lane, v);
        if (n < N) a.var[(int64_t)reg * a.stridevar + n] = sq[v] + ((a.noise_kind == NOISE_DIAGONAL) ? s[n] : s[0]);
      }
    }
    if (more) {
#pragma unroll
      for (int u = 0; u < NL; ++u) av[u] = an[u];
    }
  }
}

}  // namespace blr
