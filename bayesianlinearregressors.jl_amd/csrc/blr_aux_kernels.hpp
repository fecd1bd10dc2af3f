// Auxiliary kernels of the path: standalone small Cholesky, marginal stream, weight draws,
// projection of draws, fixed-order log-evidence sum.
#pragma once
#include "blr_common.hpp"
#include "blr_fused_small.hpp"

namespace blr {

// ---- U = chol(Lw).U for a dense D x D precision (D <= 128), one workgroup per regressor ----------
// reference :41 / :51 / sampling_functions.jl:29 call _cholesky(Lw) before every var / rand / draw.
template <typename T>
__global__ __launch_bounds__(kThreads) void chol_small_kernel(const T* __restrict__ Lw, int64_t ldl, int64_t strideLw,
                                                              T* __restrict__ U, int64_t ldu, int64_t strideU,
                                                              int32_t* __restrict__ info, int D, int B) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* const P = reinterpret_cast<T*>(smem);
  T* const dinv = P + D * (D + 1) / 2;
  const int tid = threadIdx.x;
  for (int reg = blockIdx.x; reg < B; reg += gridDim.x) {
    const T* A = Lw + (int64_t)reg * strideLw;
    T* out = U + (int64_t)reg * strideU;
    __syncthreads();
    for (int idx = tid; idx < D * D; idx += kThreads) {
      int i = idx / D, k = idx % D;
      if (k <= i) P[pidx(i, k)] = A[(int64_t)i * ldl + k];
    }
    int r = chol_packed(P, dinv, D, tid);
    if (tid == 0) info[reg] = r;
    if (r == 0) {
      for (int idx = tid; idx < D * D; idx += kThreads) {
        int c = idx / D, rr = idx % D;
        out[(int64_t)c * ldu + rr] = (rr <= c) ? P[pidx(c, rr)] : T(0);
      }
    }
  }
}

// ---- marginal stream --------------------------------------------------------------------------------
// mean_n = x_n'mw (:33), var_n = |U^-T x_n|^2 + s_n (:40-43): arguments of marginals_mfma_kernel (blr_large.hpp).
// prior_kind is UPPER_FACTOR (U D x D, upper) or DIAGONAL (d[D]); var == NULL: mean only.
template <typename T>
struct MarginalArgs {
  const T* X; int64_t ldx, strideX;
  const T* s; int64_t strides;
  const T* mw; int64_t stridemw;
  const T* U; int64_t ldu, strideU;
  T* mean; int64_t stridemean;
  T* var; int64_t stridevar;
  const int32_t* info;  // per-regressor status of a preceding factorisation (may be NULL)
  int layout, noise_kind, prior_kind;
  int D, N, B;
  int reg0;  // first regressor of this launch (grid.y is limited to 65535)
};

// ---- Y = X'W .+ sqrt.(s) .* Z2   (:52) ---------------------------------------------------------------
// 64 x 64 output tile per workgroup, 256 threads, X and W tiles staged in LDS in chunks of 32 rows of d.
template <typename T>
__global__ __launch_bounds__(kThreads) void rand_project_kernel(const T* __restrict__ X, int64_t ldx, int layout,
                                                                const T* __restrict__ W, int64_t ldw,
                                                                const T* __restrict__ s, int noise_kind,
                                                                const T* __restrict__ Z2, int64_t ldz2,
                                                                T* __restrict__ Y, int64_t ldy, int D, int N, int64_t S) {
  __shared__ T xs[32][65];
  __shared__ T ws[32][65];
  const int tid = threadIdx.x;
  const int n0 = blockIdx.x * 64;
  const int64_t s0 = (int64_t)blockIdx.y * 64;
  const int tn = tid & 63, ts = tid >> 6;  // thread computes n = n0+tn, samples s0 + ts + 4*j, j < 16
  T accv[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) accv[j] = T(0);
  // the next 32-row chunk of X and W is loaded into registers while the current one is consumed from LDS (the first
  // version waited out one global-memory latency per chunk: 32 of them at D = 1024)
  T px[8], pw[8];
  auto prefetch = [&](int d0) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = tid + u * kThreads;
      int dd, t;
      if (layout == LAYOUT_COLVECS) { dd = idx % 32; t = idx / 32; }
      else                          { t = idx % 64; dd = idx / 64; }
      const int d = d0 + dd, n = n0 + t;
      const bool okx = d < D && n < N;
      const T xv = X[okx ? ((layout == LAYOUT_COLVECS) ? (int64_t)n * ldx + d : (int64_t)d * ldx + n) : 0];
      px[u] = okx ? xv : T(0);
      const int dd2 = idx % 32, t2 = idx / 32;
      const int d2 = d0 + dd2;
      const int64_t sidx = s0 + t2;
      const bool okw = d2 < D && sidx < S;
      const T wv = W[okw ? sidx * ldw + d2 : 0];
      pw[u] = okw ? wv : T(0);
    }
  };
  prefetch(0);
  for (int d0 = 0; d0 < D; d0 += 32) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = tid + u * kThreads;
      int dd, t;
      if (layout == LAYOUT_COLVECS) { dd = idx % 32; t = idx / 32; }
      else                          { t = idx % 64; dd = idx / 64; }
      xs[dd][t] = px[u];
      ws[idx % 32][idx / 32] = pw[u];
    }
    if (d0 + 32 < D) prefetch(d0 + 32);
    __syncthreads();
#pragma unroll 4
    for (int dd = 0; dd < 32; ++dd) {
      T xv = xs[dd][tn];
#pragma unroll
      for (int j = 0; j < 16; ++j) accv[j] += xv * ws[dd][ts + 4 * j];
    }
  }
  const int n = n0 + tn;
  if (n < N) {
    const T sd = Z2 ? sqrt((noise_kind == NOISE_DIAGONAL) ? s[n] : s[0]) : T(0);  // Z2 == NULL: Y = X'W (blr_apply_weights_*)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      int64_t sidx = s0 + ts + 4 * j;
      if (sidx < S) Y[sidx * ldy + n] = Z2 ? accv[j] + sd * Z2[sidx * ldz2 + n] : accv[j];
    }
  }
}

// ---- the same projection on the matrix cores: ColVecs inputs with 16-byte aligned columns ---------------------------------
// Y (N x S) = X' W is a tall-skinny GEMM that reads X once: 128 inputs x 64 draws per workgroup, the contraction over d in
// chunks of KC rows staged in MFMA fragment order (A side: X' rows = inputs, B side: W columns = draws; both operands have the
// contraction index contiguous in memory, so one 16-byte load is VEC consecutive k values of one row and lands as VEC
// strided LDS words).  The next chunk's loads are in flight while the current one is multiplied; one barrier per chunk.
// The scalar-FMA kernel above stays for RowVecs / unaligned inputs.
template <typename T>
struct ProjCfg {
  static constexpr int TN = 128, TS = 64;
  static constexpr int VEC = Mfma<T>::VEC;
  static constexpr int KC = 128 / (int)sizeof(T);            // contraction rows per chunk: 32 (f32) / 16 (f64) = 8 vectors per row
  static constexpr int KST = KC / 4;                         // k-steps per chunk
  static constexpr int SIDE_A = KST * (TN / 16) * 64;        // elements
  static constexpr int SIDE_B = KST * (TS / 16) * 64;
  static constexpr int LDS_BYTES = 2 * (SIDE_A + SIDE_B) * (int)sizeof(T);  // 48 KB
  static constexpr int VA = TN * (KC / VEC) / kThreads;      // vectors per thread per chunk: 4 (A), 2 (B)
  static constexpr int VB = TS * (KC / VEC) / kThreads;
};

template <typename T>
__global__ __launch_bounds__(kThreads, 2) void rand_project_mfma_kernel(const T* __restrict__ X, int64_t ldx, const T* __restrict__ W,
                                                                        int64_t ldw, const T* __restrict__ s, int noise_kind,
                                                                        const T* __restrict__ Z2, int64_t ldz2, T* __restrict__ Y,
                                                                        int64_t ldy, int D, int N, int64_t S) {
  using Cf = ProjCfg<T>;
  using acc4 = typename Mfma<T>::acc4;
  constexpr int VEC = Cf::VEC, VPR = Cf::KC / VEC;  // vectors per row and chunk
  typedef T vecT __attribute__((ext_vector_type(Mfma<T>::VEC)));
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* const base = reinterpret_cast<T*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = blockIdx.x * Cf::TN;
  const int64_t s0 = (int64_t)blockIdx.y * Cf::TS;
  acc4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[i][k] = acc4{T(0), T(0), T(0), T(0)};
  // TWO chunks in flight per workgroup (register sets 0 / 1): 512 workgroups of 128 inputs are two per CU, and one 24 KB chunk each
  // kept 48 KB per CU in flight -- 2 TB/s of a stream that has nothing else to wait for
  vecT pa[2][Cf::VA], pb[2][Cf::VB];
  auto prefetch = [&](int d0, vecT (&qa)[Cf::VA], vecT (&qb)[Cf::VB]) {
#pragma unroll
    for (int u = 0; u < Cf::VA; ++u) {
      const int vi = u * kThreads + tid, dv = vi % VPR, nl = vi / VPR;
      const int d = d0 + dv * VEC, n = n0 + nl;
      const bool ok = d < D && n < N;  // D is a multiple of VEC on this path: a vector is inside or outside as a whole
      const vecT v = *reinterpret_cast<const vecT*>(X + (ok ? (int64_t)n * ldx + d : 0));
      qa[u] = ok ? v : vecT(T(0));
    }
#pragma unroll
    for (int u = 0; u < Cf::VB; ++u) {
      const int vi = u * kThreads + tid, dv = vi % VPR, sl = vi / VPR;
      const int d = d0 + dv * VEC;
      const int64_t sidx = s0 + sl;
      const bool ok = d < D && sidx < S;
      const vecT v = *reinterpret_cast<const vecT*>(W + (ok ? sidx * ldw + d : 0));
      qb[u] = ok ? v : vecT(T(0));
    }
  };
  // fragment image [k-step][16-row block][lane = (k & 3) * 16 + row % 16], the 64 words of k-step j rotated by 4 j: a thread holds
  // VEC consecutive k of ONE row, and the eight threads that share a row (one per k-step of the chunk) would all store to the
  // same bank (k-steps are 512 words apart): 8-way conflicts on every ds_write_b32; rotated, a group of 32 lanes covers 32 banks
  auto store = [&](T* slot, const vecT (&qa)[Cf::VA], const vecT (&qb)[Cf::VB]) {
    T* const A = slot;
    T* const Bm = slot + Cf::SIDE_A;
#pragma unroll
    for (int u = 0; u < Cf::VA; ++u) {
      const int vi = u * kThreads + tid, dv = vi % VPR, nl = vi / VPR;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const int k = dv * VEC + e;
        A[(((k >> 2) * (Cf::TN / 16) + (nl >> 4)) << 6) + ((((k & 3) << 4) + (nl & 15) + 4 * (k >> 2)) & 63)] = qa[u][e];
      }
    }
#pragma unroll
    for (int u = 0; u < Cf::VB; ++u) {
      const int vi = u * kThreads + tid, dv = vi % VPR, sl = vi / VPR;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const int k = dv * VEC + e;
        Bm[(((k >> 2) * (Cf::TS / 16) + (sl >> 4)) << 6) + ((((k & 3) << 4) + (sl & 15) + 4 * (k >> 2)) & 63)] = qb[u][e];
      }
    }
  };
  auto multiply = [&](const T* slot) {
#pragma unroll
    for (int j = 0; j < Cf::KST; ++j) {
      T fa[2], fb[4];
      const int rl = (lane + 4 * j) & 63;  // (the image of k-step j is rotated by 4 j lanes, above)
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = slot[((j * (Cf::TN / 16) + 2 * wave + i) << 6) + rl];
#pragma unroll
      for (int k = 0; k < 4; ++k) fb[k] = slot[Cf::SIDE_A + ((j * (Cf::TS / 16) + k) << 6) + rl];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[i][k] = Mfma<T>::mma(fa[i], fb[k], acc[i][k]);
    }
  };
  const int nchunks = (D + Cf::KC - 1) / Cf::KC;
  T* const slot0 = base;
  T* const slot1 = base + (Cf::SIDE_A + Cf::SIDE_B);
  prefetch(0, pa[0], pb[0]);
  store(slot0, pa[0], pb[0]);
  if (nchunks > 1) prefetch(Cf::KC, pa[1], pb[1]);
  __syncthreads();
  // chunk c sits in slot c & 1, chunk c + 1 is in flight in register set (c + 1) & 1; set c & 1 is free for chunk c + 2
  for (int c = 0; c < nchunks; c += 2) {
    if (c + 2 < nchunks) prefetch((c + 2) * Cf::KC, pa[0], pb[0]);
    multiply(slot0);
    if (c + 1 < nchunks) store(slot1, pa[1], pb[1]);
    __syncthreads();
    if (c + 1 >= nchunks) break;
    if (c + 3 < nchunks) prefetch((c + 3) * Cf::KC, pa[1], pb[1]);
    multiply(slot1);
    if (c + 2 < nchunks) store(slot0, pa[0], pb[0]);
    __syncthreads();
  }
  // Y[n, s] = acc + sqrt(s_n) z2[n, s]: lane (column = draw, 4 rows = inputs)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t sidx = s0 + 16 * k + (lane & 15);
      if (sidx >= S) continue;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int n = n0 + 16 * (2 * wave + i) + Mfma<T>::crow(lane, v);
        if (n < N) {
          if (Z2) {
            const T sd = sqrt((noise_kind == NOISE_DIAGONAL) ? s[n] : s[0]);
            Y[sidx * ldy + n] = acc[i][k][v] + sd * Z2[sidx * ldz2 + n];
          } else {
            Y[sidx * ldy + n] = acc[i][k][v];
          }
        }
      }
    }
}

// ---- random-Fourier feature map (BASELINE config 5): Phi[f, n] = scale * cos(sum_k Omega[k, f] x[k, n] + phase[f]) ----
// The reference ships no feature maps (a BasisFunctionRegressor takes any callable, basis_function_regression.jl:7-9);
// this is the phi of config 5.  Phi is written once (D x N, column-major) and then consumed by the plain path: the
// Gram is MFMA-bound by two orders of magnitude at D = 2048, so regenerating features per macro tile (17x the cos
// work) would cost more than the 2 x 128 MiB of extra traffic it saves.
// cos of the feature phase.  f32: the phase in revolutions, reduced to [-1/2, 1/2] by subtracting the nearest integer, then the
// hardware cosine (v_cos_f32 takes revolutions) -- the argument's own fp32 rounding (|phase| 1e-7) dominates the error either
// way, and the library cosine's generic range reduction made this kernel 4 % of config 5 (74 us for 33.5 M features).
__device__ __forceinline__ double rff_cos(double x) { return cos(x); }
__device__ __forceinline__ float rff_cos(float x) {
  const float rev = x * 0.15915494309189535f;  // 1 / (2 pi)
  return __builtin_amdgcn_cosf(rev - __builtin_rintf(rev));
}

template <typename T>
__global__ __launch_bounds__(kThreads) void rff_features_kernel(const T* __restrict__ Xin, int64_t ldxin,
                                                                const T* __restrict__ Omega, int64_t ldo,
                                                                const T* __restrict__ phase, T scale, int Din, int D, int N,
                                                                T* __restrict__ Phi, int64_t ldphi) {
  constexpr int NT = 32;   // columns per workgroup (one thread: one feature, NT columns)
  constexpr int KT = 16;   // input dimensions per LDS tile
  __shared__ T xs[KT][NT];  // the workgroup's tile of inputs: read once, then LDS broadcasts (the first version issued one
                            // wave-uniform global load per (k, column) and thread: a chain of 128 dependent-latency loads)
  const int f = blockIdx.x * kThreads + threadIdx.x;
  const int n0 = blockIdx.y * NT;
  const bool fin = f < D;
  T acc[NT];
  const T ph = fin ? phase[f] : T(0);
#pragma unroll
  for (int j = 0; j < NT; ++j) acc[j] = ph;
  for (int k0 = 0; k0 < Din; k0 += KT) {
    __syncthreads();
    for (int idx = threadIdx.x; idx < KT * NT; idx += kThreads) {
      const int kk = idx % KT, j = idx / KT;  // consecutive threads: consecutive k of one column (contiguous in memory)
      const int k = k0 + kk, n = n0 + j;
      xs[kk][j] = (k < Din && n < N) ? Xin[(int64_t)n * ldxin + k] : T(0);
    }
    __syncthreads();
    const int kend = min(KT, Din - k0);
    for (int kk = 0; kk < kend; ++kk) {
      const T om = fin ? Omega[(int64_t)f * ldo + k0 + kk] : T(0);
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] += om * xs[kk][j];
    }
  }
  if (!fin) return;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + j;
    if (n < N) Phi[(int64_t)n * ldphi + f] = scale * rff_cos(acc[j]);
  }
}

// ---- fixed-order sum of logpdf[B] --------------------------------------------------------
// One workgroup.  Thread t sums elements t, t+256, ... in order, then a fixed tree over threads: the
// result depends only on B and the values, never on launch geometry.
__global__ __launch_bounds__(kThreads) void logpdf_sum_kernel(const double* __restrict__ lp, int64_t B,
                                                              double* __restrict__ total) {
  __shared__ double part[kThreads];
  const int tid = threadIdx.x;
  double v = 0.0;
  int64_t i = tid;
  for (; i + 7 * kThreads < B; i += 8 * kThreads) {  // eight loads in flight, added in the order of the plain loop (same bits)
    double u[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) u[k] = lp[i + (int64_t)k * kThreads];
#pragma unroll
    for (int k = 0; k < 8; ++k) v += u[k];
  }
  for (; i < B; i += kThreads) v += lp[i];
  part[tid] = v;
  __syncthreads();
  for (int m = kThreads / 2; m >= 1; m >>= 1) {
    if (tid < m) part[tid] += part[tid + m];
    __syncthreads();
  }
  if (tid == 0) *total = part[0];
}

}  // namespace blr
