// Dense observation-noise covariance and full predictive covariance.
//
//   reference src/bayesian_linear_regression.jl:79-82   _cholesky(Sigma_y) general branch: Bt = Us' \ (Uw' \ X)', dy = Us' \ (y - X'mw)
//   reference :35-38, :45                               cov(fx) = alpha' alpha + Sigma_y,  alpha = Uw' \ X;  mean_and_cov
//   reference :52                                       rand: ... + Us' * randn(N, S)
//
// Nothing here is a new numerical kernel: a dense Sigma_y is WHITENED AWAY with the machinery of the large-D path --
//   [ Sigma_y ]  blocked Cholesky (chol_large)  [ L          ]        L L' = Sigma_y  (Us = L')
//   [ X       ]  with the rows below carried    [ X L^-T     ]  =     X Us^-1
//   [ y'      ]  through TRSM + trailing update [ (L^-1 y)'  ]  =     (Us^-T y)'
// after which the whitened problem has unit diagonal noise and goes through the ordinary fused / large-D update; the
// evidence gets -logdet(Sigma_y)/2 on top.  cov(fx) = Y Y' + Sigma_y with Y = X' Lw^-T produced by the tall-matrix panel
// sweep of the marginal path and Y Y' by gram_tile_kernel (the operand's roles swapped: N "rows", D "observations").
#pragma once
#include "blr_large.hpp"

namespace blr {

// rows [row0, row0 + D) of M <- X (either layout), row row0 + D <- y, the remaining R - D - 1 rows and every column n >= N zero
template <typename T>
__global__ __launch_bounds__(kThreads) void whiten_fill_kernel(const T* __restrict__ X, int64_t ldx, int layout, const T* __restrict__ y,
                                                               int D, int N, int NP, int R, T* __restrict__ M, int64_t ld, int row0) {
  const int64_t total = (int64_t)R * NP;
  for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += (int64_t)gridDim.x * kThreads) {
    const int r = (int)(e % R), n = (int)(e / R);
    T v = T(0);
    if (n < N) {
      if (r < D) v = (layout == LAYOUT_COLVECS) ? X[(int64_t)n * ldx + r] : X[(int64_t)r * ldx + n];
      else if (r == D) v = y[n];
    }
    M[(int64_t)n * ld + row0 + r] = v;
  }
}

// out[n] = M[row + n * ld]
template <typename T>
__global__ __launch_bounds__(kThreads) void row_extract_kernel(const T* __restrict__ M, int64_t ld, int row, int N, T* __restrict__ out) {
  for (int n = blockIdx.x * kThreads + threadIdx.x; n < N; n += gridDim.x * kThreads) out[n] = M[(int64_t)n * ld + row];
}

// evidence of the whitened problem -> evidence of the original one; a Sigma_y that is not positive definite wins over whatever
// the inner update reported on the garbage it was given (reference :79 throws before :86 is reached)
__global__ void dense_finish_kernel(double* logpdf, int32_t* info, const double* logdet_Sy, const int32_t* noise_info,
                                    const int32_t* prior_info) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (*prior_info != 0) {  // reference :78 comes before :79
    *info = *prior_info;
    if (logpdf) *logpdf = __longlong_as_double(0x7ff8000000000000LL);
  } else if (*noise_info != 0) {
    *info = *noise_info;
    if (logpdf) *logpdf = __longlong_as_double(0x7ff8000000000000LL);
  } else if (logpdf && *info == 0) {
    *logpdf -= 0.5 * *logdet_Sy;
  }
}

// Y[n + d * ldy] = X(d, n) / sqrt(dprior[d])  (diagonal prior precision: alpha' = X' Uw^-1 is a column scaling); zero padding
template <typename T>
__global__ __launch_bounds__(kThreads) void diag_prior_rows_kernel(const T* __restrict__ X, int64_t ldx, int layout, const T* __restrict__ dprior,
                                                                   int D, int N, int NP, int DPc, T* __restrict__ Y, int64_t ldy) {
  const int64_t total = (int64_t)NP * DPc;
  for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += (int64_t)gridDim.x * kThreads) {
    const int n = (int)(e % NP), d = (int)(e / NP);
    T v = T(0);
    if (n < N && d < D) {
      const T x = (layout == LAYOUT_COLVECS) ? X[(int64_t)n * ldx + d] : X[(int64_t)d * ldx + n];
      v = x / sqrt(dprior[d]);
    }
    Y[(int64_t)d * ldy + n] = v;
  }
}

// C = (lower-triangular 128 x 128 tiles of Y Y') + Sigma_y, written as the full symmetric N x N matrix (any ldc)
// noise_kind 0: s[0] on the diagonal; 1: s[n]; 2: dense N x N (upper triangle read, lds)
template <typename T>
__global__ __launch_bounds__(kThreads) void cov_assemble_kernel(const T* __restrict__ Gpart, int ntiles, int N, int noise_kind,
                                                                const T* __restrict__ s, int64_t lds, T* __restrict__ C, int64_t ldc) {
  const int t = blockIdx.x;
  if (t >= ntiles) return;
  int I = 0;
  while ((I + 1) * (I + 2) / 2 <= t) ++I;
  const int J = t - I * (I + 1) / 2;
  constexpr int kChunk = kPB * kPB / 16;
  const T* tile = Gpart + (int64_t)t * (kPB * kPB);
  for (int e = blockIdx.y * kChunk + threadIdx.x; e < (int)(blockIdx.y + 1) * kChunk; e += kThreads) {
    const int rl = e % kPB, cl = e / kPB;
    const int row = I * kPB + rl, col = J * kPB + cl;
    if (row >= N || col >= N || col > row) continue;
    T v = tile[e];
    if (noise_kind == 2) v += s[(int64_t)row * lds + col];  // upper entry (col, row)
    else if (row == col) v += (noise_kind == NOISE_DIAGONAL) ? s[row] : s[0];
    C[(int64_t)col * ldc + row] = v;
    C[(int64_t)row * ldc + col] = v;
  }
}

// Y[n + j * ldy] += sum_{m <= n} L[n + m * ld] Z2[m + j * ldz2]   (Us' Z2 with Us' = L, reference :52); 64 x 16 tile per workgroup
template <typename T>
__global__ __launch_bounds__(kThreads) void lower_mult_add_kernel(const T* __restrict__ L, int64_t ld, int N, const T* __restrict__ Z2,
                                                                  int64_t ldz2, T* __restrict__ Y, int64_t ldy, int64_t S) {
  __shared__ T zs[64][17];
  const int tn = threadIdx.x & 63, tj = threadIdx.x >> 6;  // 64 rows x 4 column groups of 4
  const int n = blockIdx.x * 64 + tn;
  const int64_t j0 = (int64_t)blockIdx.y * 16;
  T acc[4] = {T(0), T(0), T(0), T(0)};
  const int mend = min(N, (int)(blockIdx.x + 1) * 64);  // rows of this tile need m <= n < mend
  for (int m0 = 0; m0 < mend; m0 += 64) {
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * 16; e += kThreads) {
      const int mm = e & 63, jj = e >> 6;
      const int m = m0 + mm;
      zs[mm][jj] = (m < N && j0 + jj < S) ? Z2[(j0 + jj) * ldz2 + m] : T(0);
    }
    __syncthreads();
    if (n < N) {
      const int lim = min(64, n - m0 + 1);
      for (int mm = 0; mm < lim; ++mm) {
        const T l = L[(int64_t)(m0 + mm) * ld + n];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] += l * zs[mm][4 * tj + u];
      }
    }
  }
  if (n < N) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t j = j0 + 4 * tj + u;
      if (j < S) Y[j * ldy + n] += acc[u];
    }
  }
}

}  // namespace blr
