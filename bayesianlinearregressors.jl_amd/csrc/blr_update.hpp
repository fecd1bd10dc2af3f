// Rank-k update of a RESIDENT posterior state.
//
//   reference test/bayesian_linear_regression.jl:49-70   "repeated conditioning": posterior(f'1(X2, S2), y2)
//   reference src/bayesian_linear_regression.jl:93        the posterior carries (mw', Lw') forward; :72-89 then re-derives
//                                                         everything from Lw' at O(D^3) per call
//
// State: posterior mean m [D] and the upper factor T [D x D] of the precision (A = T'T), both device-resident.
// k new observations (x_i, y_i, s_i) arrive.  With w_i = x_i / sqrt(s_i), e_i = (y_i - x_i'm) / sqrt(s_i) the update is the
// least-squares problem   min_d |T d|^2 + sum_i (w_i'd - e_i)^2,   m' = m + d,   solved the square-root-information way:
// each new row [w_i | e_i] is rotated into [T | u] (u starts at 0) by D Givens rotations,
//     [ T | u ]      [ T' | u' ]
//     [ w | e ]  ->  [ 0  | rho ] ,      2 D^2 flops per observation -- O(k D^2), no D^3 term, no Gram matrix,
// after which  T' is the new factor (T''T' = T'T + sum w_i w_i'),  d = T'^-1 u',  and the evidence of the batch under the
// OLD state is   log p(y) = -1/2 [ k log 2pi + sum log s_i + 2 sum_j log(T'_jj / T_jj) + sum_i rho_i^2 ]
// (rho_i^2 are the squared one-step-ahead standardised residuals; the middle term is logdet A' - logdet A).
// Orthogonal transformations only: T' stays a valid factor however ill-conditioned A is (no downdating anywhere).
//
// One workgroup per regressor (four waves load / store, wave 0 sweeps); T lives in LDS as packed upper rows with the u column appended (row j: columns j..D), the
// rotation of step j is a readlane + 2 FMAs per owned column.  The sweep is a serial chain of D dependent rotations per
// observation (measured ~40 us per observation at D = 128: readlane -> a^2 + b^2 -> rsqrt -> c, s -> FMA), so it wins for a
// single new observation only; blr_update_factor_* routes everything else to the in-place re-factorisation of the same state
// (PRIOR_UPPER_FACTOR), whose cost does not depend on k.  Measured crossover: DESIGN.md K10, tools/update_bench.py.
#pragma once
#include "blr_common.hpp"

namespace blr {

constexpr int kSweepMaxD = 128;
constexpr int kSweepMaxK = 16;

template <typename T>
struct SweepArgs {
  const T* X; int64_t ldx, strideX; int layout;
  const T* y; int64_t stridey;
  const T* s; int64_t strides; int noise_kind;
  T* mw; int64_t stridemw;          // in/out
  T* Tf; int64_t ldt, strideT;      // in/out, upper factor (strictly-lower part is neither read nor written)
  double* logpdf; int32_t* info;
  int D, k;
};

template <typename T>
__host__ __device__ constexpr int sweep_lds_bytes(int D) { return ((D + 1) * (D + 2) / 2 + 2 * (D + 1)) * (int)sizeof(T) + 64; }  // rows + m + d + flags

template <typename T>
__global__ __launch_bounds__(kThreads) void rank1_sweep_kernel(SweepArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int D = a.D, tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int64_t reg = blockIdx.x;
  T* const R = reinterpret_cast<T*>(smem);                // packed rows: row j at off(j), columns j..D (column D = u_j)
  T* const mv = R + (D + 1) * (D + 2) / 2;                // m [D]
  T* const dv = mv + (D + 1);                             // d [D]
  int* const iscr = reinterpret_cast<int*>(dv + (D + 1)); // flags
  auto off = [&](int j) { return j * (D + 1) - (j * (j - 1)) / 2 - j; };  // R[off(j) + col] = element (j, col)
  const T* Tg = a.Tf + reg * a.strideT;
  T* mg = a.mw + reg * a.stridemw;
  // ---- load (all four waves): the D x D block column by column, every load independent of the others
  if (tid == 0) iscr[0] = 0x7fffffff;
  __syncthreads();
  for (int e = tid; e < D * D; e += kThreads) {
    const int c = e / D, j = e - c * D;
    if (j <= c) {
      const T v = Tg[(int64_t)c * a.ldt + j];
      R[off(j) + c] = v;
      if (j == c && !(v > T(0))) atomicMin(&iscr[0], c + 1);  // first non-positive diagonal entry of the factor
    }
  }
  for (int j = tid; j < D; j += kThreads) { R[off(j) + D] = T(0); mv[j] = mg[j]; }
  __syncthreads();
  const bool bad_factor = iscr[0] != 0x7fffffff;
  int bad_noise = 0;
  double logpdf = 0.0;
  if (wave == 0) {
    double logd0 = 0.0;
    for (int j = lane; j < D; j += 64) logd0 += log((double)R[off(j) + j]);
    logd0 = wave_allreduce(logd0);
    const T* Xg = a.X + reg * a.strideX;
    const T* yg = a.y + reg * a.stridey;
    const T* sg = a.s + reg * a.strides;
    const bool diag = a.noise_kind == NOISE_DIAGONAL;
    const int c0 = lane, c1 = lane + 64;
    const bool has0 = c0 < D, has1 = c1 < D;
    double quad = 0.0, logs = 0.0;
    for (int i = 0; i < a.k; ++i) {
      // the new row, scaled: the lane owns columns lane, lane + 64 and (lane 0 only) column D = the rhs entry e
      const T si = diag ? sg[i] : sg[0];
      if (!(si > T(0)) && bad_noise == 0) bad_noise = i + 1;  // reference :79: cholesky(Sigma_y) throws
      const T rs = fast_rsqrt(si > T(0) ? si : T(1));
      logs += log((double)si);
      T w0 = T(0), w1 = T(0), w2 = T(0);
      if (has0) w0 = (a.layout == LAYOUT_COLVECS) ? Xg[(int64_t)i * a.ldx + c0] : Xg[(int64_t)c0 * a.ldx + i];
      if (has1) w1 = (a.layout == LAYOUT_COLVECS) ? Xg[(int64_t)i * a.ldx + c1] : Xg[(int64_t)c1 * a.ldx + i];
      double mu = (double)w0 * (double)(has0 ? mv[c0] : T(0)) + (double)w1 * (double)(has1 ? mv[c1] : T(0));
      mu = wave_allreduce(mu);
      w0 *= rs; w1 *= rs;
      if (lane == 0) w2 = (T)(((double)yg[i] - mu) * (double)rs);
      // D rotations.  The serial chain of a step is readlane -> a^2 + b^2 -> rsqrt -> c, s -> the owner lane's w; the row
      // itself does not depend on the previous step, so its elements are fetched one step ahead.
      const T* rp = R;
      T t0 = has0 ? rp[c0] : T(0), t1 = has1 ? rp[c1] : T(0), t2 = rp[D], aj = rp[0];
      for (int j = 0; j < D; ++j) {
        T* rw = R + off(j);
        const T* rn = R + off(j + 1 < D ? j + 1 : j);
        const int jn = j + 1 < D ? j + 1 : j;
        const T n0 = (has0 && c0 >= jn) ? rn[c0] : T(0), n1 = (has1 && c1 >= jn) ? rn[c1] : T(0), n2 = rn[D], an = rn[jn];
        const T bj = (j < 64) ? readlane(w0, j) : readlane(w1, j - 64);
        const T h2 = aj * aj + bj * bj;
        const T ri = fast_rsqrt(h2);
        const T cs = aj * ri, sn = bj * ri;
        if (has0 && c0 >= j) { rw[c0] = (c0 == j) ? h2 * ri : cs * t0 + sn * w0; w0 = cs * w0 - sn * t0; }
        if (has1 && c1 >= j) { rw[c1] = (c1 == j) ? h2 * ri : cs * t1 + sn * w1; w1 = cs * w1 - sn * t1; }
        if (lane == 0) { rw[D] = cs * t2 + sn * w2; w2 = cs * w2 - sn * t2; }
        t0 = n0; t1 = n1; t2 = n2; aj = an;
      }
      const T rho = readlane(w2, 0);
      quad += (double)rho * (double)rho;
    }
    // ---- d = T'^-1 u, column-oriented (no reductions): d_j = u_j / T_jj, then u_c -= T_cj d_j for c < j
    T u0 = has0 ? R[off(c0) + D] : T(0), u1 = has1 ? R[off(c1) + D] : T(0);
    T r0 = has0 ? fast_rcp(R[off(c0) + c0]) : T(0), r1 = has1 ? fast_rcp(R[off(c1) + c1]) : T(0);
    double logd1 = (has0 ? log((double)R[off(c0) + c0]) : 0.0) + (has1 ? log((double)R[off(c1) + c1]) : 0.0);
    logd1 = wave_allreduce(logd1);
    {
      T e0 = (has0 && c0 < D - 1) ? R[off(c0) + D - 1] : T(0), e1 = (has1 && c1 < D - 1) ? R[off(c1) + D - 1] : T(0);
      for (int j = D - 1; j >= 0; --j) {
        const int jn = j > 0 ? j - 1 : 0;
        const T f0 = (has0 && c0 < jn) ? R[off(c0) + jn] : T(0), f1 = (has1 && c1 < jn) ? R[off(c1) + jn] : T(0);
        const T dj = (j < 64) ? readlane(u0, j) * readlane(r0, j) : readlane(u1, j - 64) * readlane(r1, j - 64);
        if (lane == (j & 63)) { if (j < 64) u0 = dj; else u1 = dj; }
        if (c0 < j) u0 -= e0 * dj;
        if (c1 < j) u1 -= e1 * dj;
        e0 = f0; e1 = f1;
      }
    }
    if (has0) dv[c0] = u0;
    if (has1) dv[c1] = u1;
    const double kLog2Pi = 1.8378770664093454835606594728112;
    logpdf = -0.5 * ((double)a.k * kLog2Pi + logs + 2.0 * (logd1 - logd0) + quad);
    if (lane == 0) iscr[1] = bad_noise;
  }
  __syncthreads();
  // ---- write the state back (all four waves); untouched when anything failed
  const int bad = bad_factor ? iscr[0] : iscr[1];  // the same code the re-factorisation route reports for a bad factor
  if (bad == 0) {
    for (int e = tid; e < D * D; e += kThreads) {
      const int c = e / D, j = e - c * D;
      if (j <= c) a.Tf[reg * a.strideT + (int64_t)c * a.ldt + j] = R[off(j) + c];
    }
    for (int j = tid; j < D; j += kThreads) mg[j] = mv[j] + dv[j];
  }
  if (tid == 0) {
    a.info[reg] = bad;
    if (a.logpdf) a.logpdf[reg] = bad ? __longlong_as_double(0x7ff8000000000000LL) : logpdf;
  }
}

}  // namespace blr
