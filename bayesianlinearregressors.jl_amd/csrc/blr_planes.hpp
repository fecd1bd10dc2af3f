// fp32 Gram matrices at D > 128 (configs 3 and 5) from PRE-SPLIT operands: the low-precision planes of the design matrix are made ONCE
// per element, in the fragment order of the matrix instruction, and the Gram launch is nothing but LDS-DMA, fragment reads and
// v_mfma_f32_32x32x16_{f16, bf16}.
//
// Why (round 5, gram_tile_kernel<float, true>): an fp32 number is exactly three bf16 numbers, and the six products hh, hm, mh, mm, hl,
// lh under fp32 accumulation are as accurate as an fp32 fma chain (tools/bf3_unit.hip) at 14 x the rate of v_mfma_f32_16x16x4_f32 --
// but that kernel split its operands inside the matrix loop, once per macro tile that touches an element (8 x at config 3, 16 x at
// config 5): ~ 290 vector instructions per 16 columns and wave next to 24 matrix instructions, matrix pipe 37 % busy.  With the split
// moved out of the loop (NP = 3 below) the matrix pipe IS the time -- and, on random operands, at the clock the part then holds it is no
// faster than before (config 3: 374 + 118 us for the Gram launch + the planes pass against 522; same-box end to end 0.79 / 0.80 ms).
// What buys time is fewer products.  NP = 2, the default:
//   * every row r gets a power-of-two scale 2^e_r that puts (a bound of) its largest entry of z = x sqrt(w) into [2^12, 2^13) -- the
//     bound is twice the largest entry of a SAMPLE of the row (rowmax_kernel: the first 32 columns of every column chunk; the planes
//     pass checks that every entry fits and asks for the exact maxima + a second planes pass when one does not: PlanesArgs::redo; a
//     random-Fourier basis is bounded by its own scale factor);  z' = z 2^-e_r = h + l + rho with
//     h = fp16(z'), l = fp16(z' - h), both rounded to nearest: 22 significant bits for every entry within 2^-14 of its row's largest,
//     an absolute 2^-36 of the row's largest below that (fp16 subnormals) -- against fp32's 24 bits the inputs are perturbed by at most
//     2^-23 of themselves, signed and zero-mean: over N observations that is 2^-23 sqrt(3 / N) of a diagonal entry, 1e-9 at config 3,
//     where fp32 LAPACK's own accumulation error is 3e-7;
//   * three products h h, h l, l h (l l is 2^-22 of the result and dropped) under fp32 accumulation, the product of the leading planes
//     in an accumulator of its own (every matrix instruction rounds its accumulator once: the small sum's roundings are 2^-11 of the
//     large one's);  G_ij = 2^(e_i + e_j) sum_n z'_in z'_jn, the scales applied to the finished tile (exact).
//   Half the matrix instructions and two thirds of the bytes of NP = 3.  Tests hold both to 4 x the error of fp32 LAPACK on the same
//   inputs (test_c3_full_size, test_c5_full_size, test_large_d_fp32_gram_on_bf16_matrix_cores_vs_f32_route).
//
// Kernels:
//   * rowmax_kernel (NP = 2): max_n |x_rn| sqrt(w_n) per row -- over a sample of the columns, doubled, or over all of them --,
//     atomicMax on the bit patterns (order-independent).
//   * planes_kernel (one pass over X, or over the raw inputs of a random-Fourier basis -- reference src/basis_function_regression.jl:41
//     materialises phi(x); here phi is evaluated once per element and leaves as planes, the fp32 feature matrix never exists):
//     z = x sqrt(w)   (w_n = 1 / s_n under diagonal noise, 1 otherwise: G = sum_n w_n x_n x_n' = Z Z', both operands the same)
//     stored as Xp[k-block of 16 columns][row block of 128][32-row sub-block j][plane p][lane][8 x 16 bit]: one KiB per (j, p) is ONE
//     matrix operand (lane = row r of the sub-block + 32 x (columns 8 .. 15)), 4 NP consecutive KiB are one side of a macro tile's half.
//     The same pass accumulates b = X r (r = delta / s, reference :57) in fp64 per row and column chunk: the Gram launch carries no
//     right-hand side.
//   * gram_planes4_kernel (the default, end of this file): two 256-thread workgroups per CU, 64 x 64 per wave.
//   * gram_planes_kernel: one 512-thread workgroup per CU and (macro tile, column range); a ring of halves of 8 NP KiB (A side + B
//     side), issued SLOTS - 1 halves ahead; wave (i, c) owns the two 32 x 32 tiles (i, 2c), (i, 2c + 1) of the 128 x 128 macro tile:
//     per half 3 NP fragment reads of 16 bytes per lane and 12 (NP = 3) or 6 (NP = 2) matrix instructions.  A diagonal macro tile
//     computes its tiles with column block <= row block (10 of 16) from the A side alone.  Split-K partial tiles in the layout of
//     gram_tile_kernel: gram_reduce_kernel is unchanged.
#pragma once
#include "blr_large.hpp"

namespace blr {

constexpr int kPlanesThreads = 512;
template <int NP> struct PlanesCfg {
  static constexpr int SIDE = 4 * NP * 1024;       // bytes of one side of a k-block: four 32-row sub-blocks x NP planes x 1 KiB
  static constexpr int HALF = 2 * SIDE;            // A side + B side of one k-block (16 columns)
  static constexpr int KBS = NP == 3 ? 1 : 2;      // k-blocks per stage = per workgroup barrier: 12 matrix instructions per wave either way
                                                   // (NP = 2 with one k-block per barrier: 327 us for config 3's Gram launch, the barrier and
                                                   // the fragment reads of 8 waves in step weigh as much as its 6 instructions)
  static constexpr int STAGE = KBS * HALF;         // 24 KiB / 32 KiB
  static constexpr int SLOTS = NP == 3 ? 6 : 4;    // stages in the ring (five -- all of the LDS -- measured the same)
  static constexpr int AHEAD = SLOTS - 1;          // stages in flight beyond the one being computed
  static constexpr int LDS = SLOTS * STAGE;        // 144 KiB / 128 KiB
  static constexpr int PW = NP * KBS;              // LDS-DMA pieces per loader wave and stage (waves 0 - 3: A side, 4 - 7: B side)
  static_assert(LDS <= 160 * 1024, "LDS of one CU");
};

typedef _Float16 gram_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 gram_h2 __attribute__((ext_vector_type(2)));

// row scales of the fp16 planes from the row's largest |z| (bit pattern; 0 for an all-zero row): z 2^-e in [2^12, 2^13)
__device__ __forceinline__ void planes_row_scale(unsigned maxbits, float& down /* 2^-e */, float& up /* 2^e */) {
  int E = (int)((maxbits >> 23) & 0xffu);  // biased exponent of the largest entry
  E = E < 13 ? 13 : (E > 254 ? 254 : E);   // (zero / denormal rows: any scale does; Inf / NaN rows stay Inf / NaN)
  down = __uint_as_float((unsigned)(266 - E) << 23);
  up = __uint_as_float((unsigned)(E - 12) << 23);
}

// ---- the producer --------------------------------------------------------------------------------------------------------------------
struct PlanesArgs {
  // source 0: X (D x N, ColVecs, element (d, n) at X[d + n ldx]); source 1: a random-Fourier basis phi_f(x_n) = scale cos(Omega_f' x_n + phase_f)
  const float* X; int64_t ldx;
  const float* Xin; int64_t ldxin; const float* Omega; int64_t ldo; const float* phase; float scale; int Din;
  const float* wsq;     // [N] sqrt(1 / s_n) (diagonal noise) or NULL
  const float* r;       // [N] delta_n / s_n for b = X r, or NULL
  unsigned short* Xp;   // planes (layout above)
  double* bpart;        // [nchunks][NC][128] partial sums of b (fixed order: one chunk, one writer)
  unsigned* rowmax;     // NP = 2: [DP] bit patterns of the rows' largest |z| (rowmax_kernel; a basis: |scale| max_n sqrt(w_n))
  int D, N, NC, NKB, nchunks;
  int64_t grp_X, grp_ws;  // blockIdx.z = regressor of a group: element stride of X, byte stride of wsq / r / Xp / bpart / rowmax
  // Speculative row scales (NP = 2, source 0).  The exact row maxima cost a pass over X of their own (58 us of config 3's 0.58 ms).
  // Instead: rowmax_kernel looks at the first `sample_kb` k-blocks of every column chunk only and stores TWICE what it finds (head-room:
  // the scale then puts the sample's largest entry into [2^11, 2^12), an entry of up to 16 x the sample's largest still is a finite fp16
  // number); planes_kernel raises `redo` when an entry does not fit all the same; the SECOND pair of launches (redo_pass = 1: the exact
  // row maxima over all columns, then the planes again) returns at once unless it is raised.  Gaussian rows: the largest of 65536 entries
  // is 1.3 x the largest of 2048; Student-t(3): 3 x.  Either way the result is a deterministic function of the inputs.
  int sample_kb;     // rowmax_kernel: k-blocks per chunk it reads (0: all of them)
  int redo_pass;     // 1: this launch runs only if *redo != 0
  unsigned* redo;    // [1] raised by planes_kernel (zeroed with rowmax by the launch before); NULL: exact row maxima, no check
  unsigned long long* redo_total;  // cumulative count of regressors whose planes were made twice (blr_get_stat "planes_redone"), or NULL
  // Shared-X multi-output evidence (blr_logpdf_multi_f32 at D > 128; reference: logpdf(fx, Y::Matrix)): Y != NULL makes the LAST row
  // block (NC - 1) hold the S <= 128 residual rows rho_s = (Y[:, s] - mu) sqrt(w) instead of rows of X.  The Gram launch then leaves
  // b_s = X Sigma^-1 (y_s - mu) in the macro tiles (NC - 1, J) -- the rows the blocked factorisation carries along as right-hand
  // sides -- and this pass the partial sums of q_s = (y_s - mu)' Sigma^-1 (y_s - mu) in fp64 (isotropic noise: without the 1 / s).
  int xs_chunk;         // source 1: 1 = the raw inputs of the workgroup's WHOLE column chunk are staged once ([k-block][8][16] floats, input
                        // dimensions padded to 8 with zeros: Din <= 8), the lane's row of Omega lives in registers -- per k-block the loop
                        // used to pay two workgroup barriers, a dependent global round trip and 64 cached loads of Omega per lane (config
                        // 5: 77 us for a pass whose arithmetic and stores take 35)
  const float* Y; int64_t ldY; int S;
  const float* mu;      // [N] x_n'mw (colstats_kernel), or NULL for a zero prior mean
  double* qsp;          // [nchunks][128]
};

__device__ __forceinline__ float rff_feature(const PlanesArgs& a, const float* __restrict__ om /* Omega_f */, float ph, const float* __restrict__ xs /* LDS: [Din][16] */,
                                             int col) {
  float acc = ph;
  for (int k = 0; k < a.Din; ++k) acc = __builtin_fmaf(om[k], xs[k * 16 + col], acc);
  return a.scale * rff_cos(acc);
}

constexpr int kPlanesChunkKb = 64;  // k-blocks per column chunk at most (the chunk's r and sqrt(w) live in LDS: 2 x 4 KiB)

// max_n |x_rn| sqrt(w_n) per row r (rowmax zeroed by the launch before: prior_diag_kernel's scratch initialisation).  One thread per row
// of the row block and column parity; positive floats order like their bit patterns, the maximum does not depend on the order.
template <bool RFF>
__global__ __launch_bounds__(kThreads) void rowmax_kernel(PlanesArgs a) {
  const int tid = threadIdx.x;
  if (const int64_t g = blockIdx.z) {
    if (!RFF) a.X += g * a.grp_X;
    a.wsq = ws_shift(a.wsq, g * a.grp_ws); a.rowmax = ws_shift(a.rowmax, g * a.grp_ws);
    if (a.redo) a.redo = ws_shift(a.redo, g * a.grp_ws);
  }
  if (a.redo_pass && *a.redo == 0u) return;  // (uniform over the regressor's workgroups)
  const int per = (a.NKB + a.nchunks - 1) / a.nchunks;
  const int n0 = 16 * blockIdx.x * per;
  const int n1 = min(a.N, 16 * min(a.NKB, (int)blockIdx.x * per + (a.sample_kb > 0 ? min(per, a.sample_kb) : per)));
  if constexpr (RFF) {  // |phi| <= |scale|: the bound is |scale| times the largest weight, the same for every row
    if (blockIdx.y != 0) return;
    __shared__ float red[kWaves];
    float m = 0.f;
    for (int n = n0 + tid; n < n1; n += kThreads) m = fmaxf(m, a.wsq ? a.wsq[n] : 1.f);
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = fabsf(a.scale) * fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    for (int d = tid; d < a.NC * kPB; d += kThreads) atomicMax(&a.rowmax[d], __float_as_uint(m));
  } else {
    const int row = blockIdx.y * kPB + (tid & 127);
    const bool yblk = a.Y != nullptr && (int)blockIdx.y == a.NC - 1;  // the residual rows of a multi-output call
    if (yblk ? (tid & 127) >= a.S : row >= a.D) return;
    const float* const yrow = yblk ? a.Y + (int64_t)(tid & 127) * a.ldY : nullptr;
    float m = 0.f;
    for (int n = n0 + (tid >> 7); n < n1; n += 32) {  // sixteen columns in flight per thread (four: 194 us for config 3's 268 MB)
      float v[16];
      if (yblk) {
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = (n + 2 * u < n1) ? yrow[n + 2 * u] - (a.mu ? a.mu[n + 2 * u] : 0.f) : 0.f;
      } else {
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = (n + 2 * u < n1) ? a.X[(int64_t)(n + 2 * u) * a.ldx + row] : 0.f;
      }
      if (a.wsq) {
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] *= (n + 2 * u < n1) ? a.wsq[n + 2 * u] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) m = fmaxf(m, fabsf(v[u]));
    }
    if (a.sample_kb > 0) m *= 2.f;  // head-room of a sampled maximum (an Inf stays an Inf: the planes pass then asks for the exact pass)
    atomicMax(&a.rowmax[row], __float_as_uint(m));  // (a NaN entry: fmaxf drops it; the planes keep it, the factorisation reports it)
  }
}

template <int NP, bool RFF>
__global__ __launch_bounds__(kThreads) void planes_kernel(PlanesArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* const rs = reinterpret_cast<float*>(smem);            // the chunk's r_n          [16 per]
  float* const wsm = rs + 16 * kPlanesChunkKb;                 // the chunk's sqrt(w_n)    [16 per] (1 without weights, 0 beyond N)
  float* const mus = wsm + 16 * kPlanesChunkKb;                // the chunk's mu_n (residual rows of a multi-output call)
  float* const xs = mus + 16 * kPlanesChunkKb;                 // RFF: the k-block's raw inputs [Din][16]
  const int tid = threadIdx.x, lane = tid & 63, j = tid >> 6;  // wave j: rows 32 j .. 32 j + 31 of row block I
  const int I = blockIdx.y;
  if (const int64_t g = blockIdx.z) {
    if (!RFF) a.X += g * a.grp_X;
    a.wsq = ws_shift(a.wsq, g * a.grp_ws); a.r = ws_shift(a.r, g * a.grp_ws); a.Xp = ws_shift(a.Xp, g * a.grp_ws);
    a.bpart = ws_shift(a.bpart, g * a.grp_ws); a.rowmax = ws_shift(a.rowmax, g * a.grp_ws);
    if (a.redo) a.redo = ws_shift(a.redo, g * a.grp_ws);
  }
  if (a.redo_pass && *a.redo == 0u) return;  // (uniform over the regressor's workgroups; nobody of this launch writes the flag)
  const bool check = NP == 2 && !RFF && a.redo != nullptr && !a.redo_pass;
  float umax = 0.f;  // largest scaled entry this lane has split (NP = 2)
  const int per = (a.NKB + a.nchunks - 1) / a.nchunks;  // (<= kPlanesChunkKb: the host sizes nchunks)
  const int kb0 = blockIdx.x * per, kb1 = min(a.NKB, kb0 + per);
  const int r32 = lane & 31, kh = lane >> 5;
  const int row = I * kPB + 32 * j + r32;
  const bool yblk = !RFF && a.Y != nullptr && I == a.NC - 1;   // the residual rows of a multi-output call (uniform over the workgroup)
  const bool row_ok = yblk ? 32 * j + r32 < a.S : row < a.D;
  const float* const yrow = yblk ? a.Y + (int64_t)(32 * j + r32) * a.ldY : nullptr;
  // the chunk's per-column scalars once, through LDS (as loads inside the loop they were a dependent L2 round trip per k-block: 234 us
  // for config 3's 671 MB)
  for (int c = tid; c < 16 * (kb1 - kb0); c += kThreads) {
    const int n = 16 * kb0 + c;
    rs[c] = (a.r && n < a.N) ? a.r[n] : 0.f;
    wsm[c] = n < a.N ? (a.wsq ? a.wsq[n] : 1.f) : 0.f;
    if (yblk) mus[c] = (a.mu && n < a.N) ? a.mu[n] : 0.f;
  }
  float omr[8];  // RFF, xs_chunk: this lane's row of Omega, zero beyond Din
  if constexpr (RFF) {
    if (a.xs_chunk) {
      for (int idx = tid; idx < 8 * 16 * (kb1 - kb0); idx += kThreads) {
        const int k = idx & 7, c = idx >> 3;   // consecutive threads: consecutive input dimensions of one column (contiguous in memory)
        const int n = 16 * kb0 + c;
        xs[(c >> 4) * 128 + k * 16 + (c & 15)] = (k < a.Din && n < a.N) ? a.Xin[(int64_t)n * a.ldxin + k] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) omr[k] = (row < a.D && k < a.Din) ? a.Omega[(int64_t)row * a.ldo + k] : 0.f;
    }
  }
  float down = 1.f, up = 1.f;
  if constexpr (NP == 2) planes_row_scale(a.rowmax[row], down, up);
  (void)up;
  __syncthreads();
  double bacc = 0.0;
  const float* om = nullptr;
  float ph = 0.f;
  if constexpr (RFF) {
    if (row_ok) { om = a.Omega + (int64_t)row * a.ldo; ph = a.phase[row]; }
  }
  float xa[8], xb[8];  // raw entries of this lane's row in columns 16 kb + 8 kh .. + 7, two k-blocks ahead
  auto fetch = [&](int kb, float (&dst)[8]) {
    if constexpr (!RFF) {
      if (yblk) {  // eight consecutive observations of output column 32 j + r32: contiguous in Y
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int n = 16 * kb + 8 * kh + e;
          dst[e] = (row_ok && kb < kb1 && n < a.N) ? yrow[n] - mus[16 * (kb - kb0) + 8 * kh + e] : 0.f;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int n = 16 * kb + 8 * kh + e;
          dst[e] = (row_ok && kb < kb1 && n < a.N) ? a.X[(int64_t)n * a.ldx + row] : 0.f;
        }
      }
    }
  };
  fetch(kb0, xa);
  fetch(kb0 + 1, xb);
  constexpr int FR = 4 * NP;  // KiB per row block and k-block
  for (int kb = kb0; kb < kb1; ++kb) {
    float x[8];
    if constexpr (RFF) {
      if (a.xs_chunk) {  // (uniform over the launch)
        const float* const xk = xs + (kb - kb0) * 128;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float acc = ph;
#pragma unroll
          for (int k = 0; k < 8; ++k) acc = __builtin_fmaf(omr[k], xk[k * 16 + 8 * kh + e], acc);  // (same order as rff_feature; a zero term adds nothing)
          x[e] = (row_ok && 16 * kb + 8 * kh + e < a.N) ? a.scale * rff_cos(acc) : 0.f;
        }
      } else {
      __syncthreads();
      for (int idx = tid; idx < a.Din * 16; idx += kThreads) {
        const int k = idx % a.Din, c = idx / a.Din;  // consecutive threads: consecutive input dimensions of one column (contiguous in memory)
        const int n = 16 * kb + c;
        xs[k * 16 + c] = n < a.N ? a.Xin[(int64_t)n * a.ldxin + k] : 0.f;
      }
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = (row_ok && 16 * kb + 8 * kh + e < a.N) ? rff_feature(a, om, ph, xs, 8 * kh + e) : 0.f;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) { x[e] = xa[e]; xa[e] = xb[e]; }
      fetch(kb + 2, xb);  // in flight while this k-block and the next are split and stored
    }
    float z[8];
    const float* rk = rs + 16 * (kb - kb0) + 8 * kh;
    const float* wk = wsm + 16 * (kb - kb0) + 8 * kh;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = x[e] * wk[e];
    if (yblk) {  // (uniform) a residual row: its share of q_s = sum_n w_n d_n^2
#pragma unroll
      for (int e = 0; e < 8; ++e) bacc += (double)z[e] * (double)z[e];
    } else {     // b = X r (exact products, fp64 sum: as the Gram kernels' b partials)
#pragma unroll
      for (int e = 0; e < 8; ++e) bacc += (double)x[e] * (double)rk[e];
    }
    gram_u4* dst = reinterpret_cast<gram_u4*>(reinterpret_cast<char*>(a.Xp) + (((int64_t)kb * a.NC + I) * FR + NP * j) * 1024) + lane;
    if constexpr (NP == 3) {
      gram_u4 H, M, L;
#pragma unroll
      for (int q = 0; q < 4; ++q) {  // (bf3_split_pack's arithmetic, without the lane exchange: the layout is made here)
        const float u = z[2 * q], v = z[2 * q + 1];
        const unsigned hh = bf3_pk(u, v);
        const float ur = u - __uint_as_float(hh << 16), vr = v - __uint_as_float(hh & 0xffff0000u);
        const unsigned mm = bf3_pk(ur, vr);
        const float ul = ur - __uint_as_float(mm << 16), vl = vr - __uint_as_float(mm & 0xffff0000u);
        H[q] = hh; M[q] = mm; L[q] = bf3_pk(ul, vl);
      }
      dst[0] = H; dst[64] = M; dst[128] = L;  // three KiB, each written by one wave instruction
    } else {
      gram_u4 H, L;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float u = z[2 * q] * down, v = z[2 * q + 1] * down;   // (exact: a power of two)
        umax = fmaxf(umax, fmaxf(fabsf(u), fabsf(v)));              // (one v_max3_f32; a NaN is passed by, as in rowmax_kernel)
        const _Float16 hu = (_Float16)u, hv = (_Float16)v;          // round to nearest even
        const _Float16 lu = (_Float16)(u - (float)hu), lv = (_Float16)(v - (float)hv);
        const gram_h2 hp = {hu, hv}, lp = {lu, lv};
        H[q] = __builtin_bit_cast(unsigned, hp); L[q] = __builtin_bit_cast(unsigned, lp);
      }
      dst[0] = H; dst[64] = L;
    }
  }
  // beyond the largest finite fp16 number: the sampled scale did not hold for this row
  if (check && __any(umax > 65504.f) && lane == 0) {
    if (atomicOr(a.redo, 1u) == 0u && a.redo_total) atomicAdd(a.redo_total, 1ull);  // (the one thread that raises it counts the regressor)
  }
  if (yblk) {
    bacc += __shfl_xor(bacc, 32);
    if (kh == 0 && a.qsp) a.qsp[(int64_t)blockIdx.x * kPB + 32 * j + r32] = bacc;
  } else if (a.bpart) {
    bacc += __shfl_xor(bacc, 32);  // the two column halves of a row
    if (kh == 0) a.bpart[((int64_t)blockIdx.x * a.NC + I) * kPB + 32 * j + r32] = bacc;
  }
}

// ---- the consumer --------------------------------------------------------------------------------------------------------------------
struct GramPlanesArgs {
  const unsigned short* Xp;
  int NC, NKB;
  float* Gpart;            // [nsplit][ntiles][128 * 128] column-major tiles (row = A-side row): gram_tile_kernel's layout
  int ntiles, nsplit;
  const float* s_iso;      // isotropic noise: the variance (device scalar; 1 / s is applied to the finished tile), else NULL
  const unsigned* rowmax;  // NP = 2: the rows' largest |z| (bit patterns): the tile leaves scaled by 2^(e_i + e_j)
  int xcd_swizzle;
  int64_t grp_ws, grp_s;   // blockIdx.y = regressor of a group: byte stride of Xp / Gpart / rowmax, element stride of s_iso
};

template <int NP>
__global__ __launch_bounds__(kPlanesThreads, 2) void gram_planes_kernel(GramPlanesArgs a) {
  using C = PlanesCfg<NP>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int ti = wave >> 1, tc = wave & 1;  // tile row i, tile columns 2 tc, 2 tc + 1 of the macro tile
  // work item: (split, tile), one XCD owning contiguous runs of them (gram_tile_kernel: the tiles of one column range share an L2)
  int w = blockIdx.x;
  if (a.xcd_swizzle) {
    const int nwg = gridDim.x, xcd = w & 7, qq = nwg >> 3, rr = nwg & 7;
    w = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (w >> 3);
  }
  if (const int64_t g = blockIdx.y) {
    a.Xp = ws_shift(a.Xp, g * a.grp_ws); a.Gpart = ws_shift(a.Gpart, g * a.grp_ws); a.rowmax = ws_shift(a.rowmax, g * a.grp_ws);
    if (a.s_iso) a.s_iso += g * a.grp_s;
  }
  const float post_scale = a.s_iso ? 1.0f / a.s_iso[0] : 1.0f;
  const int t = w % a.ntiles, sidx = w / a.ntiles;
  int I = 0;
  while ((I + 1) * (I + 2) / 2 <= t) ++I;
  const int J = t - I * (I + 1) / 2;
  const bool diag = I == J;
  // column range of this split in stages of KBS k-blocks (the host pads the planes to whole stages: zero columns beyond N)
  const int nst_all = a.NKB / C::KBS;
  const int per = (nst_all + a.nsplit - 1) / a.nsplit;
  const int st0 = sidx * per, st1 = min(nst_all, st0 + per);
  const int nh = st1 > st0 ? st1 - st0 : 0;   // stages of this work item
  const int kb0 = st0 * C::KBS;

  // Two accumulators per tile: the product of the leading planes (magnitude 1) in one, the small products (2^-8 .. 2^-16 of it with bf16
  // planes, 2^-11 with fp16 planes) in the other.  Every matrix instruction rounds its accumulator once; with all of them in one register
  // the sum picked up six roundings of ITS magnitude per 16 columns -- 1.4e-6 of max |A| over the 74 halves of a column range at config
  // 3's shape reduced to N = 8192, 4.04 x the error of fp32 LAPACK -- now one.
  gram_f16v acc[2], accs[2];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int v = 0; v < 16; ++v) { acc[k][v] = 0.f; accs[k][v] = 0.f; }
  // one product, into the accumulator of its class
#define BLR_PM3(ACC, XP, YP) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gram_bf8, XP), __builtin_bit_cast(gram_bf8, YP), ACC, 0, 0, 0)
#define BLR_PM2(ACC, XP, YP) ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gram_h8, XP), __builtin_bit_cast(gram_h8, YP), ACC, 0, 0, 0)
  // which of the wave's two tiles exist: a diagonal macro tile keeps column block <= row block
  const bool t0 = !diag || 2 * tc <= ti, t1 = !diag || 2 * tc + 1 <= ti;

  // LDS-DMA: waves 0 - 3 bring the A side (pieces NP w .. NP w + NP - 1 of its 4 NP), waves 4 - 7 the B side (none for a diagonal macro
  // tile); always issued (beyond the last half the last one is fetched again into a slot nobody reads: no branch in the loop, one
  // constant in the wait)
  const bool loader = wave < 4 || !diag;
  const int side = wave >> 2, p0 = NP * (wave & 3);
  unsigned ring_addr = lds_addr_of(smem);
  asm volatile("" : "+v"(ring_addr));
  const uint64_t kbstep = (uint64_t)a.NC * (uint64_t)C::SIDE;
  uint64_t next = (uint64_t)(uintptr_t)a.Xp + ((uint64_t)kb0 * a.NC + (uint64_t)(side ? J : I)) * (uint64_t)C::SIDE + (uint64_t)p0 * 1024u;
  const unsigned voff = (unsigned)lane * 16u;
  int hi = 0;  // stage the next issue belongs to
  auto issue = [&]() {
#if defined(BLR_PLANES_EXP) && BLR_PLANES_EXP == 2  /* timing experiment: no LDS-DMA beyond the prologue */
    if (loader && hi < C::AHEAD) {
#else
    if (loader) {
#endif
      const unsigned slot = ring_addr + (unsigned)((hi % C::SLOTS) * C::STAGE + side * C::SIDE + p0 * 1024);
#pragma unroll
      for (int u = 0; u < C::KBS; ++u)
#pragma unroll
        for (int c = 0; c < NP; ++c)
          glds_s<16>(uni((int64_t)(next + (uint64_t)u * kbstep + (uint64_t)c * 1024u)), voff, slot + (unsigned)(u * C::HALF + c * 1024));
    }
    if (hi + 1 < nh) next += (uint64_t)C::KBS * kbstep;
    ++hi;
  };
  // the wave's 3 NP operand fragments of a half: one row sub-block of the A side, two of the B side (the A side again on a diagonal tile)
  const int offa = (ti * NP) * 1024 + lane * 16, offb = (diag ? 0 : C::SIDE) + (2 * tc * NP) * 1024 + lane * 16;
  // One half: the fragment reads, the LDS-DMA pieces of half h + AHEAD, the matrix instructions -- the two tiles' chains interleaved (a
  // dependent instruction waits for its predecessor), smallest terms first within an accumulator.
  // (Measured and not shipped, NP = 3, gram launch of config 3 on one box: the fragments of half h + 1 read into a second register set
  // under this half's instructions 410 us against 374; the same with reads and pieces pinned one behind each matrix instruction 432.
  // The loop is not short of overlap: without its matrix instructions it takes 171 us, without its LDS-DMA 361 -- the matrix pipe on
  // random operands at the clock the part then holds (MI355X_MICROARCH.md, DVFS give-back (5): 1.5 - 1.7 GHz) IS the time, and a
  // denser instruction stream lowers that clock further.)
  constexpr int NF = 3 * NP * C::KBS;  // fragments of a stage: per k-block A, B0, B1 with NP planes each
  struct Frags { gram_u4 v[NF]; };
  auto read_stage = [&](const char* slot, Frags& f) {
#pragma unroll
    for (int u = 0; u < C::KBS; ++u) {
#pragma unroll
      for (int p = 0; p < NP; ++p) f.v[u * 3 * NP + p] = *reinterpret_cast<const gram_u4*>(slot + u * C::HALF + offa + p * 1024);
#pragma unroll
      for (int p = 0; p < NP; ++p) f.v[u * 3 * NP + NP + p] = *reinterpret_cast<const gram_u4*>(slot + u * C::HALF + offb + p * 1024);
#pragma unroll
      for (int p = 0; p < NP; ++p) f.v[u * 3 * NP + 2 * NP + p] = *reinterpret_cast<const gram_u4*>(slot + u * C::HALF + offb + (NP + p) * 1024);
    }
  };
  auto compute = [&](const Frags& f) {
#if defined(BLR_PLANES_EXP) && BLR_PLANES_EXP == 1  /* timing experiment: no matrix instructions */
    acc[0][0] += __uint_as_float(f.v[0][0] ^ f.v[NP][0] ^ f.v[NF - 1][3]);
    return;
#endif
    auto kblock = [&](auto utag) {
      constexpr int u = decltype(utag)::value;
      if constexpr (u < C::KBS) {
#define A(p) f.v[u * 3 * NP + (p)]
#define B0(p) f.v[u * 3 * NP + NP + (p)]
#define B1(p) f.v[u * 3 * NP + 2 * NP + (p)]
      if constexpr (NP == 3) {  // planes h, m, l
        if (t0 && t1) {
          BLR_PM3(accs[0], A(2), B0(0)); BLR_PM3(accs[1], A(2), B1(0));
          BLR_PM3(acc[0], A(0), B0(0));  BLR_PM3(acc[1], A(0), B1(0));
          BLR_PM3(accs[0], A(0), B0(2)); BLR_PM3(accs[1], A(0), B1(2));
          BLR_PM3(accs[0], A(1), B0(1)); BLR_PM3(accs[1], A(1), B1(1));
          BLR_PM3(accs[0], A(1), B0(0)); BLR_PM3(accs[1], A(1), B1(0));
          BLR_PM3(accs[0], A(0), B0(1)); BLR_PM3(accs[1], A(0), B1(1));
        } else if (t0) {
          BLR_PM3(accs[0], A(2), B0(0)); BLR_PM3(acc[0], A(0), B0(0)); BLR_PM3(accs[0], A(0), B0(2));
          BLR_PM3(accs[0], A(1), B0(1)); BLR_PM3(accs[0], A(1), B0(0)); BLR_PM3(accs[0], A(0), B0(1));
        }
      } else {  // planes h, l
        if (t0 && t1) {
          BLR_PM2(accs[0], A(1), B0(0)); BLR_PM2(accs[1], A(1), B1(0));
          BLR_PM2(acc[0], A(0), B0(0));  BLR_PM2(acc[1], A(0), B1(0));
          BLR_PM2(accs[0], A(0), B0(1)); BLR_PM2(accs[1], A(0), B1(1));
        } else if (t0) {
          BLR_PM2(accs[0], A(1), B0(0)); BLR_PM2(acc[0], A(0), B0(0)); BLR_PM2(accs[0], A(0), B0(1));
        }
      }
#undef A
#undef B0
#undef B1
      }
    };
    kblock(std::integral_constant<int, 0>{});
    kblock(std::integral_constant<int, 1>{});
  };
  const auto end_of_stage = [&]() {
    // stage h + 1 must have landed (the younger ones may stay in flight), then everybody's pieces are visible
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::PW * (C::AHEAD - 1)) : "memory");
    __syncthreads();
  };
  if (nh > 0) {
    for (int q = 0; q < C::AHEAD; ++q) issue();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::PW * (C::AHEAD - 1)) : "memory");  // stage 0 has landed
    __syncthreads();
    // The two waves of a SIMD (w, w + 4) work half a stage apart.  Waves 0 - 3: read stage h, compute it.  Waves 4 - 7: compute stage
    // h - 1 from the registers they filled in the interval before, THEN read stage h -- their matrix instructions run under their
    // partners' fragment reads and LDS-DMA issue and the other way round (in step, every wave of the CU read, then every wave computed:
    // the Gram launch of config 3 took what its matrix instructions and its data movement take one after the other, 262 us of 180 + 115).
    // Stage h + AHEAD goes into the slot of stage h - 1, which everybody had read by the barrier that ended interval h - 1.
    if (wave < 4) {
#pragma unroll 1
      for (int h = 0; h < nh; ++h) {
        Frags f;
        read_stage(smem + (h % C::SLOTS) * C::STAGE, f);
        issue();
        compute(f);
        end_of_stage();
      }
    } else {
      // (their fragment reads are not consumed before the barrier: drained explicitly, so that no read of a slot is still in the LDS
      // queue when, an interval later, a partner's LDS-DMA piece is issued into it)
      const auto end_of_stage_lag = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        end_of_stage();
      };
      Frags f;
      read_stage(smem, f);
      issue();
      end_of_stage_lag();
#pragma unroll 1
      for (int h = 1; h < nh; ++h) {
        // (priority over the partner for the matrix instructions: they run first and back to back, the partner's -- whose fragments
        // are still on their way from the LDS -- behind them; interleaved, this wave's reads below started late and ended the interval)
        __builtin_amdgcn_s_setprio(2);
        compute(f);  // stage h - 1
        __builtin_amdgcn_s_setprio(0);
        read_stage(smem + (h % C::SLOTS) * C::STAGE, f);  // (into the registers the instructions above have read)
        issue();
        end_of_stage_lag();
      }
      compute(f);  // stage nh - 1
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the pieces issued beyond the last stage)
  }
#undef BLR_PM3
#undef BLR_PM2
  // row scales of the macro tile's rows and columns (NP = 2), through the ring (everybody left it at the last barrier)
  float* const sI = reinterpret_cast<float*>(smem);
  float* const sJ = sI + kPB;
  if constexpr (NP == 2) {
    __syncthreads();
    if (tid < 2 * kPB) {
      float down, up;
      planes_row_scale(a.rowmax[(tid < kPB ? I : J) * kPB + (tid & (kPB - 1))], down, up);
      sI[tid] = up;
    }
    __syncthreads();
  }
  // ---- epilogue: the wave's tiles into the split's partial tile, column-major (row = A-side row): a lane holds 4 consecutive rows
  float* out = a.Gpart + ((int64_t)sidx * a.ntiles + t) * (kPB * kPB);
  typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (!(k == 0 ? t0 : t1)) continue;  // (wave-uniform; the reduction never reads the strictly upper tiles of a diagonal macro tile)
    const int col = 32 * (2 * tc + k) + (lane & 31);
    const float cs = NP == 2 ? post_scale * sJ[col] : post_scale;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row0 = 32 * ti + 8 * q + 4 * (lane >> 5);
      f4 v = {acc[k][4 * q] + accs[k][4 * q], acc[k][4 * q + 1] + accs[k][4 * q + 1], acc[k][4 * q + 2] + accs[k][4 * q + 2], acc[k][4 * q + 3] + accs[k][4 * q + 3]};
      if constexpr (NP == 2) {
        const f4 rsc = *reinterpret_cast<const f4*>(sI + row0);
        v *= rsc;
      }
      v *= cs;
      *reinterpret_cast<f4*>(out + (int64_t)col * kPB + row0) = v;
    }
  }
}

// ---- the consumer, 64 x 64 per wave (NP = 2) ------------------------------------------------------------------------------------------
// Four waves per workgroup, each owning a 64 x 64 quadrant of the macro tile (four 32 x 32 tiles, eight accumulators of 16 registers),
// two workgroups per CU (ring of four halves of 16 KiB each): per half and wave 8 fragment reads for 12 matrix instructions where
// gram_planes_kernel reads 12 for 12, and the two workgroups of a CU drift apart by themselves -- one's reads and barrier under the
// other's matrix instructions.  Same planes, same partial tiles.
struct Planes4Cfg {
  static constexpr int SIDE = 8 * 1024, HALF = 2 * SIDE, SLOTS = 4, AHEAD = SLOTS - 1, LDS = SLOTS * HALF;
};
__global__ __launch_bounds__(kThreads, 2) void gram_planes4_kernel(GramPlanesArgs a) {
  using C = Planes4Cfg;
  constexpr int NP = 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  int w = blockIdx.x;
  if (a.xcd_swizzle) {
    const int nwg = gridDim.x, xcd = w & 7, qq = nwg >> 3, rr = nwg & 7;
    w = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (w >> 3);
  }
  if (const int64_t g = blockIdx.y) {
    a.Xp = ws_shift(a.Xp, g * a.grp_ws); a.Gpart = ws_shift(a.Gpart, g * a.grp_ws); a.rowmax = ws_shift(a.rowmax, g * a.grp_ws);
    if (a.s_iso) a.s_iso += g * a.grp_s;
  }
  const float post_scale = a.s_iso ? 1.0f / a.s_iso[0] : 1.0f;
  const int t = w % a.ntiles, sidx = w / a.ntiles;
  int I = 0;
  while ((I + 1) * (I + 2) / 2 <= t) ++I;
  const int J = t - I * (I + 1) / 2;
  const bool diag = I == J;
  const int per = (a.NKB + a.nsplit - 1) / a.nsplit;
  const int kb0 = sidx * per, kb1 = min(a.NKB, kb0 + per);
  const int nh = kb1 > kb0 ? kb1 - kb0 : 0;
  gram_f16v acc[2][2], accs[2][2];  // [row sub-block a][column sub-block b]: leading product | the two small ones
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int v = 0; v < 16; ++v) { acc[x][y][v] = 0.f; accs[x][y][v] = 0.f; }
  // tile (a, b) of the quadrant exists unless the macro tile is diagonal and the tile lies above its diagonal
  const bool ex00 = !diag || 2 * wc <= 2 * wr, ex01 = !diag || 2 * wc + 1 <= 2 * wr, ex10 = !diag || 2 * wc <= 2 * wr + 1, ex11 = !diag || 2 * wc + 1 <= 2 * wr + 1;
  unsigned ring_addr = lds_addr_of(smem);
  asm volatile("" : "+v"(ring_addr));
  const uint64_t kbstep = (uint64_t)a.NC * (uint64_t)C::SIDE;
  const uint64_t baseA = (uint64_t)(uintptr_t)a.Xp + ((uint64_t)kb0 * a.NC + (uint64_t)I) * (uint64_t)C::SIDE + (uint64_t)(2 * wave) * 1024u;
  const uint64_t baseB = (uint64_t)(uintptr_t)a.Xp + ((uint64_t)kb0 * a.NC + (uint64_t)J) * (uint64_t)C::SIDE + (uint64_t)(2 * wave) * 1024u;
  uint64_t adv = 0;
  const unsigned voff = (unsigned)lane * 16u;
  int hi = 0;
  auto run = [&](auto dtag) {
    constexpr bool DG = decltype(dtag)::value;
    constexpr int PW = DG ? 2 : 4;  // LDS-DMA pieces per wave and half: its two KiB of the A side (+ of the B side)
    auto issue = [&]() {
      const unsigned slot = ring_addr + (unsigned)((hi % C::SLOTS) * C::HALF + 2 * wave * 1024);
      glds_s<16>(uni((int64_t)(baseA + adv)), voff, slot);
      glds_s<16>(uni((int64_t)(baseA + adv + 1024u)), voff, slot + 1024u);
      if constexpr (!DG) {
        glds_s<16>(uni((int64_t)(baseB + adv)), voff, slot + (unsigned)C::SIDE);
        glds_s<16>(uni((int64_t)(baseB + adv + 1024u)), voff, slot + (unsigned)C::SIDE + 1024u);
      }
      if (hi + 1 < nh) adv += kbstep;
      ++hi;
    };
    const int offa = (2 * wr * NP) * 1024 + lane * 16, offb = (DG ? 0 : C::SIDE) + (2 * wc * NP) * 1024 + lane * 16;
    for (int q = 0; q < C::AHEAD; ++q) issue();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW * (C::AHEAD - 1)) : "memory");
    __syncthreads();
#define BLR_PM2(ACC, XP, YP) ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gram_h8, XP), __builtin_bit_cast(gram_h8, YP), ACC, 0, 0, 0)
#pragma unroll 1
    for (int h = 0; h < nh; ++h) {
      const char* slot = smem + (h % C::SLOTS) * C::HALF;
      gram_u4 A0h = *reinterpret_cast<const gram_u4*>(slot + offa), A0l = *reinterpret_cast<const gram_u4*>(slot + offa + 1024);
      gram_u4 A1h = *reinterpret_cast<const gram_u4*>(slot + offa + 2048), A1l = *reinterpret_cast<const gram_u4*>(slot + offa + 3072);
      gram_u4 B0h = *reinterpret_cast<const gram_u4*>(slot + offb), B0l = *reinterpret_cast<const gram_u4*>(slot + offb + 1024);
      gram_u4 B1h = *reinterpret_cast<const gram_u4*>(slot + offb + 2048), B1l = *reinterpret_cast<const gram_u4*>(slot + offb + 3072);
      issue();
      __builtin_amdgcn_s_setprio(2);  // over the other workgroup's wave on this SIMD while the matrix instructions issue (same-box A/B: c3 -0.7 %, 8 x c3 -1.5 %)
      // four independent chains interleaved; within a tile the small products, then the leading one
      if (ex10) {  // (wave-uniform; the full quadrant in every off-diagonal macro tile)
        if (ex00) BLR_PM2(accs[0][0], A0l, B0h);
        BLR_PM2(accs[1][0], A1l, B0h);
        if (ex01) BLR_PM2(accs[0][1], A0l, B1h);
        if (ex11) BLR_PM2(accs[1][1], A1l, B1h);
        if (ex00) BLR_PM2(accs[0][0], A0h, B0l);
        BLR_PM2(accs[1][0], A1h, B0l);
        if (ex01) BLR_PM2(accs[0][1], A0h, B1l);
        if (ex11) BLR_PM2(accs[1][1], A1h, B1l);
        if (ex00) BLR_PM2(acc[0][0], A0h, B0h);
        BLR_PM2(acc[1][0], A1h, B0h);
        if (ex01) BLR_PM2(acc[0][1], A0h, B1h);
        if (ex11) BLR_PM2(acc[1][1], A1h, B1h);
      }
      __builtin_amdgcn_s_setprio(0);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW * (C::AHEAD - 1)) : "memory");
      __syncthreads();
    }
#undef BLR_PM2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  if (nh > 0) {
    if (diag) run(std::true_type{});
    else run(std::false_type{});
  }
  float* const sI = reinterpret_cast<float*>(smem);
  float* const sJ = sI + kPB;
  __syncthreads();
  {
    float down, up;
    planes_row_scale(a.rowmax[(tid < kPB ? I : J) * kPB + (tid & (kPB - 1))], down, up);
    sI[tid] = up;  // (256 threads: rows of I, then rows of J)
  }
  __syncthreads();
  float* out = a.Gpart + ((int64_t)sidx * a.ntiles + t) * (kPB * kPB);
  typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const bool ex = x == 0 ? (y == 0 ? ex00 : ex01) : (y == 0 ? ex10 : ex11);
      if (!ex) continue;
      const int col = 32 * (2 * wc + y) + (lane & 31);
      const float cs = post_scale * sJ[col];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row0 = 32 * (2 * wr + x) + 8 * q + 4 * (lane >> 5);
        f4 v = {acc[x][y][4 * q] + accs[x][y][4 * q], acc[x][y][4 * q + 1] + accs[x][y][4 * q + 1], acc[x][y][4 * q + 2] + accs[x][y][4 * q + 2],
                acc[x][y][4 * q + 3] + accs[x][y][4 * q + 3]};
        v *= *reinterpret_cast<const f4*>(sI + row0);
        v *= cs;
        *reinterpret_cast<f4*>(out + (int64_t)col * kPB + row0) = v;
      }
    }
}


}  // namespace blr
