// One WAVEFRONT per regressor: the fused posterior update for D = 32 / 64 (config 4 of BASELINE.json: batches of small
// independent regressors), ColVecs data with 16-byte aligned columns; every prior kind.
//
//   reference src/bayesian_linear_regression.jl:55-69, :72-89 (logpdf + posterior; the same direct Gram form as
//   blr_fused_small.hpp:  A = Lw + X S^-1 X',  b = X S^-1 (y - X'mw),  A = L L',  u = L^-1 b,  m = L^-T u)
//
// Why a second kernel: with four waves per regressor (blr_fused_small.hpp) a D = 64 update spends more time in workgroup
// barriers, in waiting for its slowest wave and in re-reading fragments than in the matrix pipe -- 10 lower-triangular
// tiles do not divide over 4 waves (3 + 3 + 2 + 2), each wave reads 7 fragments for 3 MFMAs, and every stage of 8 k-steps
// pays a vmcnt(0) + s_barrier convoy (measured, tools/fused_bench -DBLR_GRAM_STAMPS: 10.5 k cycles per stage for 1.5 k
// cycles of MFMA per wave; 47 % MFMA-busy with 4 workgroups per CU).  Here ONE wave owns all 10 tiles:
//   * 4 fragment reads per k-step feed 10 MFMAs (the A and B side of a tile are the same 4 row-block fragments) and the
//     column-vector work (delta, b, quadratic form) reuses the same registers -- LDS traffic per k-step drops 7x;
//   * no workgroup barrier anywhere: the wave stages its own X through a two-slot ring with LDS-DMA and waits on vmcnt;
//   * the 64 rows of a panel of the blocked Cholesky are exactly one row per lane;
//   * 18 KB of LDS per wave -> 8 waves per CU, two per SIMD: while one factors / substitutes on the vector ALU the other
//     keeps the matrix pipe busy.
// Anything outside the fast path (RowVecs, unaligned columns, D not 32 / 64) stays on blr_fused_small.hpp.
#pragma once
#ifndef BLR_WAVE_NT
#define BLR_WAVE_NT true  /* the X stream of the per-regressor kernels is read once: non-temporal LDS-DMA pieces (blr_common.hpp, glds_s) */
#endif
#include "blr_fused_small.hpp"

namespace blr {

template <typename T, int NB>
struct WaveCfg {
  static constexpr int DP = 16 * NB;
  static constexpr int NT = NB * (NB + 1) / 2;
#ifndef BLR_WAVE_STAGE_BYTES
#define BLR_WAVE_STAGE_BYTES 4096
#endif
  static constexpr int KS = (BLR_WAVE_STAGE_BYTES / 64) / (NB * (int)sizeof(T));  // k-steps per stage: every stage is 4 KiB of X
  static constexpr int NSC = 4 * KS;                   // columns per stage
#ifndef BLR_WAVE_RING_BYTES
#define BLR_WAVE_RING_BYTES 16384
#endif
  static constexpr int DEPTH = BLR_WAVE_RING_BYTES / BLR_WAVE_STAGE_BYTES;  // ring slots (16 KiB): all but one in flight while one is consumed
  static constexpr int SLOT = KS * NB * 64;            // elements per ring slot (fragment order [k-step][row block][lane])
  static constexpr int PACKED = DP * (DP + 1) / 2;
  static constexpr int RING_BYTES = DEPTH * SLOT * (int)sizeof(T);
  static constexpr int P_BYTES = PACKED * (int)sizeof(T);
  static constexpr int REGION0 = ((RING_BYTES > P_BYTES ? RING_BYTES : P_BYTES) + 15) & ~15;  // P aliases the ring
  static constexpr int OFF_Y = REGION0;                                      // ybuf[DEPTH][NSC]  (LDS-DMA target)
  static constexpr int OFF_S = OFF_Y + DEPTH * NSC * (int)sizeof(T);         // sbuf[DEPTH][NSC]  (LDS-DMA target, diagonal noise)
  static constexpr int OFF_W = OFF_S + DEPTH * NSC * (int)sizeof(T);         // wbuf[NSC]
  static constexpr int OFF_B = OFF_W + NSC * (int)sizeof(T);                 // bvec[DP]
  static constexpr int OFF_SCR = (OFF_B + DP * (int)sizeof(T) + 15) & ~15;   // 4 doubles + 4 ints
  static constexpr int OFF_PART = OFF_SCR + 48;                              // N split over several waves: this wave's partial
                                                                             // b (DP doubles), quadratic form, logdet, bad index
  static constexpr int LDS_BYTES = OFF_PART + DP * 8 + 32;                   // one wave's slice of the workgroup's LDS
  static constexpr int FPG = (1024 / (int)sizeof(T)) / 64;               // fragments per 1 KiB LDS-DMA piece
  static constexpr int NG = KS * NB / FPG;                               // X pieces per stage
  static constexpr int YL = NSC * (int)sizeof(T) / 4;                    // dword lanes of the y / s piece of a stage
  static_assert(NB % FPG == 0, "pieces must not straddle k-steps");
  static_assert(KS >= 1 && KS * NB * 16 * 4 * (int)sizeof(T) == BLR_WAVE_STAGE_BYTES, "whole stages");
  static_assert(YL == 8 || YL == 16 || YL == 32, "y piece");
};

// The three phases are separate NOINLINE functions that hand their results over through LDS (as blr_fused_small.hpp does):
// in one function body the compiler keeps the 80 accumulator registers of the Gram loop alive across the fully unrolled
// elimination and spills ~400 SGPRs and ~50 VGPRs; split, each phase gets the whole register file to itself.

// ---- phase 1: streaming Gram.  Out: P = lower triangle of A = Lw + X S^-1 X' (packed), bvec = X S^-1 (y - X'mw),
//      scr[0] = quadratic form, scr[1] = logdet Sigma_y, iscr[0] = first bad variance (0x7fffffff: none)
// The wave keeps THREE 4 KiB stages in flight while it consumes a fourth (with one stage of look-ahead the HBM stream cost
// 0.47 ms of a 1.21 ms launch at config 4: every wave sat out a memory latency per stage).  Everything a stage needs -- X,
// y and, for diagonal noise, the variances -- travels by LDS-DMA issued from inline asm, so the only vector-memory
// operations in the loop are these pieces and the stage's arrival is a COUNTED s_waitcnt vmcnt(2 x pieces-per-stage).
// NW > 1: the observations of ONE regressor are split over the NW waves of the workgroup (contiguous runs of whole stages;
// each wave streams through its own ring in its own LDS slice `smem0 + w * LDS_BYTES`), the partial statistics of waves
// 1 .. NW-1 are handed to wave 0 through their slices and added in wave order -- a fixed order, so the result is bitwise
// reproducible -- and wave 0 alone continues with the factorisation.  Used for batches too small to give every wave slot of
// the chip a regressor of its own (config 4 sharded over 8 GPUs: 1024 regressors per GPU for 2048 wave slots).
template <typename T, int NB, bool DIAG, int NW>
BLR_PHASE void wave_gram(char* smem0, const BLR_GLOBAL T* X, int64_t ldx, const BLR_GLOBAL T* y, const BLR_GLOBAL T* s,
                         const BLR_GLOBAL T* mw, T dpr, int N, int prior_kind, const BLR_GLOBAL T* Lw, int64_t ldl) {
  using C = WaveCfg<T, NB>;
  using acc4 = typename Mfma<T>::acc4;
  constexpr int VEC = Mfma<T>::VEC;
  constexpr int PPS = C::NG + 1 + (DIAG ? 1 : 0);  // LDS-DMA instructions per stage
  const int w = (NW == 1) ? 0 : uni((int)(threadIdx.x >> 6));
  char* const smem = smem0 + w * C::LDS_BYTES;
  T* const ring = reinterpret_cast<T*>(smem);
  T* const P = reinterpret_cast<T*>(smem);  // after the loop
  T* const ybuf = reinterpret_cast<T*>(smem + C::OFF_Y);
  T* const sbuf = reinterpret_cast<T*>(smem + C::OFF_S);
  T* const wbuf = reinterpret_cast<T*>(smem + C::OFF_W);
  T* const bvec = reinterpret_cast<T*>(smem + C::OFF_B);
  double* const scr = reinterpret_cast<double*>(smem + C::OFF_SCR);
  int* const iscr = reinterpret_cast<int*>(smem + C::OFF_SCR + 32);
  int lane = threadIdx.x & 63;
  asm volatile("" : "+v"(lane));
  const int r16 = lane & 15, q4 = lane >> 4;
  N = uni(N);
  const unsigned voff = glds_lane_offset<T>(ldx, lane);
  const unsigned ring_addr = uni((int)lds_addr_of(ring));
  const unsigned ybuf_addr = uni((int)lds_addr_of(ybuf)), sbuf_addr = uni((int)lds_addr_of(sbuf));
  X = uni(X); y = uni(y); s = uni(s);

  T mwf[NB];
  bool mwz = true;
#pragma unroll
  for (int I = 0; I < NB; ++I) { mwf[I] = mw[16 * I + r16]; mwz = mwz && (mwf[I] == T(0)); }
  mwz = __all(mwz);
  T s_iso = DIAG ? T(1) : s[0];
  int bad_noise = (DIAG || s_iso > T(0)) ? 0x7fffffff : 1;
  // every compiler-visible load has been consumed before the loop: hipcc must not park an s_waitcnt vmcnt(0) inside it
#pragma unroll
  for (int I = 0; I < NB; ++I) asm volatile("" : "+v"(mwf[I]));
  asm volatile("" : "+v"(s_iso));

  acc4 acc[C::NT];
#pragma unroll
  for (int t = 0; t < C::NT; ++t) acc[t] = acc4{T(0), T(0), T(0), T(0)};
  double bacc[NB];
#pragma unroll
  for (int I = 0; I < NB; ++I) bacc[I] = 0.0;
  double qacc = 0.0, lacc = 0.0;
  const int nfull_all = N / C::NSC;  // whole stages; a ragged tail is handled after the pipeline has drained
  // this wave's run of whole stages [t0, t0 + nfull): the LAST waves take the remainder, wave 0 -- which also factorises --
  // the smallest share
  const int share = nfull_all / NW, rem = nfull_all - share * NW;
  const int t0 = (NW == 1) ? 0 : w * share + max(0, w - (NW - rem));
  const int nfull = (NW == 1) ? nfull_all : share + (w >= NW - rem ? 1 : 0);
  // Stages are issued strictly in order, so a piece's global address is a running 64-bit scalar plus a per-piece constant
  // (two scalar adds; the (n0 + 4 j) ldx multiplications per piece were a dozen scalar instructions each, in front of the MFMAs)
  uint64_t nextX = (uint64_t)(uintptr_t)(X + (int64_t)t0 * C::NSC * ldx);
  uint64_t nextY = (uint64_t)(uintptr_t)(y + (int64_t)t0 * C::NSC), nextS = (uint64_t)(uintptr_t)(s + (DIAG ? (int64_t)t0 * C::NSC : 0));
  const uint64_t stepX = (uint64_t)((int64_t)C::NSC * ldx * (int64_t)sizeof(T));
  uint64_t offg[C::NG];
#pragma unroll
  for (int g = 0; g < C::NG; ++g) {
    const int j = (g * C::FPG) / NB, I0 = (g * C::FPG) % NB;
    offg[g] = (uint64_t)(((int64_t)(4 * j) * ldx + 16 * I0) * (int64_t)sizeof(T));
  }
  auto issue = [&](int td) {     // exactly PPS LDS-DMA instructions; td only selects the ring slot
    const int sl = td & (C::DEPTH - 1);
    const unsigned slot_addr = ring_addr + (unsigned)(sl * C::SLOT * (int)sizeof(T));
#pragma unroll
    for (int g = 0; g < C::NG; ++g) glds_s<16, 64, BLR_WAVE_NT>(uni((int64_t)(nextX + offg[g])), voff, slot_addr + (unsigned)(g * 1024));
    glds_s<4, C::YL>(uni((int64_t)nextY), (unsigned)(lane * 4), ybuf_addr + (unsigned)(sl * C::NSC * (int)sizeof(T)));
    if constexpr (DIAG) glds_s<4, C::YL>(uni((int64_t)nextS), (unsigned)(lane * 4), sbuf_addr + (unsigned)(sl * C::NSC * (int)sizeof(T)));
    nextX += stepX;
    nextY += (uint64_t)(C::NSC * sizeof(T));
    if constexpr (DIAG) nextS += (uint64_t)(C::NSC * sizeof(T));
  };
  // one stage of compute: KS k-steps on the slot image, y from yb, weights from wbuf (DIAG)
  auto compute = [&](const T* slot, const T* yb, bool data) {
#pragma unroll
    for (int j = 0; j < C::KS; ++j) {
      T f[NB];
#pragma unroll
      for (int I = 0; I < NB; ++I) f[I] = slot[(j * NB + I) * 64 + lane];
      const T yv = yb[4 * j + q4];
      T w = T(1);
      if constexpr (DIAG) w = wbuf[4 * j + q4];
      T fa[NB];
#pragma unroll
      for (int I = 0; I < NB; ++I) fa[I] = DIAG ? f[I] * w : f[I];
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int K = 0; K <= I; ++K) {
#if BLR_EXP == 2
          acc[I * (I + 1) / 2 + K][0] += fa[I] * f[K];
#else
          acc[I * (I + 1) / 2 + K] = Mfma<T>::mma(fa[I], f[K], acc[I * (I + 1) / 2 + K]);
#endif
        }
      // column-vector work on the same registers: delta_n, b += x_n w_n delta_n, quadratic form   (:82-84, :57)
#if BLR_EXP == 3
      bacc[0] += (double)w + (double)yv; continue;
#endif
      T mu = T(0);
      if (!mwz) {
#pragma unroll
        for (int I = 0; I < NB; ++I) mu += f[I] * mwf[I];
        mu = row16_allreduce(mu);
      }
      const T delta = data ? yv - mu : T(0);  // pseudo-observations of a factor prior enter the Gram matrix only
      const T rn = DIAG ? delta * w : delta;
      if (r16 == 0) qacc += (double)delta * (double)rn;
#pragma unroll
      for (int I = 0; I < NB; ++I) bacc[I] += (double)f[I] * (double)rn;
    }
  };
  // variances of the stage in sbuf -> weights in wbuf, log-determinant, positivity (reference :79-84)
  auto weights = [&](const T* sb, int n0) {
    if (lane < C::NSC) {
      const T sv = sb[lane];
      wbuf[lane] = T(1) / sv;
      lacc += log((double)sv);
      if (!(sv > T(0))) bad_noise = min(bad_noise, n0 + lane + 1);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };

  prior_kind = uni(prior_kind);
  if (w == 0 && prior_kind == PRIOR_UPPER_FACTOR) {
    // a carried-forward factor U (PDMat prior, blr_update_factor_*): U'U = sum_j u_j u_j' with u_j = row j of U, i.e. D
    // pseudo-observation columns with element (d, j) = U[j + d ldl] for j <= d.  Isotropic noise accumulates X X' unscaled and
    // divides by s once at the end, so the pseudo-columns carry sqrt(s) here.  A handful of stages, loaded synchronously.
    const T pscale = DIAG ? T(1) : sqrt(s_iso);
    for (int j0 = 0; j0 < C::DP; j0 += C::NSC) {
      T* slot = ring;
      for (int e = lane; e < C::SLOT; e += 64) {
        const int F = e >> 6, l = e & 63;          // fragment F = kstep * NB + I, lane l = (r, q)
        const int jj = F / NB, I = F - jj * NB;
        const int d = 16 * I + (l & 15), j = j0 + 4 * jj + (l >> 4);
        slot[e] = (j <= d) ? Lw[(int64_t)d * ldl + j] * pscale : T(0);
      }
      if (lane < C::NSC) { ybuf[lane] = T(0); if constexpr (DIAG) wbuf[lane] = T(1); }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      compute(slot, ybuf, false);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
  }
#if BLR_EXP != 4
  for (int td = 0; td < C::DEPTH - 1 && td < nfull; ++td) issue(td);
#endif
#pragma unroll 1
  for (int t = 0; t < nfull; ++t) {
    const int sl = t & (C::DEPTH - 1);
    const int rem = nfull - 1 - t;  // stages issued after stage t: min(rem, DEPTH - 2)
    if (rem >= C::DEPTH - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((C::DEPTH - 2) * PPS) : "memory");
    else if constexpr (C::DEPTH > 4) {
      if (rem >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PPS) : "memory");
      else if (rem == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PPS) : "memory");
      else if (rem == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPS) : "memory");
      else if (rem == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      if (rem == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_wave_barrier();
#if BLR_EXP != 4
    if (t + C::DEPTH - 1 < nfull) issue(t + C::DEPTH - 1);  // into the slot stage t - 1 was read from
#endif
    if constexpr (DIAG) weights(sbuf + sl * C::NSC, (t0 + t) * C::NSC);
#if BLR_EXP == 5  // streaming only: what the memory system delivers to this access pattern
    qacc += (double)ring[sl * C::SLOT + lane];
#elif BLR_EXP == 6 || BLR_EXP == 7  // streaming + an idle pause as long as a stage's MFMAs (6: one wave's 1280 cycles, 7: two waves' 2560 -- the pipe is shared)
    qacc += (double)ring[sl * C::SLOT + lane];
#pragma unroll 1
    for (int z = 0; z < (BLR_EXP == 6 ? 20 : 40); ++z) __builtin_amdgcn_s_sleep(1);
#else
    compute(ring + sl * C::SLOT, ybuf + sl * C::NSC, true);
#endif
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  if (w == NW - 1 && nfull_all * C::NSC < N) {
    // ragged tail (N not a multiple of the stage width): loaded synchronously, lane by lane, zeros beyond N
    typedef T vecT __attribute__((ext_vector_type(Mfma<T>::VEC)));
    const int n0 = nfull_all * C::NSC;
    T* slot = ring;
    const int e0 = lane * VEC;
    const int fl = e0 >> 6, ls = e0 & 63, q = ls >> 4, r = ls & 15;
#pragma unroll
    for (int g = 0; g < C::NG; ++g) {
      const int F = g * C::FPG + fl;
      const int j = F / NB, I = F - j * NB;
      const int n = n0 + 4 * j + q, d = 16 * I + r;
      T* dst = slot + g * (C::FPG * 64);
      if (n < N) {
        glds16((const BLR_GLOBAL void*)(X + (int64_t)n * ldx + d), dst);
      } else {
        vecT z;
#pragma unroll
        for (int c = 0; c < VEC; ++c) z[c] = T(0);
        *reinterpret_cast<vecT*>(dst + e0) = z;
      }
    }
    if (lane < C::NSC) {
      const bool valid = n0 + lane < N;
      ybuf[lane] = valid ? y[n0 + lane] : T(0);
      if constexpr (DIAG) {
        const T sv = valid ? s[n0 + lane] : T(1);
        wbuf[lane] = valid ? T(1) / sv : T(0);
        if (valid) {
          lacc += log((double)sv);
          if (!(sv > T(0))) bad_noise = min(bad_noise, n0 + lane + 1);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    compute(slot, ybuf, true);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }

  // ---- reductions: b (over the 4 column groups of the lanes), quadratic form, logdet Sigma_y, noise check
  const T w_iso = T(1) / s_iso;
  double quad = wave_allreduce(qacc);
  double lsum = DIAG ? wave_allreduce(lacc) : 0.0;
  {
    int bn = bad_noise;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) bn = min(bn, __shfl_xor(bn, off));
    bad_noise = bn;
  }
  double bsum[NB];
#pragma unroll
  for (int I = 0; I < NB; ++I) {
    double v = bacc[I];
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    bsum[I] = v;
  }
  if constexpr (NW > 1) {
    double* const part = reinterpret_cast<double*>(smem + C::OFF_PART);
    if (w != 0) {
      // hand the raw partials over: G (lower triangle, packed like P), b, quadratic form, logdet, first bad variance
#pragma unroll
      for (int I = 0; I < NB; ++I) {
        if (q4 == 0) part[16 * I + r16] = bsum[I];
#pragma unroll
        for (int K = 0; K <= I; ++K) {
          const int t = I * (I + 1) / 2 + K;
          const int col = 16 * K + r16;
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int row = 16 * I + Mfma<T>::crow(lane, v);
            if (col <= row) P[pidx(row, col)] = acc[t][v];
          }
        }
      }
      if (lane == 0) {
        part[C::DP] = quad;
        part[C::DP + 1] = lsum;
        reinterpret_cast<int*>(part + C::DP + 2)[0] = bad_noise;
      }
      __syncthreads();  // A: partials visible to wave 0
      __syncthreads();  // B: wave 0 has consumed them -- the slice is free for the next regressor
      return;
    }
    __syncthreads();    // A
#pragma unroll
    for (int ww = 1; ww < NW; ++ww) {  // wave order: fixed
      const double* pw = reinterpret_cast<const double*>(smem0 + ww * C::LDS_BYTES + C::OFF_PART);
#pragma unroll
      for (int I = 0; I < NB; ++I) bsum[I] += pw[16 * I + r16];
      quad += pw[C::DP];
      lsum += pw[C::DP + 1];
      bad_noise = min(bad_noise, reinterpret_cast<const int*>(pw + C::DP + 2)[0]);
    }
  }
  const double logdet_Sy = DIAG ? lsum : (double)N * log((double)s_iso);
  if (!DIAG) quad *= (double)w_iso;
#pragma unroll
  for (int I = 0; I < NB; ++I) {
    double v = bsum[I];
    if (!DIAG) v *= (double)w_iso;
    if (q4 == 0) bvec[16 * I + r16] = (T)v;  // b = X S^-1 (y - X'mw)
  }
  if (lane == 0) { scr[0] = quad; scr[1] = logdet_Sy; iscr[0] = bad_noise; }
  // A = Lw + (1/s) X X'  (isotropic) -- the prior sits on the diagonal of the diagonal tiles -- -> packed lower triangle
#pragma unroll
  for (int I = 0; I < NB; ++I) {
    const T dvI = __shfl(dpr, 16 * I + r16);  // Lw[16 I + r] (diagonal prior): lane (r, q) holds column r of the tile
#pragma unroll
    for (int K = 0; K <= I; ++K) {
      const int t = I * (I + 1) / 2 + K;
      const int col = 16 * K + r16;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int rl = Mfma<T>::crow(lane, v);
        const int row = 16 * I + rl;
        T g = acc[t][v];
        if constexpr (NW > 1) {
          if (col <= row) {
#pragma unroll
            for (int ww = 1; ww < NW; ++ww) g += reinterpret_cast<const T*>(smem0 + ww * C::LDS_BYTES)[pidx(row, col)];
          }
        }
        T val = DIAG ? g : g * w_iso;
        if (prior_kind == PRIOR_DIAGONAL) {
          if (I == K && rl == r16) val += dvI;
        } else if (prior_kind == PRIOR_DENSE) {
          if (col <= row) val += Lw[(int64_t)row * ldl + col];  // UPPER triangle of the caller's matrix, as LAPACK 'U'
        }
        if (col <= row) P[pidx(row, col)] = val;
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if constexpr (NW > 1) __syncthreads();  // B
}

// ---- phase 2: blocked Cholesky of P, one row per lane, trailing matrix in the accumulators (see phase_chol), with the
//      forward substitution u = L^-1 b riding along.  Out: P = L, bvec = u.  Returns 0 or the failing leading minor.
template <typename T, int NB>
BLR_PHASE int wave_chol(char* smem) {
  using C = WaveCfg<T, NB>;
  using acc4 = typename Mfma<T>::acc4;
  T* const P = reinterpret_cast<T*>(smem);
  T* const bvec = reinterpret_cast<T*>(smem + C::OFF_B);
  int lane = threadIdx.x & 63;  // also called by waves 1 .. NW-1 of a split workgroup (dense prior), each in its own slice
  asm volatile("" : "+v"(lane));
  const int r16 = lane & 15, q4 = lane >> 4;
  constexpr int D = C::DP;
  // trailing tiles (I, K), K >= 1, from P (diagonal tiles: both halves, so the tile is symmetric)
  acc4 acc[C::NT];
#pragma unroll
  for (int I = 1; I < NB; ++I)
#pragma unroll
    for (int K = 1; K <= I; ++K) {
      const int t = I * (I + 1) / 2 + K;
      const int col = 16 * K + r16;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int row = 16 * I + Mfma<T>::crow(lane, v);
        acc[t][v] = P[pidx(max(row, col), min(row, col))];
      }
    }
  // (round 5: the panel step as in phase_chol -- rows loaded unmasked (entries right of the diagonal of a diagonal-block row are dead
  // values), predicated stores through an address select to a dump word instead of an exec-mask branch each, the elimination as one
  // basic block with the next pivot's reciprocal started early and the 16 reciprocal square roots taken once after the loop, fragment
  // addresses without per-element triangle arithmetic)
  const int pr = (r16 * (r16 + 1)) >> 1;  // pidx(r16, 0)
  T* const dummy = reinterpret_cast<T*>(smem + C::OFF_Y) + r16;  // (the y ring is dead here)
  int cr[4], pcr[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    cr[v] = Mfma<T>::crow(lane, v);
    pcr[v] = (cr[v] * (cr[v] + 1)) >> 1;
  }
  int info = 0;
#pragma unroll 1
  for (int J = 0; J < NB; ++J) {
    const bool is_diag = lane < 16;
    const int ri = is_diag ? 16 * J + lane : 16 * (J + 1) + (lane - 16);
    const bool active = ri < D;
    const int ria = active ? ri : 0;
    T* const rowp = P + (((ria * (ria + 1)) >> 1) + 16 * J);
    T arow[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) arow[c] = rowp[c];
    T bl = bvec[ria];
    T own_rsq;
    {
      T d2 = readlane(arow[0], 0);
      T rc = fast_rcp(d2);
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        if (!(d2 > T(0)) && info == 0) info = 16 * J + c + 1;  // wave-uniform
        const T tm = arow[c] * rc;
        if (c + 1 < 16) {
          arow[c + 1] = fused_madd(-tm, readlane(arow[c], c + 1), arow[c + 1]);
          d2 = readlane(arow[c + 1], c + 1);
          rc = fast_rcp(d2);
        }
        const T bc = readlane(bl, c);
        if (lane > c) bl = fused_madd(-tm, bc, bl);
#pragma unroll
        for (int k = c + 2; k < 16; ++k) arow[k] = fused_madd(-tm, readlane(arow[c], k), arow[k]);
      }
      T piv = T(1);
#pragma unroll
      for (int c = 0; c < 16; ++c) piv = (lane == c) ? arow[c] : piv;
      own_rsq = fast_rsqrt(piv);
#pragma unroll
      for (int c = 0; c < 16; ++c) arow[c] *= readlane(own_rsq, c);  // L[i][c] = a_ic / sqrt(d2); lane c: d2 / sqrt(d2)
    }
    if (is_diag) bl *= own_rsq;
    if (info != 0) break;
    {
      const int lim = active ? (is_diag ? lane : 15) : -1;  // columns 0 .. lim of this lane's row are stored
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        T* dst = (c <= lim) ? rowp + c : dummy;
        *dst = arow[c];
      }
      T* bd = active ? bvec + ri : dummy;
      *bd = bl;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // trailing update from the finished panel; block column J + 1 goes straight back to P
    const int pc = pr + 16 * J + q4;
#pragma unroll
    for (int I = 1; I < NB; ++I)
#pragma unroll
      for (int K = 1; K <= I; ++K) {
        if (K <= J) continue;  // wave-uniform
        const int t = I * (I + 1) / 2 + K;
        const T* pI = P + ((128 * I * I + 8 * I) + (16 * I) * r16 + pc);  // pidx(16 I + r16, 16 J + q4)
        const T* pK = P + ((128 * K * K + 8 * K) + (16 * K) * r16 + pc);
        T fa[4], fb[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { fa[ks] = pI[4 * ks]; fb[ks] = pK[4 * ks]; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc[t] = Mfma<T>::mma(-fa[ks], fb[ks], acc[t]);
        if (K == J + 1) {
          const int sb = (128 * I * I + 8 * I) + 16 * K + r16;
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            T* dst = (I != K || r16 <= cr[v]) ? P + (sb + (16 * I) * cr[v] + pcr[v]) : dummy;
            *dst = acc[t][v];
          }
        }
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  return info;
}

// ---- phase 3: m = L^-T u (column-oriented, 8 pivots per block of prefetched rows), |u|^2, logdet A.  In: P = L, bvec = u.
//      Out: bvec = m, scr[2] = |u|^2, scr[3] = logdet A.  A function of its own so that its 2 x DP lane masks (lane < k, lane == k for
//      every pivot of the unrolled loop) are made where they are used: inlined in the kernel, hipcc hoisted them out of the loop over
//      the regressors and parked 256 scalar registers in VGPR lanes -- the kernel's "348 SGPR spills".
template <typename T, int NB>
BLR_PHASE void wave_backsolve(char* smem) {
  using C = WaveCfg<T, NB>;
  T* const P = reinterpret_cast<T*>(smem);
  T* const bvec = reinterpret_cast<T*>(smem + C::OFF_B);
  double* const scr = reinterpret_cast<double*>(smem + C::OFF_SCR);
  int lane = threadIdx.x & 63;
  asm volatile("" : "+v"(lane));
  constexpr int D = C::DP;
  const bool in = lane < D;
  T b0 = in ? bvec[lane] : T(0);
  const T lii = in ? P[pidx(lane, lane)] : T(1);
  const T r0 = fast_rcp(lii);
  const double uu = wave_allreduce((double)b0 * (double)b0);
  const double logdetA = 2.0 * wave_allreduce(in ? log((double)lii) : 0.0);
  for (int kb = D - 1; kb >= 0; kb -= 8) {
    T row[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = kb - u;
      row[u] = (k >= 0 && lane < k) ? P[pidx(k, 0) + lane] : T(0);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = kb - u;
      if (k >= 0) {
        const T mk = readlane(b0, k) * readlane(r0, k);
        if (lane == k) b0 = mk;
        b0 -= row[u] * mk;
      }
    }
  }
  if (in) bvec[lane] = b0;
  if (lane == 0) { scr[2] = uu; scr[3] = logdetA; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

#ifdef BLR_WAVE_CLK
__device__ unsigned long long g_waveclk[2];
#endif
template <typename T, int NB, int NW = 1>
__global__ __launch_bounds__(64 * NW, 2) void fused_wave_kernel(PosteriorArgs<T> a_kernarg) {
  using C = WaveCfg<T, NB>;
  // The arguments are READ AGAIN from the kernarg segment after every phase call instead of being kept: the phases clobber every
  // scalar register (no callee-saved registers, BLR_PHASE), and hipcc kept some forty 64-bit fields and what it derived from them
  // alive across the calls in VGPR lanes -- 348 v_writelane + as many v_readlane among the kernel's 3400 instructions, every regressor.
  // A scalar load from the kernarg segment hits the constant cache.
  typedef const __attribute__((address_space(4))) PosteriorArgs<T>* ArgPtr;
  const ArgPtr ap0 = (ArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
  ArgPtr ap = ap0;
#define a (*ap)
#define BLR_FORGET_ARGS() do { unsigned z__ = 0; asm volatile("" : "+s"(z__)); ap = reinterpret_cast<ArgPtr>(reinterpret_cast<const __attribute__((address_space(4))) char*>(ap0) + z__); } while (0)
  extern __shared__ __attribute__((aligned(16))) char smem_all[];
  // NW > 1: every wave works in its own slice; waves 1 .. NW-1 only take part in the prior check (redundantly: every early
  // exit below must be taken by ALL waves, the Gram phase contains workgroup barriers) and in the Gram phase
  const int wv = (NW == 1) ? 0 : uni((int)(threadIdx.x >> 6));
  char* const smem = smem_all + wv * C::LDS_BYTES;
  T* const P = reinterpret_cast<T*>(smem);
  T* const bvec = reinterpret_cast<T*>(smem + C::OFF_B);
  double* const scr = reinterpret_cast<double*>(smem + C::OFF_SCR);
  int* const iscr = reinterpret_cast<int*>(smem + C::OFF_SCR + 32);
  const int lane = threadIdx.x & 63;
  constexpr int D = C::DP;
  const double kNaN = __longlong_as_double(0x7ff8000000000000LL);

#ifdef BLR_WAVE_CLK  /* tools/fused_bench.hip: the clock the part holds under this kernel (shader cycles against the 100 MHz counter) */
  const unsigned long long wck0 = __builtin_amdgcn_s_memtime(), wrt0 = __builtin_amdgcn_s_memrealtime();
#endif
  for (int64_t reg = blockIdx.x; reg < a.B; reg += gridDim.x) {
    const BLR_GLOBAL T* mw = as_global(a.mw + reg * a.stridemw);
    const BLR_GLOBAL T* Lw = as_global(a.Lw + reg * a.strideLw);

    // ---- prior (reference :78): positive definite, logdet.  Diagonal: entries; carried-forward factor: its diagonal;
    //      dense: the blocked Cholesky below on a copy in the packed triangle
    T dpr = T(1);
    double logdet_Lw = 0.0;
    int info = 0;
    if (a.prior_kind == PRIOR_DENSE) {
      for (int c = 0; c < D; ++c)
        if (lane <= c) P[pidx(c, lane)] = Lw[(int64_t)c * a.ldl + lane];  // upper entry (lane, c) -> lower (c, lane)
      if (lane < D) bvec[lane] = T(0);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      info = wave_chol<T, NB>(smem);
      BLR_FORGET_ARGS();
      logdet_Lw = 2.0 * wave_allreduce((info == 0 && lane < D) ? log((double)P[pidx(lane, lane)]) : 0.0);
      __builtin_amdgcn_wave_barrier();
    } else {
      const bool in = lane < D;
      if (in) dpr = (a.prior_kind == PRIOR_DIAGONAL) ? Lw[lane] : Lw[(int64_t)lane * a.ldl + lane];
      const bool ok = !in || (dpr > T(0));
      const unsigned long long badm = __ballot(!ok);
      if (badm) info = __ffsll((long long)badm);
      logdet_Lw = wave_allreduce((in && ok) ? log((double)dpr) : 0.0);
      if (a.prior_kind != PRIOR_DIAGONAL) logdet_Lw *= 2.0;
    }
    if (info != 0) {  // uniform over the workgroup: every wave ran the same check on the same prior
      if (wv == 0 && lane == 0) { a.info[reg] = info; if (a.logpdf) a.logpdf[reg] = kNaN; }
      continue;
    }
    if (a.noise_kind == NOISE_DIAGONAL)
      wave_gram<T, NB, true, NW>(smem_all, as_global(a.X + reg * a.strideX), a.ldx, as_global(a.y + reg * a.stridey),
                                 as_global(a.s + reg * a.strides), mw, dpr, a.N, a.prior_kind, Lw, a.ldl);
    else
      wave_gram<T, NB, false, NW>(smem_all, as_global(a.X + reg * a.strideX), a.ldx, as_global(a.y + reg * a.stridey),
                                  as_global(a.s + reg * a.strides), mw, dpr, a.N, a.prior_kind, Lw, a.ldl);
    BLR_FORGET_ARGS();
    if (wv != 0) continue;  // the factorisation is wave 0's
    const double quad = scr[0], logdet_Sy = scr[1];
    const int N = a.N;
    if (iscr[0] != 0x7fffffff) {  // Sigma_y is not positive definite: PosDefException(index), as :79 would throw
      if (lane == 0) { a.info[reg] = iscr[0]; if (a.logpdf) a.logpdf[reg] = kNaN; }
      __builtin_amdgcn_wave_barrier();
      continue;
    }
#if BLR_EXP >= 1 && BLR_EXP <= 7
    if (lane == 0) { a.info[reg] = 0; if (a.logpdf) a.logpdf[reg] = quad + (double)P[lane]; }
    continue;
#endif
    if (a.Lw_post) {  // posterior precision Lw' = A, full symmetric (:92)
      T* out = a.Lw_post + reg * a.strideLp;
      for (int c = 0; c < D; ++c)
        if (lane < D) out[(int64_t)c * a.ldlp + lane] = (lane >= c) ? P[pidx(lane, c)] : P[pidx(c, lane)];
    }
    info = wave_chol<T, NB>(smem);
    BLR_FORGET_ARGS();
    if (info != 0) {
      if (lane == 0) { a.info[reg] = info; if (a.logpdf) a.logpdf[reg] = kNaN; }
      __builtin_amdgcn_wave_barrier();
      continue;
    }
    if (a.T_post) {  // T = L' (upper, column-major), strictly-lower part zero
      T* out = a.T_post + reg * a.strideT;
      for (int c = 0; c < D; ++c)
        if (lane < D) out[(int64_t)c * a.ldt + lane] = (lane <= c) ? P[pidx(c, lane)] : T(0);
    }

    // ---- m = L^-T u, |u|^2, logdet A: a phase of its own (see wave_backsolve)
    wave_backsolve<T, NB>(smem);
    BLR_FORGET_ARGS();
    {
      const bool in = lane < D;
      if (a.mw_post && in) a.mw_post[reg * a.stride_mwpost + lane] = as_global(a.mw + reg * a.stridemw)[lane] + bvec[lane];  // :68
      if (lane == 0) {
        a.info[reg] = 0;
        if (a.logpdf) {
          const double LOG2PI = 1.8378770664093454835606594728112;
          a.logpdf[reg] = -0.5 * ((double)N * LOG2PI + logdet_Sy + quad + scr[3] - logdet_Lw - scr[2]);  // :84 + :57
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // P / ring reuse by the next regressor
  }
#ifdef BLR_WAVE_CLK
  if ((blockIdx.x & 63) == 0 && threadIdx.x == 0) {
    atomicAdd(&g_waveclk[0], __builtin_amdgcn_s_memtime() - wck0);
    atomicAdd(&g_waveclk[1], __builtin_amdgcn_s_memrealtime() - wrt0);
  }
#endif
#undef a
#undef BLR_FORGET_ARGS
}

}  // namespace blr
