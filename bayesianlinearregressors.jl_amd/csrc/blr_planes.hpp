// fp32 Gram matrices at D > 128 (configs 3 and 5) from PRE-SPLIT operands: the bf16 x 3 representation of the design matrix is made
// ONCE per element, in the fragment order of the bf16 matrix instruction, and the Gram launch is nothing but LDS-DMA, fragment reads
// and v_mfma_f32_32x32x16_bf16.
//
// Why (round 5, gram_tile_kernel<float, true>): an fp32 number is exactly three bf16 numbers, and the six products hh, hm, mh, mm, hl,
// lh under fp32 accumulation are as accurate as an fp32 fma chain (tools/bf3_unit.hip) at 14 x the rate of v_mfma_f32_16x16x4_f32 --
// but that kernel split its operands inside the matrix loop, once per macro tile that touches an element (8 x at config 3, 16 x at
// config 5): ~ 290 vector instructions per 16 columns and wave next to 24 matrix instructions, matrix pipe 37 % busy.  Here:
//   * planes_kernel (one pass over X, or over the raw inputs of a random-Fourier basis -- reference src/basis_function_regression.jl:41
//     materialises phi(x); here phi is evaluated once per element and leaves as planes, the fp32 feature matrix never exists):
//       z_dn = x_dn sqrt(w_n)   (w_n = 1 / s_n under diagonal noise, 1 otherwise: G = sum_n w_n x_n x_n' = Z Z', both operands the same)
//       h = bf16(z), m = bf16(z - h), l = bf16(z - h - m), each rounded to nearest
//     stored as Xp[k-block of 16 columns][row block of 128][32-row sub-block j][plane p][lane][8 bf16]: one KiB per (j, p) is ONE matrix
//     operand (lane = row r of the sub-block + 32 x (columns 8 .. 15)), twelve consecutive KiB are one side of a macro tile's half.
//     The same pass accumulates b = X r (r = delta / s, reference :57) in fp64 per row and column chunk: the Gram launch carries no
//     right-hand side.
//   * gram_planes_kernel: one 512-thread workgroup per CU and (macro tile, column range); a ring of SIX halves of 24 KiB (A side + B side,
//     12 pieces of 1 KiB each, issued five halves ahead: the operands of the next ~ 4 k cycles are in flight); wave (i, c) owns the two
//     32 x 32 tiles (i, 2c), (i, 2c + 1) of the 128 x 128 macro tile: per half 9 fragment reads of 16 bytes per lane and 12 matrix
//     instructions (six products per tile, smallest terms first).  A diagonal macro tile computes its tiles with column block <= row
//     block (10 of 16) from the A side alone.  Split-K partial tiles in the layout of gram_tile_kernel: gram_reduce_kernel is unchanged.
#pragma once
#include "blr_large.hpp"

namespace blr {

constexpr int kPlanesThreads = 512;
constexpr int kPlanesHalf = 24 * 1024;   // bytes of one half in the ring: A side 12 KiB + B side 12 KiB
constexpr int kPlanesSlots = 6;
constexpr int kPlanesAhead = kPlanesSlots - 1;  // halves in flight beyond the one being computed
constexpr int kPlanesLds = kPlanesSlots * kPlanesHalf;
static_assert(kPlanesLds <= 160 * 1024, "LDS of one CU");

// ---- the producer --------------------------------------------------------------------------------------------------------------------
struct PlanesArgs {
  // source 0: X (D x N, ColVecs, element (d, n) at X[d + n ldx]); source 1: a random-Fourier basis phi_f(x_n) = scale cos(Omega_f' x_n + phase_f)
  const float* X; int64_t ldx;
  const float* Xin; int64_t ldxin; const float* Omega; int64_t ldo; const float* phase; float scale; int Din;
  const float* wsq;     // [N] sqrt(1 / s_n) (diagonal noise) or NULL
  const float* r;       // [N] delta_n / s_n for b = X r, or NULL
  unsigned short* Xp;   // planes (layout above)
  double* bpart;        // [nchunks][NC][128] partial sums of b (fixed order: one chunk, one writer)
  int D, N, NC, NKB, nchunks;
  int64_t grp_X, grp_ws;  // blockIdx.z = regressor of a group: element stride of X, byte stride of wsq / r / Xp / bpart
};

__device__ __forceinline__ float rff_feature(const PlanesArgs& a, const float* __restrict__ om /* Omega_f */, float ph, const float* __restrict__ xs /* LDS: [Din][16] */,
                                             int col) {
  float acc = ph;
  for (int k = 0; k < a.Din; ++k) acc = __builtin_fmaf(om[k], xs[k * 16 + col], acc);
  return a.scale * rff_cos(acc);
}

constexpr int kPlanesChunkKb = 64;  // k-blocks per column chunk at most (the chunk's r and sqrt(w) live in LDS: 2 x 4 KiB)

template <bool RFF>
__global__ __launch_bounds__(kThreads) void planes_kernel(PlanesArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* const rs = reinterpret_cast<float*>(smem);            // the chunk's r_n          [16 per]
  float* const wsm = rs + 16 * kPlanesChunkKb;                 // the chunk's sqrt(w_n)    [16 per] (1 without weights, 0 beyond N)
  float* const xs = wsm + 16 * kPlanesChunkKb;                 // RFF: the k-block's raw inputs [Din][16]
  const int tid = threadIdx.x, lane = tid & 63, j = tid >> 6;  // wave j: rows 32 j .. 32 j + 31 of row block I
  const int I = blockIdx.y;
  if (const int64_t g = blockIdx.z) {
    if (!RFF) a.X += g * a.grp_X;
    a.wsq = ws_shift(a.wsq, g * a.grp_ws); a.r = ws_shift(a.r, g * a.grp_ws); a.Xp = ws_shift(a.Xp, g * a.grp_ws);
    a.bpart = ws_shift(a.bpart, g * a.grp_ws);
  }
  const int per = (a.NKB + a.nchunks - 1) / a.nchunks;  // (<= kPlanesChunkKb: the host sizes nchunks)
  const int kb0 = blockIdx.x * per, kb1 = min(a.NKB, kb0 + per);
  const int r32 = lane & 31, kh = lane >> 5;
  const int row = I * kPB + 32 * j + r32;
  const bool row_ok = row < a.D;
  // the chunk's per-column scalars once, through LDS (as loads inside the loop they were a dependent L2 round trip per k-block: 234 us
  // for config 3's 671 MB)
  for (int c = tid; c < 16 * (kb1 - kb0); c += kThreads) {
    const int n = 16 * kb0 + c;
    rs[c] = (a.r && n < a.N) ? a.r[n] : 0.f;
    wsm[c] = n < a.N ? (a.wsq ? a.wsq[n] : 1.f) : 0.f;
  }
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
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int n = 16 * kb + 8 * kh + e;
        dst[e] = (row_ok && kb < kb1 && n < a.N) ? a.X[(int64_t)n * a.ldx + row] : 0.f;
      }
    }
  };
  fetch(kb0, xa);
  fetch(kb0 + 1, xb);
  for (int kb = kb0; kb < kb1; ++kb) {
    float x[8];
    if constexpr (RFF) {
      __syncthreads();
      for (int idx = tid; idx < a.Din * 16; idx += kThreads) {
        const int k = idx % a.Din, c = idx / a.Din;  // consecutive threads: consecutive input dimensions of one column (contiguous in memory)
        const int n = 16 * kb + c;
        xs[k * 16 + c] = n < a.N ? a.Xin[(int64_t)n * a.ldxin + k] : 0.f;
      }
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = (row_ok && 16 * kb + 8 * kh + e < a.N) ? rff_feature(a, om, ph, xs, 8 * kh + e) : 0.f;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) { x[e] = xa[e]; xa[e] = xb[e]; }
      fetch(kb + 2, xb);  // in flight while this k-block and the next are split and stored
    }
    float z[8];
    const float* rk = rs + 16 * (kb - kb0) + 8 * kh;
    const float* wk = wsm + 16 * (kb - kb0) + 8 * kh;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      bacc += (double)x[e] * (double)rk[e];  // (exact products, fp64 sum: as the Gram kernels' b partials)
      z[e] = x[e] * wk[e];
    }
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
    gram_u4* dst = reinterpret_cast<gram_u4*>(reinterpret_cast<char*>(a.Xp) + (((int64_t)kb * a.NC + I) * 12 + 3 * j) * 1024) + lane;
    dst[0] = H; dst[64] = M; dst[128] = L;  // three KiB, each written by one wave instruction
  }
  if (a.bpart) {
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
  int xcd_swizzle;
  int64_t grp_ws, grp_s;   // blockIdx.y = regressor of a group: byte stride of Xp / Gpart, element stride of s_iso
};

__global__ __launch_bounds__(kPlanesThreads, 2) void gram_planes_kernel(GramPlanesArgs a) {
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
    a.Xp = ws_shift(a.Xp, g * a.grp_ws); a.Gpart = ws_shift(a.Gpart, g * a.grp_ws);
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

  // Two accumulators per tile: the product of the leading planes (h h, magnitude 1) in one, the five small products (2^-8 .. 2^-16 of
  // it) in the other.  Every matrix instruction rounds its accumulator once; with all six in one register the sum picked up six
  // roundings of ITS magnitude per 16 columns -- 1.4e-6 of max |A| over the 74 halves of a column range at config 3's shape reduced to
  // N = 8192, 4.04 x the error of fp32 LAPACK -- now one (the small sum's roundings are 2^-7 of that).
  gram_f16v acc[2], accs[2];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int v = 0; v < 16; ++v) { acc[k][v] = 0.f; accs[k][v] = 0.f; }
  // one product of the six, into the accumulator of its class
#define BLR_PM(ACC, XP, YP) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gram_bf8, XP), __builtin_bit_cast(gram_bf8, YP), ACC, 0, 0, 0)
  // which of the wave's two tiles exist: a diagonal macro tile keeps column block <= row block
  const bool t0 = !diag || 2 * tc <= ti, t1 = !diag || 2 * tc + 1 <= ti;

  // LDS-DMA: waves 0 - 3 bring the A side (pieces 3 w .. 3 w + 2 of its 12), waves 4 - 7 the B side (none for a diagonal macro tile);
  // three pieces per wave and half, always issued (beyond the last half the last one is fetched again into a slot nobody reads: no
  // branch in the loop, one constant in the wait)
  const bool loader = wave < 4 || !diag;
  const int side = wave >> 2, p0 = 3 * (wave & 3);
  unsigned ring_addr = lds_addr_of(smem);
  asm volatile("" : "+v"(ring_addr));
  const uint64_t kbstep = (uint64_t)a.NC * 12u * 1024u;
  uint64_t next = (uint64_t)(uintptr_t)a.Xp + ((uint64_t)kb0 * a.NC + (uint64_t)(side ? J : I)) * 12u * 1024u + (uint64_t)p0 * 1024u;
  const unsigned voff = (unsigned)lane * 16u;
  int hi = 0;  // half the next issue belongs to
  // piece c (0 .. 2) of the half being issued; piece 2 moves on to the next half
  auto issue_piece = [&](auto ctag) {
    constexpr int c = decltype(ctag)::value;
#if defined(BLR_PLANES_EXP) && BLR_PLANES_EXP == 2  /* timing experiment: no LDS-DMA beyond the prologue */
    if (loader && hi < kPlanesAhead) {
#else
    if (loader) {
#endif
      const unsigned slot = ring_addr + (unsigned)((hi % kPlanesSlots) * kPlanesHalf + side * (kPlanesHalf / 2) + p0 * 1024);
      glds_s<16>(uni((int64_t)(next + (uint64_t)c * 1024u)), voff, slot + (unsigned)c * 1024u);
    }
    if constexpr (c == 2) {
      if (hi + 1 < nh) next += kbstep;
      ++hi;
    }
  };
  auto issue = [&]() {
    issue_piece(std::integral_constant<int, 0>{}); issue_piece(std::integral_constant<int, 1>{}); issue_piece(std::integral_constant<int, 2>{});
  };
  // the wave's nine operand fragments of a half: one row sub-block of the A side, two of the B side (the A side again on a diagonal tile)
  struct Frags { gram_u4 v[9]; };  // A.h A.m A.l | B0.h B0.m B0.l | B1.h B1.m B1.l
  const int offa = (ti * 3) * 1024 + lane * 16, offb = (diag ? 0 : kPlanesHalf / 2) + (2 * tc * 3) * 1024 + lane * 16;
  auto read_frag = [&](const char* slot, Frags& f, auto itag) {
    constexpr int i = decltype(itag)::value;
    f.v[i] = *reinterpret_cast<const gram_u4*>(slot + (i < 3 ? offa + i * 1024 : offb + (i - 3) * 1024));
  };
  // One half: nine fragment reads, the three LDS-DMA pieces of half h + 5, twelve matrix instructions -- the two tiles' chains interleaved
  // (a dependent instruction waits for its predecessor), smallest terms first within an accumulator.
  // (Measured and not shipped, gram launch of config 3 on one box: the fragments of half h + 1 read into a second register set under
  // this half's instructions 410 us against 374; the same with reads and pieces pinned one behind each matrix instruction 432.  The
  // loop is not short of overlap: without its matrix instructions it takes 171 us, without its LDS-DMA 361 -- the matrix pipe on
  // random bf16 operands at the clock the part then holds (MI355X_MICROARCH.md, DVFS give-back (5): 1.5 - 1.7 GHz) IS the time, and
  // a denser instruction stream lowers that clock further.)
  auto half = [&](const char* slot) {
    Frags f;
    read_frag(slot, f, std::integral_constant<int, 0>{}); read_frag(slot, f, std::integral_constant<int, 1>{}); read_frag(slot, f, std::integral_constant<int, 2>{});
    read_frag(slot, f, std::integral_constant<int, 3>{}); read_frag(slot, f, std::integral_constant<int, 4>{}); read_frag(slot, f, std::integral_constant<int, 5>{});
    read_frag(slot, f, std::integral_constant<int, 6>{}); read_frag(slot, f, std::integral_constant<int, 7>{}); read_frag(slot, f, std::integral_constant<int, 8>{});
    issue();  // half h + 5 into the slot of half h - 1 (everybody left it at the barrier that ended half h - 1)
#if defined(BLR_PLANES_EXP) && BLR_PLANES_EXP == 1  /* timing experiment: no matrix instructions */
    acc[0][0] += __uint_as_float(f.v[0][0] ^ f.v[3][0] ^ f.v[8][3]);
    return;
#endif
    const gram_u4 &Ah = f.v[0], &Am = f.v[1], &Al = f.v[2], &B0h = f.v[3], &B0m = f.v[4], &B0l = f.v[5], &B1h = f.v[6], &B1m = f.v[7], &B1l = f.v[8];
    if (t0 && t1) {
      BLR_PM(accs[0], Al, B0h); BLR_PM(accs[1], Al, B1h);
      BLR_PM(acc[0], Ah, B0h);  BLR_PM(acc[1], Ah, B1h);
      BLR_PM(accs[0], Ah, B0l); BLR_PM(accs[1], Ah, B1l);
      BLR_PM(accs[0], Am, B0m); BLR_PM(accs[1], Am, B1m);
      BLR_PM(accs[0], Am, B0h); BLR_PM(accs[1], Am, B1h);
      BLR_PM(accs[0], Ah, B0m); BLR_PM(accs[1], Ah, B1m);
    } else if (t0) {
      BLR_PM(accs[0], Al, B0h); BLR_PM(acc[0], Ah, B0h); BLR_PM(accs[0], Ah, B0l);
      BLR_PM(accs[0], Am, B0m); BLR_PM(accs[0], Am, B0h); BLR_PM(accs[0], Ah, B0m);
    }
  };
  if (nh > 0) {
    for (int q = 0; q < kPlanesAhead; ++q) issue();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (kPlanesAhead - 1)) : "memory");  // half 0 has landed
    __syncthreads();
#pragma unroll 1
    for (int h = 0; h < nh; ++h) {
      half(smem + (h % kPlanesSlots) * kPlanesHalf);
      // end of half h: half h + 1 must have landed (the four younger ones may stay in flight), then everybody's pieces are visible
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (kPlanesAhead - 1)) : "memory");
      __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the pieces issued beyond the last half)
  }
#undef BLR_PM
  // ---- epilogue: the wave's tiles into the split's partial tile, column-major (row = A-side row): a lane holds 4 consecutive rows
  float* out = a.Gpart + ((int64_t)sidx * a.ntiles + t) * (kPB * kPB);
  typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (!(k == 0 ? t0 : t1)) continue;  // (wave-uniform; the reduction never reads the strictly upper tiles of a diagonal macro tile)
    const int col = 32 * (2 * tc + k) + (lane & 31);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f4 v = {acc[k][4 * q] + accs[k][4 * q], acc[k][4 * q + 1] + accs[k][4 * q + 1], acc[k][4 * q + 2] + accs[k][4 * q + 2], acc[k][4 * q + 3] + accs[k][4 * q + 3]};
      v *= post_scale;
      *reinterpret_cast<f4*>(out + (int64_t)col * kPB + 32 * ti + 8 * q + 4 * (lane >> 5)) = v;
    }
  }
}

}  // namespace blr
