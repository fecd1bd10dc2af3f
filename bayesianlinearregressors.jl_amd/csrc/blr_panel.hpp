// Panel step of the large-D blocked Cholesky (D > 128): factor the 128 x 128 diagonal block L_pp AND apply X <- X L_pp^-T to the
// rows below it, one launch per panel (reference: `cholesky(Symmetric(...))` in bayesian_linear_regression.jl:86 and the
// triangular solves around it; this is the latency chain of configs 3 and 5).
//
// Round-3 form, WAVE-SPECIALISED.  The round-2 kernel ran every wave through the same three sections per 16 columns
// (row-per-lane elimination of the whole panel, write-back, trailing MFMAs: 8.2 k cycles, all of it serial).  Here
//   * wave 0 (the CHAIN wave) does the only inherently serial work.  Per 16 columns it factors ONE 16 x 16 diagonal tile, one
//     row per lane, with every broadcast a DPP `row_newbcast` operand of the multiply-add itself (no v_readlane / SGPR round
//     trip), forms the tile's row-scaled INVERSE in the same pass (each 16-lane DPP row carries four columns of it, i.e.
//     exactly the four MFMA B-fragments), then solves the ONE sub-diagonal tile the next diagonal tile needs and applies it
//     -- two groups of 4 MFMAs -- and goes on to the next tile.  It waits for nobody's trailing update.
//   * waves 1.. (UPDATE waves) own every other tile -- the 28 off-diagonal tiles of the block, its diagonal tiles 2..7 until
//     the step before they are due (then they are handed to the chain wave through LDS), and the 8 x ER/16 tiles of this
//     workgroup's ER rows of X -- in MFMA accumulators, dealt round-robin in column order so that every step's active set is
//     balanced.  The triangular solve of a column of tiles is 4 MFMAs per tile against the inverted diagonal tile, the trailing
//     update 4 MFMAs per tile from a column-major LDS image of the finished column (one image = A- and B-fragments alike).
//   * accumulators hold the NEGATED matrix, so the trailing update is a plain multiply-accumulate with no sign flips.
// Two workgroup barriers per 16 columns.  Every workgroup factors L_pp redundantly (the chain is that factorisation whoever
// runs it) and takes ER rows of X along, which go back to memory as they are solved; workgroup 0 writes L_pp back -- a column's
// tiles one step after they were solved, the diagonal tiles at the end -- once an arrival counter says every workgroup has
// read A_pp.  The counter lives on the device and workgroup 0 zeroes it on its way out: the host keeps no state for it.
// Measured in isolation by tools/panel_bench.hip (against a host factorisation; -DBLR_STAMPS: section sums and a time line).
#pragma once
#include "blr_common.hpp"

#ifndef BLR_PANEL_WAVES
#define BLR_PANEL_WAVES 8  // 1 chain wave + 7 update waves (tools/panel_bench: 4 waves 41 us, 8 waves 20 us per f32 panel)
#endif

#ifndef BLR_PANEL_ER
#define BLR_PANEL_ER 16    // rows of X per workgroup where the chip has a CU for every workgroup (tools/panel_bench, 1920 rows
#endif                     // below the block: 64 rows x 30 workgroups 20.0 us, 32 x 60 18.6, 16 x 120 17.0; f64 47 / 42 / 37)

namespace blr {

// counters per bank of panel_chain_kernel's arrival words (one per factorisation of a grouped launch; blr_abi.hip kChainBatchMax)
constexpr int kPanelArriveWords = 128;

template <typename T, int NW_, int ER_ = BLR_PANEL_ER, int NBT_ = 8>
struct ChainCfg {
  static constexpr int NW = NW_;             // waves per workgroup: wave 0 = chain wave
  static constexpr int NU = NW_ - 1;         // update waves
  static constexpr int NBT = NBT_;           // 16 x 16 tiles along the edge of the diagonal block
  static constexpr int W = 16 * NBT_;        // columns of the panel
  static constexpr int ER = ER_;             // rows of X per workgroup (16, 32 or 64)
  static_assert(ER_ == 16 || ER_ == 32 || ER_ == 64, "whole 16-row tiles, callers pad the rows below the block to 64");
  static_assert(NBT_ == 8, "panel width: 128 columns (a 256-column instance measured 45 us against 2 x 15: its update waves fall behind)");
  static constexpr int XT = ER / 16;         // ... as 16-row tiles
  static constexpr int NB1 = NBT - 1;        // row ids 0 .. NB1-1: block row tiles 1 .. NBT-1; from NB1 on: the X row tiles
  static constexpr int NR = NB1 + XT;        // row ids of a panel image
  static_assert(NB1 + ER_ / 16 <= 3 * (NW_ - 1), "the solve handles at most three tiles of a column per update wave");
  // update-wave tiles in column order: column K holds the row ids R0(K) .. NR-1, R0 = K for K < 2 and K - 1 (the diagonal
  // tile, row tile K = id K - 1) from K = 2 on
  static constexpr int NT = NBT * NR - NBT * (NBT - 1) / 2 + (NBT - 2);
  static constexpr int SLOTS = (NT + NU - 1) / NU;
  static constexpr int NPAIR = (SLOTS + 1) / 2;
  static constexpr int IMG = NR * 256;       // elements of one panel image (tiles column-major 16 x 16)
  static constexpr int LDD = 20;             // row stride of the row-major 16 x 16 images (rows 16-byte aligned in f32 and f64)
  static constexpr int S = (int)sizeof(T);
  static constexpr int OFF_IMG = 0;                          // [2][NR][256]  solved columns (fragment images)
  static constexpr int OFF_PRE = OFF_IMG + 2 * IMG * S;      // [NR][256]     the column about to be solved (negated)
  static constexpr int OFF_LINV = OFF_PRE + IMG * S;         // [2][256]      row-scaled inverse of the diagonal tile (B-fragments)
  static constexpr int OFF_DIN = OFF_LINV + 2 * 256 * S;     // [2][256]      diagonal tile handed to the chain wave
  static constexpr int OFF_CSCR = OFF_DIN + 2 * 256 * S;     // [256]         chain wave: its own solved sub-diagonal tile
  static constexpr int OFF_ZERO = OFF_CSCR + 256 * S;        // [256]         zeros (A operand of a slot that sits out)
  static constexpr int OFF_DSCR = OFF_ZERO + 256 * S;        // [16][LDD]     chain wave: accumulator -> one row per lane
  static constexpr int OFF_LDIAG = OFF_DSCR + 16 * LDD * S;  // [NBT][16][LDD] rows of the factored diagonal tiles
  static constexpr int OFF_INFO = OFF_LDIAG + NBT * 16 * LDD * S;
  static constexpr int OFF_TAB = OFF_INFO + 16;              // [NU][NBT] first solved row id of wave u in column J
  static constexpr int LDS_BYTES = OFF_TAB + NU * NBT * 4;
  static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU must fit the 160 KB of LDS");
  __host__ __device__ static constexpr int col_begin(int K) {  // first tile of column K in the enumeration
    return K == 0 ? 0 : (K == 1 ? NR : 2 * NR - 1 + (K - 2) * (NR + 1) - ((K - 1) * K / 2 - 1));
  }
};

#ifdef BLR_STAMPS
__device__ unsigned long long g_stamps2[8];
__device__ unsigned long long g_tl[8][80];   // raw time line, workgroup 0: [wave][event]
#define BLR_TL(ev) do { if (blockIdx.x == 0 && lane == 0) g_tl[wave][ev] = __builtin_amdgcn_s_memtime(); } while (0)  // tools/panel_bench: section sums of update wave 0 (the chain wave uses g_stamps)
#define BLR_USTAMP(slot) do { if (u == 0) { unsigned long long t__ = __builtin_amdgcn_s_memtime(); ust[slot] += t__ - uprev; uprev = t__; } } while (0)
#else
#define BLR_USTAMP(slot) do {} while (0)
#define BLR_TL(ev) do {} while (0)
#endif

__device__ __forceinline__ int chain_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ---- DPP row_newbcast: lane K of the own 16-lane row as the first source of the instruction itself -------------------
// hipcc does not fold a v_mov_dpp into the consuming multiply-add here (it selects v_fma with a negated source), so the
// instructions are written out.  The hardware wants two wait states between a VALU write of a VGPR and a DPP read of it and
// the compiler cannot see through the asm to insert them: every DPP read either carries its own s_nop or sits behind
// instructions that are known to separate it from the write.
template <int K>
__device__ __forceinline__ void fmac_bc16(float& acc, float b, float m) {  // acc += (lane K's b) * m
  asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(m), "n"(K));
}
template <int K>
__device__ __forceinline__ void fmac_bc16(double& acc, double b, double m) {
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(m), "n"(K));
}
template <int K>
__device__ __forceinline__ void fmac_bc16_gap(float& acc, float b, float m) {
  asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(m), "n"(K));
}
template <int K>
__device__ __forceinline__ void fmac_bc16_gap(double& acc, double b, double m) {
  asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(m), "n"(K));
}
template <int K>
__device__ __forceinline__ float mov_bc16_gap(float b) {
  float d;
  asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(b), "n"(K));
  return d;
}
template <int K>
__device__ __forceinline__ double mov_bc16_gap(double b) {
  int lo = __double2loint(b), hi = __double2hiint(b), dlo, dhi;
  asm("s_nop 1\n\tv_mov_b32_dpp %0, %2 row_newbcast:%4 row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b32_dpp %1, %3 row_newbcast:%4 row_mask:0xf bank_mask:0xf"
      : "=&v"(dlo), "=&v"(dhi) : "v"(lo), "v"(hi), "n"(K));
  return __hiloint2double(dhi, dlo);
}
// two wait states, tied to the value that is about to be read through DPP: its producer stays in front, its DPP readers behind
template <typename T>
__device__ __forceinline__ void dpp_gap(T& x) { asm("s_nop 1" : "+v"(x)); }

template <int K, typename T>
__device__ __forceinline__ void fmac_bc16_k(T& acc, T b, T m, int k) {
  if (k == K) fmac_bc16<K>(acc, b, m);
}

// One 16 x 16 tile: Cholesky factor and (row-scaled) inverse, one row per lane, every 16-lane DPP row on its own copy.
//   a[c]  in: A(r, c) (entries right of the diagonal are dead values), out: L(r, c);       r = lane & 15
//   y[j]  out: L(r, r) Linv(r, 4 j + q)                                                     q = lane >> 4
// Column C: l = a[C] rsqrt(pivot); a[k] -= l l_k for k > C; y~ -= (l / L_CC) y~_C for the rows below C.  A non-positive pivot
// turns the rest of the diagonal into NaN (checked by the caller).
// Generic form (f64): the order of the stream is the compiler's.
template <typename T, int C>
__device__ __forceinline__ void tile_factor_col(T (&a)[16], T (&y)[4], T l, T ln, T rs, int r) {
  // on entry: l = L(r, C) (already in a[C] and past its wait states), ln = -l, rs = 1 / L(C, C)
  if constexpr (C < 16) {
    T l1 = T(0), ln1 = T(0), rs1 = T(0);
    if constexpr (C < 15) {
      fmac_bc16<(C + 1) & 15>(a[(C + 1) & 15], l, ln);
      const T d2n = mov_bc16_gap<(C + 1) & 15>(a[(C + 1) & 15]);
      rs1 = fast_rsqrt(d2n);
      l1 = a[(C + 1) & 15] * rs1;
      ln1 = -l1;
      dpp_gap(l1);
      a[(C + 1) & 15] = l1;
    }
#pragma unroll
    for (int k = C + 2; k < 16; ++k) {
      fmac_bc16_k<2>(a[k], l, ln, k);   fmac_bc16_k<3>(a[k], l, ln, k);   fmac_bc16_k<4>(a[k], l, ln, k);
      fmac_bc16_k<5>(a[k], l, ln, k);   fmac_bc16_k<6>(a[k], l, ln, k);   fmac_bc16_k<7>(a[k], l, ln, k);
      fmac_bc16_k<8>(a[k], l, ln, k);   fmac_bc16_k<9>(a[k], l, ln, k);   fmac_bc16_k<10>(a[k], l, ln, k);
      fmac_bc16_k<11>(a[k], l, ln, k);  fmac_bc16_k<12>(a[k], l, ln, k);  fmac_bc16_k<13>(a[k], l, ln, k);
      fmac_bc16_k<14>(a[k], l, ln, k);  fmac_bc16_k<15>(a[k], l, ln, k);
    }
    const T t = (r > C) ? ln * rs : T(0);  // -L(r, C) / L(C, C) for the rows below the pivot row
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (4 * j <= C) fmac_bc16_gap<C>(y[j], y[j], t);  // columns 4 j + q <= C only: the others of row C are still zero
    tile_factor_col<T, C + 1>(a, y, l1, ln1, rs1, r);
  }
}

// f32: the stream in OUR order.  The dependent chain of a column is
//   fmac a[C+1] -> (2 wait states) -> v_rsq_dpp of the pivot -> l = a[C+1] rs, -l -> (2 wait states) -> next column
// (a dependent VALU result costs ~8-10 cycles here, not the 4 of the issue cadence); everything else -- the other 14 - C
// multiply-adds, the inverse's multiply-adds, the row mask -- is filler.  hipcc keeps such a stream in source order and puts
// the whole chain BEHIND the fillers (3.0 k cycles per tile); dealt into the chain's latency slots by hand, as volatile
// single-instruction statements, the tile takes well under half of that.  NEWTON adds the refinement step of fast_rsqrt to
// the chain (three more dependent instructions per column); without it the pivot's reciprocal root is the hardware's (1 ulp).
#define BLR_VA(...) asm volatile(__VA_ARGS__)
template <int K>
__device__ __forceinline__ void vfmac_bc16(float& acc, float b, float m) {
  BLR_VA("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(m), "n"(K));
}
// fillers: multiply-adds of column C for k = K0 .. K0 + N - 1 (as far as they exist)
template <int C, int K0, int N>
__device__ __forceinline__ void col_fill(float (&a)[16], float ln) {
  if constexpr (N > 0 && K0 < 16) {
    vfmac_bc16<K0>(a[K0], a[C], ln);
    col_fill<C, K0 + 1, N - 1>(a, ln);
  }
}
template <int C, bool NEWTON>
__device__ __forceinline__ void tile_factor_col_f32(float (&a)[16], float (&y)[4], float ln, float rs, const uint64_t (&below)[16],
                                                    float half) {
  // on entry: a[C] = L(r, C) past its wait states, ln = -a[C], rs = 1 / L(C, C)
  float t;
  if constexpr (C < 15) {
    constexpr int n = C + 1;
    vfmac_bc16<n>(a[n], a[C], ln);                                                                  // chain
    BLR_VA("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(ln), "v"(rs));                                    // (wait state)
    BLR_VA("v_cndmask_b32 %0, 0, %0, %1" : "+v"(t) : "s"(below[C]));                                // rows below C only
    float rsn, lnn;
    BLR_VA("v_rsq_f32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(rsn) : "v"(a[n]), "n"(n));  // chain
    if constexpr (NEWTON) {
      float h;
      BLR_VA("v_mul_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(h) : "v"(a[n]), "v"(half), "n"(n));
      col_fill<C, C + 2, 1>(a, ln);
      BLR_VA("v_mul_f32 %0, %0, %1" : "+v"(h) : "v"(rsn));
      col_fill<C, C + 3, 1>(a, ln);
      BLR_VA("v_fma_f32 %0, -%0, %1, 0.5" : "+v"(h) : "v"(rsn));
      col_fill<C, C + 4, 1>(a, ln);
      BLR_VA("v_fmac_f32 %0, %1, %0" : "+v"(rsn) : "v"(h));
      col_fill<C, C + 5, 1>(a, ln);
    } else {
      col_fill<C, C + 2, 2>(a, ln);
      if constexpr (C + 2 >= 16) BLR_VA("s_nop 0");  // a transcendental result wants one wait state before its reader
    }
    BLR_VA("v_mul_f32 %0, -%1, %2" : "=v"(lnn) : "v"(a[n]), "v"(rsn));                              // chain: -l of column C+1
    BLR_VA("v_mul_f32 %0, %0, %1" : "+v"(a[n]) : "v"(rsn));                                         // chain:  l of column C+1
    // the inverse's multiply-adds and the remaining fillers (at least two: the wait states before a[n]'s DPP readers)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (4 * j <= C) vfmac_bc16<C>(y[j], y[j], t);
    col_fill<C, NEWTON ? C + 6 : C + 4, 16>(a, ln);
    tile_factor_col_f32<C + 1, NEWTON>(a, y, lnn, rsn, below, half);
  } else {
    BLR_VA("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(ln), "v"(rs));
    BLR_VA("v_cndmask_b32 %0, 0, %0, %1" : "+v"(t) : "s"(below[C]));
#pragma unroll
    for (int j = 0; j < 4; ++j) vfmac_bc16<C>(y[j], y[j], t);
  }
}

#ifndef BLR_PANEL_NEWTON
#define BLR_PANEL_NEWTON 0
#endif
// below[C]: lanes whose row (lane & 15) lies below row C -- wave-uniform masks, made once per kernel
__device__ __forceinline__ void tile_row_masks(uint64_t (&below)[16], int lane) {
#pragma unroll
  for (int c = 0; c < 16; ++c) below[c] = __ballot((lane & 15) > c);
}
template <typename T>
__device__ __forceinline__ void tile_factor_invert(T (&a)[16], T (&y)[4], const uint64_t (&below)[16], int lane) {
  const int r = lane & 15, q = lane >> 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) y[j] = (r == 4 * j + q) ? T(1) : T(0);
  if constexpr (sizeof(T) == 4) {
    float rs, ln, half = 0.5f;
    BLR_VA("s_nop 1\n\tv_rsq_f32_dpp %0, %1 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "=v"(rs) : "v"(a[0]));
    if constexpr (BLR_PANEL_NEWTON) {
      float h;
      BLR_VA("v_mul_f32_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "=v"(h) : "v"(a[0]), "v"(half));
      BLR_VA("v_mul_f32 %0, %0, %1" : "+v"(h) : "v"(rs));
      BLR_VA("v_fma_f32 %0, -%0, %1, 0.5" : "+v"(h) : "v"(rs));
      BLR_VA("v_fmac_f32 %0, %1, %0" : "+v"(rs) : "v"(h));
    } else {
      BLR_VA("s_nop 0");
    }
    BLR_VA("v_mul_f32 %0, -%1, %2" : "=v"(ln) : "v"(a[0]), "v"(rs));
    BLR_VA("v_mul_f32 %0, %0, %1\n\ts_nop 1" : "+v"(a[0]) : "v"(rs));
    tile_factor_col_f32<0, BLR_PANEL_NEWTON != 0>(a, y, ln, rs, below, half);
  } else {
    const T d2 = mov_bc16_gap<0>(a[0]);
    const T rs = fast_rsqrt(d2);
    T l = a[0] * rs;
    const T ln = -l;
    dpp_gap(l);
    a[0] = l;
    tile_factor_col<T, 0>(a, y, l, ln, rs, r);
  }
}

// ---- C-layout helpers (accumulator tile <-> memory) -----------------------------------------------------------------
// LDS image of a 16 x 16 tile, column-major: element (row, col) at 16 col + row, so MFMA fragment ks of lane l (row l & 15,
// column 4 ks + (l >> 4)) is image[64 ks + l] -- conflict-free dword reads.  An accumulator (f32: rows 4q..4q+3 of column c
// per lane) goes in as ONE 16-byte store per lane, but unswizzled the 8 lanes the hardware stores together hit 2 of the 8
// four-bank groups (4-way conflict, 32 LDS cycles per store, and all the update waves store all the time).  f32 images are
// therefore SWIZZLED: the 4-row chunk q of column c sits at chunk position q ^ ((c >> 1) & 3).  Stores are conflict-free,
// and a fragment read is still a permutation of 32 consecutive dwords per half-wave (the swizzle term is constant over
// lanes 0-31 and over 32-63 for a given ks): two per-lane offsets serve ks = {0, 2} and {1, 3}.
template <typename T>
__device__ __forceinline__ int img_store_off(int lane) {  // element offset of the lane's accumulator chunk
  const int c = lane & 15, q = lane >> 4;
  if constexpr (sizeof(T) == 4) return 16 * c + 4 * (q ^ ((c >> 1) & 3));
  else return 16 * c;
}
template <typename T>
struct FragOff {
  int a, b;  // element offsets: fragments 0 / 2 at a, a + 128; fragments 1 / 3 at b, b + 128
  __device__ __forceinline__ explicit FragOff(int lane) {
    if constexpr (sizeof(T) == 4) {
      const int rr = lane & 15, qq = lane >> 4, h = qq >> 1;
      a = 16 * qq + 4 * ((rr >> 2) ^ h) + (rr & 3);
      b = 64 + 16 * qq + 4 * ((rr >> 2) ^ ((2 + h) & 3)) + (rr & 3);
    } else {
      a = lane;
      b = 64 + lane;
    }
  }
};
template <typename T>
__device__ __forceinline__ void load_frags(const T* tile, const FragOff<T>& fo, T (&f)[4]) {
  f[0] = tile[fo.a];
  f[2] = tile[fo.a + 128];
  f[1] = tile[fo.b];
  f[3] = tile[fo.b + 128];
}
template <typename T>
__device__ __forceinline__ void tile_to_image(T* img, const typename Mfma<T>::acc4& v, int lane) {
  if constexpr (sizeof(T) == 4) {
    *reinterpret_cast<typename Mfma<T>::acc4*>(img + img_store_off<T>(lane)) = v;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) img[16 * (lane & 15) + Mfma<T>::crow(lane, e)] = v[e];
  }
}
template <typename T>
__device__ __forceinline__ typename Mfma<T>::acc4 tile_from_image(const T* img, int lane) {
  typename Mfma<T>::acc4 v;
  if constexpr (sizeof(T) == 4) {
    v = *reinterpret_cast<const typename Mfma<T>::acc4*>(img + img_store_off<T>(lane));
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = img[16 * (lane & 15) + Mfma<T>::crow(lane, e)];
  }
  return v;
}
// global column-major block (ld): accumulator tile at (row0, col0)
template <typename T>
__device__ __forceinline__ typename Mfma<T>::acc4 tile_from_global(const T* g, int64_t ld, int row0, int col0, int lane) {
  typename Mfma<T>::acc4 v;
  const T* p = g + (int64_t)(col0 + (lane & 15)) * ld + row0;
  if constexpr (sizeof(T) == 4) {
    v = *reinterpret_cast<const typename Mfma<T>::acc4*>(p + 4 * (lane >> 4));
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = p[Mfma<T>::crow(lane, e)];
  }
  return v;
}
// ... the same for a SYMMETRIC tile of which only the lower triangle is stored
template <typename T>
__device__ __forceinline__ typename Mfma<T>::acc4 diag_tile_from_global(const T* g, int64_t ld, int d0, int lane) {
  typename Mfma<T>::acc4 v;
  const int c = lane & 15;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int rr = Mfma<T>::crow(lane, e);
    v[e] = g[(int64_t)(d0 + min(rr, c)) * ld + d0 + max(rr, c)];
  }
  return v;
}
template <typename T>
__device__ __forceinline__ void tile_to_global(T* g, int64_t ld, int row0, int col0, const typename Mfma<T>::acc4& v, int lane) {
  T* p = g + (int64_t)(col0 + (lane & 15)) * ld + row0;
  if constexpr (sizeof(T) == 4) {
    *reinterpret_cast<typename Mfma<T>::acc4*>(p + 4 * (lane >> 4)) = v;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) p[Mfma<T>::crow(lane, e)] = v[e];
  }
}

// tile g of the column-ordered enumeration -> (column K, row id R)
template <typename C>
__device__ __forceinline__ void chain_tile(int g, int& K, int& R) {
  int k = 0, b = 0;
#pragma unroll
  for (int c = 1; c < C::NBT; ++c)
    if (g >= C::col_begin(c)) { k = c; b = C::col_begin(c); }
  K = k;
  R = (k < 2 ? k : k - 1) + (g - b);
}

// ---- trailing update of one update wave: slots in PAIRS from the back (the active slots are a suffix: columns ascend with
// the slot index), fragments of the next pair in flight while the current pair's 8 MFMAs issue.  A pair whose upper slot
// sits out ends the walk; a lower slot that sits out multiplies zeros.
template <typename T, typename C, int P>
__device__ __forceinline__ void trail_load(const T* img, const T* zero, const int (&sK)[C::SLOTS], const int (&sR)[C::SLOTS], int J,
                                           const FragOff<T>& fo, T (&f)[16]) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    constexpr int base = 2 * P;
    if (base + h < C::SLOTS) {
      const int s = (base + h < C::SLOTS) ? base + h : 0;
      const bool act = sK[s] > J;
      const T* ta = act ? img + sR[s] * 256 : zero;
      const T* tb = img + max(min(sK[s], C::NBT) - 1, 0) * 256;
      T fa[4], fb[4];
      load_frags<T>(ta, fo, fa);
      load_frags<T>(tb, fo, fb);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        f[8 * h + ks] = fa[ks];
        f[8 * h + 4 + ks] = fb[ks];
      }
    }
  }
}
template <typename T, typename C, int P>
__device__ __forceinline__ void trail_walk(const T* img, const T* zero, const int (&sK)[C::SLOTS], const int (&sR)[C::SLOTS], int J,
                                           const FragOff<T>& fo, typename Mfma<T>::acc4 (&acc)[C::SLOTS], T (&fc)[16], T (&fn)[16]) {
  // fc: fragments of pair P (loaded by the caller), fn: buffer for pair P - 1
  constexpr int hi = (2 * P + 1 < C::SLOTS) ? 2 * P + 1 : 2 * P;
  if (sK[hi] > J) {  // scalar branch
    if constexpr (P > 0) trail_load<T, C, P - 1>(img, zero, sK, sR, J, fo, fn);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        constexpr int base = 2 * P;
        if (base + h < C::SLOTS) {
          const int s = (base + h < C::SLOTS) ? base + h : 0;
          acc[s] = Mfma<T>::mma(fc[8 * h + ks], fc[8 * h + 4 + ks], acc[s]);
        }
      }
    if constexpr (P > 0) trail_walk<T, C, P - 1>(img, zero, sK, sR, J, fo, acc, fn, fc);
  }
}

// solve of one tile of column J (row id R): L = A Linv_J' from the negated pre-solve image and the row-scaled inverse
template <typename T>
__device__ __forceinline__ typename Mfma<T>::acc4 solve_tile(const T* pre_tile, const T (&fl)[4], T sc, const FragOff<T>& fo) {
  using acc4 = typename Mfma<T>::acc4;
  T fa[4];
  load_frags<T>(pre_tile, fo, fa);
  acc4 z0 = {T(0), T(0), T(0), T(0)}, z1 = {T(0), T(0), T(0), T(0)};
  z0 = Mfma<T>::mma(fa[0], fl[0], z0);  // (-X) (D Linv)'
  z1 = Mfma<T>::mma(fa[1], fl[1], z1);
  z0 = Mfma<T>::mma(fa[2], fl[2], z0);
  z1 = Mfma<T>::mma(fa[3], fl[3], z1);
  acc4 z;
#pragma unroll
  for (int e = 0; e < 4; ++e) z[e] = (z0[e] + z1[e]) * sc;
  return z;
}

template <typename T, int NW, int ER = BLR_PANEL_ER, int NBT = 8>
__global__ __launch_bounds__(64 * NW, 1) void panel_chain_kernel(T* Abar, int64_t lda, int col0 /* first column of the panel */,
                                                                 int nrows_total, int32_t* info, unsigned* arrive,
                                                                 unsigned arrive_target, int64_t batch_stride = 0,
                                                                 int info_stride = 0, unsigned* arrive_next = nullptr) {
  using C = ChainCfg<T, NW, ER, NBT>;
  // blockIdx.y: one of several independent factorisations that step through their panels together (regressors of a batch
  // at D > 128: every launch of the chain is latency, not throughput, so G matrices cost little more than one)
  Abar += (int64_t)blockIdx.y * batch_stride;
  info += (int64_t)blockIdx.y * info_stride;
  arrive += blockIdx.y;
  using acc4 = typename Mfma<T>::acc4;
  static_assert(C::col_begin(NBT) == C::NT, "tile enumeration");
  constexpr int NB1 = C::NB1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* const IMG = reinterpret_cast<T*>(smem + C::OFF_IMG);
  T* const PRE = reinterpret_cast<T*>(smem + C::OFF_PRE);
  T* const LINV = reinterpret_cast<T*>(smem + C::OFF_LINV);
  T* const DIN = reinterpret_cast<T*>(smem + C::OFF_DIN);
  T* const CSCR = reinterpret_cast<T*>(smem + C::OFF_CSCR);
  T* const ZERO = reinterpret_cast<T*>(smem + C::OFF_ZERO);
  T* const DSCR = reinterpret_cast<T*>(smem + C::OFF_DSCR);
  T* const LDIAG = reinterpret_cast<T*>(smem + C::OFF_LDIAG);
  int* const INFO = reinterpret_cast<int*>(smem + C::OFF_INFO);
  int* const TAB = reinterpret_cast<int*>(smem + C::OFF_TAB);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = chain_uni(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const FragOff<T> fo(lane);

  T* const blk = Abar + (int64_t)col0 * lda + col0;      // the diagonal block
  const int r0 = col0 + C::W + blockIdx.x * C::ER;      // first row of this workgroup's slice of X
  const int nr = max(0, min(C::ER, nrows_total - r0));  // 0 (nothing below the block) or ER: callers pad to 64 rows
  T* const Xg = Abar + (int64_t)col0 * lda + r0;        // X(row, col) at Xg[col * lda + row]
  // `arrive` counts the workgroups that have read A_pp, 0 .. arrive_target = gridDim.x within a launch.  Nobody re-arms it:
  // consecutive launches of a stream alternate between two banks of kPanelArriveWords counters, and every launch clears the
  // bank of its SUCCESSOR (`arrive_next`; all 128 words, whatever the group sizes) -- that bank's last user has completed, with
  // every increment it was ever going to make, before this launch started.  A wait that times out (bounded spins: a logic
  // error upstream must not hang the GPU) therefore cannot leave a pre-advanced counter behind; it reports status -999 and
  // the block is not written back.
  if (arrive_next != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < kPanelArriveWords)
    __hip_atomic_store(arrive_next + threadIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (*info != 0) return;  // an earlier panel already failed (uniform over the launch): nothing to do, nobody counts or waits
  if (tid == 0) INFO[0] = 0;
  if (tid < 256) ZERO[tid] = T(0);
  unsigned arrived = arrive_target - 1u;  // workgroup 0, thread 0: the arrival counter as read a few steps before the end

  if (wave == 0) {
    __builtin_amdgcn_s_setprio(3);
    BLR_STAMP_INIT;
    // ================================================= chain wave =================================================
    // diagonal tiles 0 and 1 come straight from memory (tile 1 only ever misses panel 0, which this wave applies itself);
    // entries above the diagonal are whatever memory holds -- never used
    acc4 d = tile_from_global<T>(blk, lda, 0, 0, lane);
    acc4 d1 = tile_from_global<T>(blk, lda, 16, 16, lane);
#pragma unroll
    for (int e = 0; e < 4; ++e) { d[e] = -d[e]; d1[e] = -d1[e]; }
    uint64_t below[16];
    tile_row_masks(below, lane);
    BLR_STAMP(0);
    BLR_TL(0);
#pragma unroll 1
    for (int J = 0; J < NBT; ++J) {
      const int par = J & 1;
      // tile J (negated; only its lower triangle means anything) -> one row per lane
#pragma unroll
      for (int e = 0; e < 4; ++e) DSCR[C::LDD * Mfma<T>::crow(lane, e) + r] = -d[e];
      T a[16], y[4];
      typedef T vecT __attribute__((ext_vector_type(Mfma<T>::VEC)));
      constexpr int V = Mfma<T>::VEC;
#pragma unroll
      for (int u = 0; u < 16 / V; ++u) {
        const vecT t = *reinterpret_cast<const vecT*>(DSCR + C::LDD * r + V * u);
#pragma unroll
        for (int e = 0; e < V; ++e) a[V * u + e] = t[e];
      }
      BLR_STAMP(1);
      tile_factor_invert<T>(a, y, below, lane);
      BLR_STAMP(2);
      BLR_TL(1 + 4 * J);
      // publish: the row-scaled inverse as B-fragments and the factor's rows (write-back, diagonal for the solves)
#pragma unroll
      for (int j = 0; j < 4; ++j) LINV[par * 256 + 64 * j + lane] = y[j];
      if (q == 0) {
#pragma unroll
        for (int u = 0; u < 16 / V; ++u) {
          vecT t;
#pragma unroll
          for (int e = 0; e < V; ++e) t[e] = a[V * u + e];
          *reinterpret_cast<vecT*>(LDIAG + (J * 16 + r) * C::LDD + V * u) = t;
        }
      }
      const T dgv = LDIAG[(J * 16 + r) * C::LDD + r];  // (own write: no barrier needed)
      const T sc = -fast_rcp(dgv);
      T ys[4];  // -Linv(r, 4 ks + q): the inverse with its row scaling divided out and the sign of the negated images folded in
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) ys[ks] = y[ks] * sc;
      __syncthreads();  // B1: the inverse of tile J and the pre-solve images of column J are there
      // A_pp and X have been read by now (the update waves waited for their loads before B1 of step 0): arrive.
      if (J == 0 && tid == 0) __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // (the answer is not needed before the write-back: the load's latency disappears behind the last steps)
      if (J == NBT - 3 && tid == 0 && blockIdx.x == 0) arrived = __hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      BLR_STAMP(3);
      BLR_TL(2 + 4 * J);
      if (J < NBT - 1) {
        // the one sub-diagonal tile that the next diagonal tile waits for, L(J+1, J) = A(J+1, J) Linv_J', then tile J+1 -= L L'
        acc4 dn;
        if (J == 0) dn = d1;
        else dn = tile_from_image<T>(DIN + ((J + 1) & 1) * 256, lane);
        T fa[4], f[4];
        load_frags<T>(PRE + J * 256, fo, fa);
        if constexpr (sizeof(T) == 4) {
          // computed TRANSPOSED (Linv_J A'): the accumulator then holds L(c, 4 q + e), and swapping the roles of the 16-lane
          // row index q and the register index e -- four v_permlane swaps -- makes register e fragment e.  No LDS round trip.
          acc4 z0 = {T(0), T(0), T(0), T(0)}, z1 = {T(0), T(0), T(0), T(0)};
          z0 = Mfma<T>::mma(ys[0], fa[0], z0);
          z1 = Mfma<T>::mma(ys[1], fa[1], z1);
          z0 = Mfma<T>::mma(ys[2], fa[2], z0);
          z1 = Mfma<T>::mma(ys[3], fa[3], z1);
          unsigned w[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) w[e] = __float_as_uint(z0[e] + z1[e]);
          auto p01 = __builtin_amdgcn_permlane16_swap(w[0], w[1], false, false);
          auto p23 = __builtin_amdgcn_permlane16_swap(w[2], w[3], false, false);
          auto p02 = __builtin_amdgcn_permlane32_swap(p01[0], p23[0], false, false);
          auto p13 = __builtin_amdgcn_permlane32_swap(p01[1], p23[1], false, false);
          f[0] = __uint_as_float(p02[0]);
          f[2] = __uint_as_float(p02[1]);
          f[1] = __uint_as_float(p13[0]);
          f[3] = __uint_as_float(p13[1]);
        } else {
          const acc4 z = solve_tile<T>(PRE + J * 256, y, sc, fo);
          tile_to_image<T>(CSCR, z, lane);
          load_frags<T>(CSCR, fo, f);
        }
        acc4 w2 = {T(0), T(0), T(0), T(0)};
        dn = Mfma<T>::mma(f[0], f[0], dn);
        w2 = Mfma<T>::mma(f[1], f[1], w2);
        dn = Mfma<T>::mma(f[2], f[2], dn);
        w2 = Mfma<T>::mma(f[3], f[3], w2);
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = dn[e] + w2[e];
      }
      BLR_STAMP(4);
      BLR_TL(3 + 4 * J);
      __syncthreads();  // B2: column J of L is there (for the update waves)
      BLR_STAMP(5);
      BLR_TL(4 + 4 * J);
    }
    BLR_STAMP_FLUSH;
  } else {
    // ================================================= update waves =================================================
    const int u = wave - 1;
    acc4 acc[C::SLOTS];
    int sK[C::SLOTS], sR[C::SLOTS], tK[C::SLOTS], tO[C::SLOTS];
    {
      // Lane s works out slot s (tile s NU + u of the column-ordered enumeration) once, in parallel; the unrolled loop below
      // only moves the results into scalars.  One load path for every kind of tile: diagonal tiles come in whole like the
      // others -- what memory holds above their diagonal is never used (the trailing update and the hand-off to the chain wave
      // are element-wise there, and the chain wave reads rows up to their diagonal only).
      const int g = lane * C::NU + u;
      int Kv = 9, Rv = 0;
      if (g < C::NT) chain_tile<C>(g, Kv, Rv);
      const bool live = Kv < 9;
      const bool isdiag = live && Kv >= 2 && Rv == Kv - 1;
      const int tKv = !live ? 100 : (isdiag ? Kv - 2 : Kv - 1);
      const int tOv = isdiag ? C::OFF_DIN / C::S + ((Kv - 2) & 1) * 256 : C::OFF_PRE / C::S + Rv * 256;
      // element offset of the tile from blk: rows of the block, or this workgroup's rows of X (r0 - p 128 further down)
      const int rowv = Rv < NB1 ? 16 * (Rv + 1) : (nr > 0 ? (r0 - col0) + 16 * (Rv - NB1) : 0);
      const int offv = (live ? 16 * Kv : 0) * (int)lda + rowv;
      const int lane_off = r * (int)lda + (sizeof(T) == 4 ? 4 * q : 0);
#pragma unroll
      for (int s = 0; s < C::SLOTS; ++s) {
        sK[s] = __builtin_amdgcn_readlane(Kv, s);
        sR[s] = __builtin_amdgcn_readlane(Rv, s);
        tK[s] = __builtin_amdgcn_readlane(tKv, s);
        tO[s] = __builtin_amdgcn_readlane(tOv, s);
        const T* base = blk + __builtin_amdgcn_readlane(offv, s);
        if constexpr (sizeof(T) == 4) {
          acc[s] = *reinterpret_cast<const acc4*>(base + lane_off);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[s][e] = base[lane_off + Mfma<T>::crow(lane, e)];
        }
      }
      if (nr == 0) {  // no rows below the block: those tiles are zero
#pragma unroll
        for (int s = 0; s < C::SLOTS; ++s)
          if (sR[s] >= NB1) acc[s] = acc4{T(0), T(0), T(0), T(0)};
      }
    }
    // this wave's tiles of column J that are solved (row ids >= J): the first one, TAB[u][J]; the others are NU, 2 NU further on
    if (lane < NBT) {
      const int J = lane;
      const int R0 = J < 2 ? J : J - 1;
      int cb = 0;
#pragma unroll
      for (int c = 1; c < NBT; ++c)
        if (J == c) cb = C::col_begin(c);
      const int g0 = cb + (J - R0);                        // enumeration index of the tile with row id J
      const int x = ((u - g0) % C::NU + C::NU) % C::NU;   // first tile at or after it that is dealt to wave u
      TAB[u * NBT + J] = J + x;
    }
#pragma unroll
    for (int s = 0; s < C::SLOTS; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[s][e] = -acc[s][e];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef BLR_STAMPS
    unsigned long long ust[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long uprev = __builtin_amdgcn_s_memtime();
#endif
    BLR_TL(0);
    T* const S0 = reinterpret_cast<T*>(smem);
    // publish pass for "step -1": the pre-solve images of column 0
#pragma unroll
    for (int s = 0; s < C::SLOTS; ++s)
      if (tK[s] == -1) tile_to_image<T>(S0 + tO[s], acc[s], lane);
    // Workgroup 0 writes the block's solved tiles back over A_pp one step after they were solved (their image lives two steps) --
    // not before every workgroup has read A_pp: the arrival counter is read behind the barrier of step 1 and needed a
    // trailing update later
    unsigned arrived_u = arrive_target;
    bool timed_out = false;  // (wave-uniform)
#pragma unroll 1
    for (int J = 0; J < NBT; ++J) {
      const int par = J & 1;
      T* const img = IMG + par * C::IMG;
      BLR_USTAMP(0);
      const int R1 = chain_uni(TAB[u * NBT + J]);
      BLR_USTAMP(1);
      BLR_TL(1 + 4 * J);
      __syncthreads();  // B1
      BLR_USTAMP(2);
      BLR_TL(2 + 4 * J);
      if (J == 1 && blockIdx.x == 0 && lane == 0) arrived_u = __hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      {
        T fl[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fl[ks] = LINV[par * 256 + 64 * ks + lane];
        // the inverse comes scaled by L(n, n) per row n = output column: divide it out here, off the chain wave's path
        const T dgv = LDIAG[(J * 16 + r) * C::LDD + r];
        if (u == 0) {  // pivot check: L(n, n) is NaN from the first non-positive pivot on
          const unsigned long long okm = __ballot(dgv > T(0)) & 0xFFFFull;
          if (okm != 0xFFFFull && lane == 0 && INFO[0] == 0) INFO[0] = 16 * J + __builtin_ctzll(~okm) + 1;
        }
        const T sc = -fast_rcp(dgv);
        // solved tiles go to the column image (everybody's trailing update; the block's own tiles are written back from it a
        // step later) and, rows of X, straight to memory: nobody else reads them
        auto put = [&](int R, const acc4& z) {
          tile_to_image<T>(img + R * 256, z, lane);
          if (R >= NB1 && nr > 0) tile_to_global<T>(Xg, lda, 16 * (R - NB1), 16 * J, z, lane);
        };
        const int R2 = R1 + C::NU, R3 = R1 + 2 * C::NU;
        if (R3 < C::NR) {  // three tiles
          const acc4 za = solve_tile<T>(PRE + R1 * 256, fl, sc, fo);
          const acc4 zb = solve_tile<T>(PRE + R2 * 256, fl, sc, fo);
          const acc4 zc = solve_tile<T>(PRE + R3 * 256, fl, sc, fo);
          put(R1, za);
          put(R2, zb);
          put(R3, zc);
        } else if (R2 < C::NR) {  // two
          const acc4 za = solve_tile<T>(PRE + R1 * 256, fl, sc, fo);
          const acc4 zb = solve_tile<T>(PRE + R2 * 256, fl, sc, fo);
          put(R1, za);
          put(R2, zb);
        } else if (R1 < C::NR) {
          const acc4 za = solve_tile<T>(PRE + R1 * 256, fl, sc, fo);
          put(R1, za);
        }
      }
      BLR_USTAMP(3);
      BLR_TL(3 + 4 * J);
      __syncthreads();  // B2
      BLR_USTAMP(4);
      BLR_TL(4 + 4 * J);
      // trailing update: -C(R, K) += L(R, J) L(K, J)'  (diagonal tiles: R = K - 1, both operands the same image tile)
      {
        T f0[16], f1[16];
        trail_load<T, C, C::NPAIR - 1>(img, ZERO, sK, sR, J, fo, f0);
        trail_walk<T, C, C::NPAIR - 1>(img, ZERO, sK, sR, J, fo, acc, f0, f1);
      }
      BLR_USTAMP(5);
      // workgroup 0: column J - 1 of the block (row tiles J .. NBT-1, ids J-1 .. NB1-1) from its image to memory
      if (blockIdx.x == 0 && J >= 1) {
        if (J == 1) {
          unsigned a = (unsigned)__builtin_amdgcn_readfirstlane((int)arrived_u);
          long long spins = 0;
          while ((int)(a - arrive_target) < 0) {  // somebody has not read A_pp yet (a workgroup that started late)
            __builtin_amdgcn_s_sleep(8);
            a = __hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (++spins > 20000000LL) {  // a logic error upstream must not hang the GPU -- and must not pass for a result
              timed_out = true;
              if (lane == 0) __hip_atomic_store(info, -999, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              break;
            }
          }
        }
        const T* const pimg = IMG + (par ^ 1) * C::IMG;
        for (int R = J - 1 + u; R < NB1 && !timed_out; R += C::NU) {
          const acc4 v = tile_from_image<T>(pimg + R * 256, lane);
          tile_to_global<T>(blk, lda, 16 * (R + 1), 16 * (J - 1), v, lane);
        }
      }
      // publish pass: what is due after this step's trailing update (pre-solve images of column J + 1, diagonal tile J + 2)
#pragma unroll
      for (int s = 0; s < C::SLOTS; ++s)
        if (tK[s] == J) tile_to_image<T>(S0 + tO[s], acc[s], lane);
    }
#ifdef BLR_STAMPS
    if (u == 0 && lane == 0 && blockIdx.x == 0)
      for (int i = 0; i < 8; ++i) g_stamps2[i] += ust[i];
#endif
  }

  // ---- the diagonal tiles (workgroup 0; its off-diagonal tiles went out a step after they were solved, the rows of X at once).
  // A failed factorisation leaves NaNs behind its first bad pivot; the status says so.
  BLR_TL(72);
  const int bad = INFO[0];
  if (blockIdx.x != 0) return;
  if (tid == 0) {
    int late = 0;
    if ((int)(arrived - arrive_target) < 0) {
      long long spins = 0;
      while ((int)(__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - arrive_target) < 0) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > 20000000LL) { late = 1; break; }  // (see the head of the kernel)
      }
    }
    if (late) __hip_atomic_store(info, -999, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (bad != 0) *info = col0 + bad;
    INFO[1] = late;
  }
  if (bad != 0) return;
  __syncthreads();  // B3
  if (INFO[1] != 0) return;  // somebody may still be reading A_pp: leave it alone
  // lower triangles, from the rows the chain wave left in LDIAG, one tile per wave
  for (int K = wave; K < NBT; K += NW) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int rho = Mfma<T>::crow(lane, e);
      if (r <= rho) blk[(int64_t)(16 * K + r) * lda + 16 * K + rho] = LDIAG[(K * 16 + rho) * C::LDD + r];
    }
  }
  BLR_TL(73);
}

}  // namespace blr
