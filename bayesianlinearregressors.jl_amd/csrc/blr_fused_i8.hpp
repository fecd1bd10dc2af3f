// Fused per-regressor inference kernel for the headline shape (D = 128, fp64, 512 <= N <= 16384 + a partial block) with the Gram matrix
// on the INT8 matrix cores: an Ozaki-style exact splitting of the fp64 inputs.
//
// Why: the f64 matrix pipe (v_mfma_f64_16x16x4, 78.6 TF spec) bounds fused_small_kernel at ~0.75 M updates/s at the clock the part
// holds under that load (profiles/r04_microbench_ring_probe_sustained.txt); the same Gram costs a fifth of the matrix-pipe time on
// v_mfma_i32_32x32x32_i8 and the update is then bound by the HBM stream of X (4.2 MB per update) and by the slicing.
//
// Arithmetic (reference src/bayesian_linear_regression.jl:86, G = X X'):
//   * per row i of X a power of two 2^e_i, its CAPACITY: e_i = exponent of the row's largest entry in the first 96 columns + 2 (isotropic
//     noise; + 3 on x times the regressor's largest 1 / sqrt(s_n) under diagonal noise).  Q_in = round-to-nearest-even(x_in 2^(47 - e_i))
//     is a 48-bit signed integer using the WHOLE two's-complement range, obtained with ONE v_add_f64 (the magic-constant trick: the
//     integer sits in the low mantissa bits of x + 1.5 2^(e_i + 5));
//   * its six two's-complement bytes are the digits: Q = sum_s d_s 2^(8 (5 - s)), d_0 signed, d_1..d_5 unsigned, stored as
//     a_s = d_s - 128 (one XOR) so that every operand of the signed int8 MFMA fits;
//     sum_n Q_in Q_jn = sum_{s,t} 2^(8 (10 - s - t)) sum_n (a_is + o_s)(a_jt + o_t),  o_0 = 0, o_s = 128:
//       - the products sum_n a_is a_jt come out of the matrix cores EXACTLY (int32, |.| < 2^31 for N <= 16384), one accumulator per
//         digit group k = s + t, for the pairs with k < NG: NG = 6 (21 pairs) under isotropic noise, 7 (26 pairs) under diagonal noise;
//       - on the DIAGONAL tiles of the 6-group plan P(t, s) = P(s, t)': the pairs s < t once, mirrored at the hand-over, the pairs
//         s = t in accumulators of their own: 12 MFMAs instead of 21 (174 per 32-column k-step for the 10 tiles instead of 260);
//       - the offset terms are rank-one -- 128 (sum_s R_s(i) + sum_t R_t(j)) over the partners of group k, + 16384 N per pair of
//         offset digits, with the digit row sums R_s(i) = sum_n a_is from one v_dot4 per packed dword -- and are kept for ALL 36 pairs;
//       - of the DROPPED pairs (k >= NG) the mean part N abar_s(i) abar_t(j) = R_s(i) R_t(j) / N is kept too (five multiply-adds per
//         entry): inputs that came from float32, integers or powers of two have constant low digits, and their dropped products are
//         systematic, not noise (3e-13 of the diagonal scale with 6 groups); on the diagonal the pair (3, 3) is a sum of squares and comes
//         exactly from one more v_dot4; what is left of the dropped pairs is zero-mean: 3e-15 of the diagonal scale at N = 4096 (1 sigma);
//   * G_ij = 2^(e_i + e_j - 94) [sum_k 2^(80 - 8k) P_k(i,j) + offsets + mean parts], evaluated once per regressor in fp64.
//   * an entry beyond its row's capacity WRAPS in the 48-bit integer: the stream has then added c c' for that column, c = x - m 2^(e_i + 1).
//     The 32-column block is marked (one compare per k-step), read again at the hand-over and x x' - c c' is added in fp64, entry by
//     entry, in a fixed order (i8 repair).  N(0,1) rows: 0.3 marked blocks per regressor at N = 4096.  More than one marked block in 16
//     (heavy tails, a feature that wakes up late), Inf / NaN: the regressor goes to the fp64 kernel -- status kI8Retry, consumed by a
//     follow-up launch of fused_small_kernel -- so the fast path never returns a wrong number; blr_get_stat counts these.
// Error (measured against the fp64 kernel, tools/i8_gram.hip): A within 3e-14 of its diagonal scale (max over 32 x 128 x 128 entries), the
// evidence within 1e-15 .. 1e-14 on generic data and 1.5e-11 where delta'delta / s and |u|^2 cancel a thousandfold (seven groups: 8e-15
// of the diagonal scale; they cost 260 MFMAs per k-step, and 0.78 against 0.91 M updates/s).
//
// Structure: ONE 512-thread workgroup per regressor and CU (8 waves, two per SIMD, <= 256 registers each): the 68 accumulators of 16
// registers of the 6-group plan (36 + 4 x 8), at most 9 per wave.
//   * X streams HBM -> LDS by LDS-DMA, one 1 KiB piece per column, through a ring of three 32-column slots (two in flight); the pieces are
//     issued unconditionally (three k-steps beyond the end they load the last block again): a branch per piece split the k-step's schedule;
//   * every thread owns one row and 8 columns of a 32-column k-step: magic add, byte transposition with v_perm_b32, digit
//     planes to LDS in MFMA fragment order (double-buffered), b += x y in fp64 on the way;
//   * (tile, slot range) items are dealt to the 8 waves so that the two waves of a SIMD (w, w + 4) carry 42 - 44 MFMAs per k-step
//     between them (tables I8Items, from tools/i8_plan_search.py);
//   * inside a k-step the order is pinned by hand: MFMA, then the fragment reads two MFMAs ahead and one of 14 chunks of the slicing;
//     ONE workgroup barrier per k-step.  With 174 MFMAs the k-step is bound by the slicing path (2.9 k cycles with or without the
//     MFMAs; 2.4 k without the slicing), no longer by the matrix pipe or by power (in-kernel clock 2.0 GHz on N(0,1) data, 2.4 on zeros);
//   * hand-over: accumulators -> fp64 -> packed triangle of A in LDS by products only, in three phases that are a 3-colouring of the
//     (tile, wave) incidence; offsets, mean parts, prior in a pass of their own (shared noinline code: the hand-over's straight-line code
//     overflows the instruction cache once per regressor); repair of marked blocks; waves 4-7 exit; waves 0-3 run the blocked Cholesky
//     of fused_small_kernel (phase_chol) and a back substitution blocked by 16 (i8_backsolve_blocked);
//   * N need not be a multiple of 32: whole k-steps go through the stream, the last N % 32 columns are added to the finished matrix,
//     to b and to y'y in fp64 at the hand-over (a rank-r term, r < 32: ~1 % of an update);
//   * a prior mean mw != 0 (reference :57, :82: delta = y - X'mw) never touches the stream: G is exact, so
//     b = X delta / s = X y / s - (G / s) mw and delta'delta / s = y'y / s - 2 mw'X y / s + mw'(G / s) mw are formed from the
//     finished matrix (one 128 x 128 symmetric matrix-vector product in LDS; diag(G) / s is kept next to A = Lw + G / s); when these
//     differences cancel more than three digits (a prior mean that already explains the data) the fp64 kernel redoes the regressor;
//   * priors: diagonal (joins in the table pass), upper factor U (U'U added in fp64 after the prior-mean terms), dense (Lw added there
//     too; logdet Lw and the check of reference :78 from i8_prior_logdet_kernel, one blocked Cholesky per prior before the launch).
// (Measured alternatives, tools/i8_gram.hip: four waves with the whole register file each -- one wave per SIMD cannot keep the matrix
// pipe fed next to the slicing and the DMA issue: 4000 cycles per k-step against 3000; the factorisations as a second launch with two
// workgroups per CU: no gain; staggered starts, static priority for waves 4-7 (BLR_I8_SETPRIO), cache-warming touches, fragment reads
// three or four MFMAs ahead (BLR_I8_LEAD): no gain or a loss.)
#pragma once
#include "blr_fused_small.hpp"

namespace blr {

constexpr int kI8Threads = 512;
constexpr int kI8Retry = kI8RetryCode;  // info: "this regressor must be redone by the fp64 kernel" (never returned to callers)
constexpr int kI8MaxN = 16384;          // int32 accumulators: 6 N 2^14 < 2^31
constexpr int kI8MinN = 512;            // below, the fixed costs of the fast path buy nothing
constexpr int kI8Probe = 256;           // regressors of a large batch's first slice (one round of workgroups): its hand-back count steers the rest
constexpr int kI8ProbeMin = 4096;       // batches up to this size go in one slice (handle option I8_PROBE_MIN lowers it: tests)
constexpr int kI8MaxRepair = 32;        // 32-column blocks with entries beyond their row's capacity that are corrected in fp64 at the hand-over; more: fp64 kernel
// Digit groups kept (k = s + t < NG) and binades of capacity above the exponent of a row's largest entry in the first 96 columns:
//   isotropic noise: 6 groups, capacity 2^(E + 2) -- between 2 and 4 times that maximum, the whole two's-complement range of the
//     48-bit integer in use (the first versions kept a spare bit and three binades: 4 bits fewer per operand, 2^8 in the products --
//     what the seventh group bought).  N(0,1) rows: capacity 8 sigma for 99 % of the rows; the entries that do outgrow a row's
//     capacity (0.3 per regressor at N = 4096) wrap around in the integer and are corrected at the hand-over (i8 repair below);
//   diagonal noise: 7 groups, capacity 2^(E + 3) of x times the regressor's LARGEST 1 / sqrt(s_n) -- that bound is loose by the spread
//     of the variances (the entries sit (max w / typical w) below it, and the truncation error grows with the SQUARE of that: 25 x for
//     s_n = exp(N(0,1))), and the seventh group is what keeps the products accurate under it.
//   The handle option I8_GROUPS = 6 | 7 overrides the choice for both noise kinds: 7 under isotropic noise for callers that carry a
//     well-explained posterior forward and want the last digit (A within 1e-14 instead of 3e-14 of its diagonal scale, 260 instead of
//     174 MFMAs per k-step); 6 under diagonal noise where the variances are known to be of one magnitude (3e-14 x (max w / typical w)^2).
constexpr int kI8GroupsIso = 6, kI8GroupsDiag = 7;
template <int NG> struct I8Mode {
  static constexpr bool SYM = NG == 6;          // diagonal tiles: products s < t once (mirrored at the hand-over), s = t in accumulators of their own
  static constexpr int CAP = NG == 6 ? 2 : 3;   // capacity 2^(E + CAP)
};
// int8 MFMAs per 32-column k-step of the two plans (bench.py prices the int8 work of a launch with these)
constexpr int kI8MfmaPerKstep = 174;      // isotropic: 6 off-diagonal tiles x 21 + 4 diagonal tiles x (9 + 3)
constexpr int kI8MfmaPerKstepDiag = 260;  // diagonal noise: 10 tiles x 26
constexpr int kI8MfmaPerKstep7 = kI8MfmaPerKstepDiag;

// BLR_I8_STAMPS: diagnostic builds only (tools/i8_gram.hip): cycle sums of workgroup 0, one row per wave --
//   (sums over the workgroups with blockIdx % 257 == 0)  [0] the k-steps (MFMAs + slicing)  [2] DMA wait + barrier  [3] repair of marked blocks  [4] whole stream  [5] hand-over + conversion (+ tail columns, prior mean)
//   [6] factorisation  [7] back substitution + outputs
#ifdef BLR_I8_STAMPS
__device__ unsigned long long g_i8clk[4];
#define I8_STAMP_DECL unsigned long long i8t_prev = __builtin_amdgcn_s_memtime(), i8t_acc[4] = {0, 0, 0, 0}
#define I8_STAMP(slot) do { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); i8t_acc[slot] += t__ - i8t_prev; i8t_prev = t__; } while (0)
#define I8_STAMP_FLUSH(W) do { if ((blockIdx.x % 257) == 0 && (threadIdx.x & 63) == 0) for (int q__ = 0; q__ < 3; ++q__) atomicAdd(&g_i8stamps[W][q__], i8t_acc[q__]); } while (0)
#define I8_KSTAMP(slot) do { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); if ((blockIdx.x % 257) == 0 && (threadIdx.x & 63) == 0) atomicAdd(&g_i8stamps[threadIdx.x >> 6][slot], t__ - i8k_prev); i8k_prev = t__; } while (0)
#define I8_KSTAMP_DECL unsigned long long i8k_prev = __builtin_amdgcn_s_memtime(); const unsigned long long i8k_c0 = i8k_prev, i8k_r0 = __builtin_amdgcn_s_memrealtime()
// whole-workgroup shader cycles and 100 MHz ticks, summed over the workgroups with blockIdx % 256 == 0: the clock the part holds under this kernel
#define I8_CLKSTAMP do { if ((blockIdx.x & 255) == 0 && threadIdx.x == 0) { atomicAdd(&g_i8clk[0], __builtin_amdgcn_s_memtime() - i8k_c0); atomicAdd(&g_i8clk[1], __builtin_amdgcn_s_memrealtime() - i8k_r0); atomicAdd(&g_i8clk[2], 1ull); } } while (0)
#else
#define I8_CLKSTAMP do {} while (0)
#define I8_STAMP_DECL do {} while (0)
#define I8_STAMP(slot) do {} while (0)
#define I8_STAMP_FLUSH(W) do {} while (0)
#define I8_KSTAMP(slot) do {} while (0)
#define I8_KSTAMP_DECL do {} while (0)
#endif

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

struct I8Cfg {
  static constexpr int D = 128, S = 6, KC = 32, NSLOT = 3;
  static constexpr int SLOT_BYTES = KC * D * 8;                 // 32 KiB of raw fp64 columns
  static constexpr int RING_BYTES = NSLOT * SLOT_BYTES;         // 96 KiB
  static constexpr int DIG_BUF = S * 4 * 1024;                  // 24 KiB: [slice][row block of 32][1 KiB fragment]
  static constexpr int OFF_DIG = RING_BYTES;
  static constexpr int OFF_YB = OFF_DIG + 2 * DIG_BUF;          // y ring: NSLOT x 32 doubles
  static constexpr int OFF_RW = OFF_YB + NSLOT * KC * 8;        // 1 / sqrt(s_n) ring (diagonal noise): NSLOT x 32 doubles
  static constexpr int OFF_XCH = OFF_RW + NSLOT * KC * 8;       // exchange: 4 x 128 ints (row maxima; sums of squares of digit 3; y'y shares)
  static constexpr int OFF_FLAG = OFF_XCH + 4 * 128 * 4;        // 16 ints
  static constexpr int OFF_OFLAG = OFF_FLAG + 64;               // one byte per 32-column block: some entry outgrew its row's capacity (<= 512 blocks + 32)
  static constexpr int OFF_OMASK = OFF_OFLAG + 544;             // the same as 8 x 64-bit masks; then the repair's column masks 32 x 4 words, 4 words of scratch
  static constexpr int LDS_BYTES = OFF_OMASK + 64 + 512 + 32;
  // after the stream the ring is dead: P, bvec, ... of SmallCfg<double, 8> live there (phase_chol / phase_backsolve layout);
  // the conversion tables live in the digit area
  static constexpr int OFF_SC = OFF_DIG;                        // 2^(e_i - 47): 128 doubles
  static constexpr int OFF_BRED = OFF_SC + 128 * 8;             // b partials: 4 x 128 doubles
  static constexpr int OFF_GD = OFF_BRED + 4 * 128 * 8;         // diag(G) / sigma^2 WITHOUT the prior (prior-mean terms): 128 doubles
  static constexpr int OFF_TD = OFF_GD + 128 * 8;               // TD: sum_n a_3(i, n)^2 2^32, the diagonal's share of the dropped digit pair (3, 3) (6-group plan): 128 doubles
  static constexpr int OFF_TAIL = OFF_TD + 128 * 8;             // the last N % 32 columns of X (fp64 rank-r term of the hand-over) + their y: 31 x 128 + 32 doubles
  // (dead ring, above the phase functions' image) the inverses of the eight diagonal blocks of L for the blocked back substitution: 8 x 16 x 16 doubles
  static constexpr int OFF_UW = 80 * 1024;
  static_assert(SmallCfg<double, 8>::LDS_BYTES <= OFF_UW && OFF_UW + 8 * 256 * 8 <= RING_BYTES, "the phase functions' LDS image and the block inverses must fit in the dead ring");
  static_assert(OFF_TAIL + (32 * 128 + 32) * 8 <= OFF_YB, "conversion tables and the tail columns / a block under repair must fit in the digit area");
  static_assert(OFF_TAIL + 4 * 32 * 33 * 8 <= OFF_YB, "the four mirror tiles of the conversion (one per diagonal block) use the same area, before the repair");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS of one CU");
};

// ---- which (tile, accumulator slots) a wave owns -----------------------------------------------------------------------------------
// item = tile (I, K) of 32 x 32 (I >= K), slots q0..q1, one accumulator of 16 registers per slot.
//   ordinary tile: slot q = digit group k = q: the products (s, t), s + t = k, both orders (min(k, 5) - max(0, k - 5) + 1 MFMAs per k-step);
//   symmetric diagonal tile (I == K, 6-group plan): P(t, s) = P(s, t)', so the group sum is Q_k + Q_k' + R_k with Q_k the pairs s < t and
//     R_k = P(k/2, k/2): slots 0 .. NG - 2 are Q_1 .. Q_{NG-1} (1 1 2 2 3 MFMAs), slots NG - 1 .. are R_0, R_2, R_4 (1 MFMA each);
//     the conversion adds Q_k(i, j) to entry (max, min) from BOTH halves of the accumulator (twice on the diagonal), R_k to the lower half.
// 7-group plan: 70 accumulators, 260 MFMAs per k-step (w0 35 + w4 29 | w1 35 + w5 29 | w2 33 + w6 32 | w3 34 + w7 33).
// 6-group plan: 68 accumulators (36 + 4 x 8), 174 MFMAs per k-step, dealt by tools/i8_plan_search.py: at most 9 accumulators per
// wave (the slicing shares the 256 registers), SIMD partners (w, w + 4) carrying 44 44 44 42 between them, 95 fragment reads per k-step.
// `phase`: a tile whose slots are split over waves is assembled in up to three rounds, a barrier between them -- its `first`
// item stores, the others add; the offset tables, the mean parts of the dropped pairs and the prior join in a pass of their own.
struct I8Item { int I, K, q0, q1, phase, first; };  // first: this item STORES its tile's entries (the tile's other items, in later phases, add)
template <int NG, int W> struct I8Items;
template <> struct I8Items<7, 0> { static constexpr int N = 2; static constexpr I8Item it[3] = {{0, 0, 0, 6, 0, 1}, {2, 0, 3, 4, 1, 0}, {0, 0, 0, -1, 0, 0}}; };
template <> struct I8Items<7, 1> { static constexpr int N = 2; static constexpr I8Item it[3] = {{1, 1, 0, 6, 0, 1}, {3, 1, 3, 4, 1, 0}, {0, 0, 0, -1, 0, 0}}; };
template <> struct I8Items<7, 2> { static constexpr int N = 2; static constexpr I8Item it[3] = {{3, 0, 2, 6, 0, 1}, {1, 0, 0, 3, 0, 1}, {0, 0, 0, -1, 0, 0}}; };
template <> struct I8Items<7, 3> { static constexpr int N = 3; static constexpr I8Item it[3] = {{3, 2, 2, 6, 0, 1}, {2, 0, 0, 2, 0, 1}, {2, 1, 6, 6, 1, 0}}; };
template <> struct I8Items<7, 4> { static constexpr int N = 2; static constexpr I8Item it[3] = {{2, 2, 0, 6, 0, 1}, {3, 0, 0, 1, 1, 0}, {0, 0, 0, -1, 0, 0}}; };
template <> struct I8Items<7, 5> { static constexpr int N = 2; static constexpr I8Item it[3] = {{3, 3, 0, 6, 0, 1}, {3, 2, 0, 1, 1, 0}, {0, 0, 0, -1, 0, 0}}; };
template <> struct I8Items<7, 6> { static constexpr int N = 2; static constexpr I8Item it[3] = {{2, 1, 0, 5, 0, 1}, {2, 0, 5, 6, 2, 0}, {0, 0, 0, -1, 0, 0}}; };
template <> struct I8Items<7, 7> { static constexpr int N = 3; static constexpr I8Item it[3] = {{1, 0, 4, 6, 1, 0}, {3, 1, 5, 6, 2, 0}, {3, 1, 0, 2, 0, 1}}; };
// 6-group plan: the phases are a 3-colouring of the (tile, wave) incidence -- no wave converts two items in one phase, no tile is touched
// by two waves in one phase -- and a tile's earliest item stores
template <> struct I8Items<6, 0> { static constexpr int N = 2; static constexpr I8Item it[3] = {{3, 2, 3, 5, 1, 0}, {3, 3, 0, 4, 0, 1}, {0, 0, 0, -1, 0, 0}}; };
template <> struct I8Items<6, 1> { static constexpr int N = 3; static constexpr I8Item it[3] = {{3, 0, 0, 5, 2, 1}, {3, 1, 0, 1, 0, 1}, {3, 3, 5, 5, 1, 0}}; };
template <> struct I8Items<6, 2> { static constexpr int N = 3; static constexpr I8Item it[3] = {{3, 1, 2, 5, 1, 0}, {3, 2, 0, 2, 0, 1}, {3, 3, 6, 7, 2, 0}}; };
template <> struct I8Items<6, 3> { static constexpr int N = 3; static constexpr I8Item it[3] = {{0, 0, 1, 2, 0, 1}, {2, 0, 0, 3, 1, 0}, {2, 2, 0, 2, 2, 0}}; };
template <> struct I8Items<6, 4> { static constexpr int N = 3; static constexpr I8Item it[3] = {{0, 0, 0, 0, 1, 0}, {1, 0, 4, 5, 0, 1}, {1, 1, 3, 7, 2, 0}}; };
template <> struct I8Items<6, 5> { static constexpr int N = 2; static constexpr I8Item it[3] = {{2, 0, 4, 5, 0, 1}, {2, 2, 3, 7, 1, 1}, {0, 0, 0, -1, 0, 0}}; };
template <> struct I8Items<6, 6> { static constexpr int N = 2; static constexpr I8Item it[3] = {{0, 0, 3, 7, 2, 0}, {1, 0, 0, 3, 1, 0}, {0, 0, 0, -1, 0, 0}}; };
template <> struct I8Items<6, 7> { static constexpr int N = 2; static constexpr I8Item it[3] = {{1, 1, 0, 2, 0, 1}, {2, 1, 0, 5, 1, 1}, {0, 0, 0, -1, 0, 0}}; };

template <int NG> constexpr bool i8_sym_tile(int I, int K) { return I8Mode<NG>::SYM && I == K; }
// slot -> kind (0: whole group, entries i >= j; 1: Q, mirrored; 2: R, entries i >= j), digit group k (scale 2^(80 - 8 k)), membership of (s, t)
template <int NG> constexpr int i8_slot_kind(int I, int K, int q) { return !i8_sym_tile<NG>(I, K) ? 0 : (q < NG - 1 ? 1 : 2); }
template <int NG> constexpr int i8_slot_k(int I, int K, int q) { return !i8_sym_tile<NG>(I, K) ? q : (q < NG - 1 ? q + 1 : 2 * (q - (NG - 1))); }
template <int NG> constexpr bool i8_slot_has(int I, int K, int q, int s, int t) {
  if (!i8_sym_tile<NG>(I, K)) return s + t == q;
  if (q < NG - 1) return s < t && s + t == q + 1;
  return s == t && s == q - (NG - 1);
}

struct I8Mma { int item, s, t; };
struct I8PlanTable {
  int nm = 0;
  I8Mma mm[96] = {};
  int first_use[4][6] = {};
  int acc_index[96] = {};
  int acc_base[4] = {};
  int nacc = 0;
};
template <int NG, int W>
struct I8Plan {
  using Items = I8Items<NG, W>;
  static constexpr int NI = Items::N;
  // the k-step's MFMA list, built ONCE at compile time: items in order, inside an item s-major from the highest slice
  // (fragment live ranges stay short)
  static constexpr I8PlanTable build() {
    I8PlanTable p;
    for (int IT = 0; IT < NI; ++IT) {
      const I8Item it = Items::it[IT];
      p.acc_base[IT] = p.nacc;
      for (int s = 5; s >= 0; --s)
        for (int t = 0; t <= 5; ++t)
          for (int q = it.q0; q <= it.q1; ++q)
            if (i8_slot_has<NG>(it.I, it.K, q, s, t)) {
              p.mm[p.nm] = I8Mma{IT, s, t};
              p.acc_index[p.nm] = p.nacc + q - it.q0;
              ++p.nm;
            }
      p.nacc += it.q1 - it.q0 + 1;
    }
    p.acc_base[NI] = p.nacc;
    for (int rb = 0; rb < 4; ++rb)
      for (int sl = 0; sl < 6; ++sl) {
        p.first_use[rb][sl] = p.nm;
        for (int i = p.nm - 1; i >= 0; --i) {
          const I8Item it = Items::it[p.mm[i].item];
          if ((it.I == rb && p.mm[i].s == sl) || (it.K == rb && p.mm[i].t == sl)) p.first_use[rb][sl] = i;
        }
      }
    return p;
  }
  static constexpr I8PlanTable T = build();
  static constexpr int NM = T.nm;
  static constexpr int NACC = T.nacc;
  static constexpr int acc_base(int IT) { return T.acc_base[IT]; }
  static constexpr int rowblock(int i, int side) { return side == 0 ? Items::it[T.mm[i].item].I : Items::it[T.mm[i].item].K; }
  static constexpr int slice(int i, int side) { return side == 0 ? T.mm[i].s : T.mm[i].t; }
  static constexpr int first_use(int rb, int sl) { return T.first_use[rb][sl]; }
  static constexpr int acc_index(int i) { return T.acc_index[i]; }
};
// (the plans as stated above)
static_assert(I8Plan<6, 0>::NM + I8Plan<6, 1>::NM + I8Plan<6, 2>::NM + I8Plan<6, 3>::NM + I8Plan<6, 4>::NM + I8Plan<6, 5>::NM + I8Plan<6, 6>::NM +
                  I8Plan<6, 7>::NM == kI8MfmaPerKstep, "6-group plan: every product exactly once");
static_assert(I8Plan<6, 0>::NACC + I8Plan<6, 1>::NACC + I8Plan<6, 2>::NACC + I8Plan<6, 3>::NACC + I8Plan<6, 4>::NACC + I8Plan<6, 5>::NACC +
                  I8Plan<6, 6>::NACC + I8Plan<6, 7>::NACC == 68, "6-group plan: 36 + 4 x 8 accumulators");
static_assert(I8Plan<7, 0>::NM + I8Plan<7, 1>::NM + I8Plan<7, 2>::NM + I8Plan<7, 3>::NM + I8Plan<7, 4>::NM + I8Plan<7, 5>::NM + I8Plan<7, 6>::NM +
                  I8Plan<7, 7>::NM == kI8MfmaPerKstep7, "7-group plan: every product exactly once");

__device__ __forceinline__ i32x4 lds_read_b128(const char* p) { return *reinterpret_cast<const i32x4*>(p); }

// Balanced digits.  The six bytes of Q' = Q + 0x8080808080 (top one signed, lower five minus 128 -- one XOR) are digits b_s in
// [-128, 127] with Q = sum_s b_s 2^(8 (5 - s)) and NO offsets: Q' - 0x8080808080 = d'_5 2^40 + sum_{k<5} (d'_k - 128) 2^(8k).  Q' comes out
// of the SAME single addition that rounds x to the row's grid: the offset is an (even) integer of that grid and rides in the low mantissa
// bits of the magic constant, so Q is still round-to-nearest-even(x 2^(47 - e_r)).  (Until round 5 the digits were the bytes of Q itself
// minus 128: every product then carried rank-one offset terms -- digit row sums from six v_dot4 per column quad and a table pass at the
// hand-over -- and low digits that are constant (float32 / integer / power-of-two inputs) were -128, so the DROPPED products were systematic
// and their mean parts had to be kept too.  Balanced, a constant-zero low digit is 0 and its products vanish; what is dropped is zero-mean
// for anything but inputs whose low digits are a fixed non-zero pattern, e.g. every entry of a row = integer + 1/3: 1e-13 of the diagonal
// scale there, tools/i8_digits_emul.py.)
constexpr long long kI8Balance = 0x8080808080LL;
__device__ __forceinline__ double i8_magic(int eb /*biased exponent of the row's capacity*/) {
  return __hiloint2double((int)(((unsigned)(eb + 5) << 20) | 0x00080080u), (int)0x80808080u);  // mantissa 2^51 + 0x8080808080 at exponent e_r + 5
}
// x + C fits the signed 48-bit integer  <=>  the mantissa of the sum lies in [2^51 - 2^47, 2^51 + 2^47) at exponent e_r + 5  <=>  its HIGH
// WORD, as an unsigned integer, lies in [hlo, hlo + 0xffff] (a too-large |x| changes the exponent field or the top mantissa bits, a
// negative sum sets the sign bit, Inf / NaN have the largest exponent field: all outside).  So the capacity test of the stream is the
// running minimum and maximum of the high words the slicing holds anyway: one v_min3_u32 + one v_max3_u32 per PAIR of entries.
__device__ __forceinline__ unsigned i8_hlo(int eb) { return ((unsigned)(eb + 5) << 20) | 0x00078000u; }
__device__ __forceinline__ bool i8_fits(unsigned hi_of_sum, unsigned hlo) { return hi_of_sum - hlo <= 0xffffu; }

// per-thread slicing state: row r = tid & 127, column octet cq = tid >> 7 of every k-step
struct I8Slice {
  double C;          // magic constant 1.5 2^(e_r + 5) + 0x8080808080 2^(e_r - 47): the integer in the low mantissa bits is Q + 0x8080808080
  unsigned hlo;      // the high words of x + C that stand for a 48-bit integer: [hlo, hlo + 0xffff] (i8_hlo)
  unsigned hmin, hmax;  // running minimum / maximum of the high words of x + C over the block being sliced (mark_block)
  int sq3;           // sum of a_3^2 over this thread's columns (6-group plan: the dropped product (3, 3) is not zero-mean on the diagonal)
  double b;          // sum_n x_rn y_n over this thread's columns
  double q;          // sum_n y_n^2 over this thread's columns (waves that hold row 0 only)
};

// The slicing of one 32-column block in 14 chunks (state between chunks in this struct).
//   x + C: rounded to a multiple of 2^(e_r - 47), its integer Q in the low mantissa bits; bytes 0..3 of Q are bytes 0..3 of the
//   low word, bytes 4, 5 are bytes 0, 1 of the high word; slice s holds byte 5 - s.  4 x 4 byte transposition of a column quad
//   with v_perm_b32 (D = perm(S0, S1, sel): selector values 0-3 take bytes of S1, 4-7 bytes of S0).
// ROWV: the inputs are RowVecs (N x D column-major: a feature's values over the observations are contiguous).  A k-step's raw block
// is then 32 LDS-DMA pieces of FOUR rows d = 4 p .. 4 p + 3 x 32 observations (16 lanes x 16 bytes per row), and within piece p the pair
// of observations l16 of row q sits at 16-byte position ((l16 + p) mod 16) 4 + q: a wave reads one pair for 64 consecutive rows,
// and the rotation by p puts the 16 lanes of every ds_read_b128 group on 16 different slots of the 256-byte bank row.
template <bool WITH_Q, bool DIAG = false, bool ROWV = false, bool WITH_SQ3 = false>
struct I8SliceSteps {
  double x[2][4], y[2][4];  // two column quads: the second quad's LDS reads are in flight while the first is sliced
  double w[DIAG ? 2 : 1][4];  // diagonal noise: 1 / sqrt(s_n) of the columns (x enters the Gram matrix as x / sqrt(s_n), y as y / sqrt(s_n))
  unsigned lo[4], hi[4], u[6], p[2][6];
  template <int Q>
  __device__ __forceinline__ void load(const char* __restrict__ raw, const double* __restrict__ yb, const double* __restrict__ wb, int r, int cq) {
    if constexpr (ROWV) {
      typedef double d2 __attribute__((ext_vector_type(2)));
      const int p = r >> 2, q = r & 3;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int l16 = cq * 4 + 2 * Q + hh;
        const d2 v = *reinterpret_cast<const d2*>(raw + p * 1024 + ((((l16 + p) & 15) << 2) + q) * 16);
        x[Q][2 * hh] = v[0];
        x[Q][2 * hh + 1] = v[1];
      }
    } else {
      const double* col = reinterpret_cast<const double*>(raw) + (cq * 8 + 4 * Q) * 128 + r;
#pragma unroll
      for (int j = 0; j < 4; ++j) x[Q][j] = col[j * 128];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      y[Q][j] = yb[cq * 8 + 4 * Q + j];  // (y: one address per wave, a broadcast)
      if constexpr (DIAG) w[Q][j] = wb[cq * 8 + 4 * Q + j];
    }
  }
  // chunk 0: reads of quad 0; chunks 1 + 6 q + c: quad q -- c = 0 reads of quad q + 1, c = 1, 2 magic add / bound check / b and q (two
  // columns each), c = 3, 4 byte transposition, c = 5 digit row sums; chunk 13: digit planes out
  template <int C>
  __device__ __forceinline__ void step(const char* __restrict__ raw, const double* __restrict__ yb, const double* __restrict__ wb, char* __restrict__ dig, int r,
                                       int cq, I8Slice& st) {
    if constexpr (C == 0) {
      load<0>(raw, yb, wb, r, cq);
    } else if constexpr (C < 13) {
      constexpr int q = (C - 1) / 6, c = (C - 1) % 6;  // column quad q of the thread's 8 columns
      if constexpr (c == 0 && q < 1) load<q + 1>(raw, yb, wb, r, cq);
      if constexpr (c == 1 || c == 2) {
        constexpr int j0 = c == 1 ? 0 : 2;
#pragma unroll
        for (int j = j0; j < j0 + 2; ++j) {
          double xv = x[q][j];
          if constexpr (DIAG) xv = __dmul_rn(xv, w[q][j]);  // (never contracted into the magic addition: the repair recomputes these bits)
          const double yv = y[q][j];
          const double t = __dadd_rn(xv, st.C);
          lo[j] = (unsigned)__double2loint(t);
          hi[j] = (unsigned)__double2hiint(t);
          st.b = __builtin_fma(xv, yv, st.b);
          if constexpr (WITH_Q) st.q = __builtin_fma(yv, yv, st.q);
        }
        // capacity test: the range of the sums' high words (v_min3_u32, v_max3_u32)
        st.hmin = min(st.hmin, min(hi[j0], hi[j0 + 1]));
        st.hmax = max(st.hmax, max(hi[j0], hi[j0 + 1]));
        asm volatile("" : "+v"(st.hmin), "+v"(st.hmax));
        // (pinned: nothing reads these sums before the end of the stream, and hipcc would sink their updates out of the
        // MFMA shadows to behind the k-step's barrier)
        asm volatile("" : "+v"(st.b));
        if constexpr (WITH_Q) asm volatile("" : "+v"(st.q));
      }
      if constexpr (c == 3) {  // byte transposition, first stage
        u[0] = __builtin_amdgcn_perm(lo[1], lo[0], 0x05010400u);  // b0(l0) b0(l1) b1(l0) b1(l1)
        u[1] = __builtin_amdgcn_perm(lo[1], lo[0], 0x07030602u);  // b2(l0) b2(l1) b3(l0) b3(l1)
        u[2] = __builtin_amdgcn_perm(lo[3], lo[2], 0x05010400u);
        u[3] = __builtin_amdgcn_perm(lo[3], lo[2], 0x07030602u);
        u[4] = __builtin_amdgcn_perm(hi[1], hi[0], 0x05010400u);  // b4(0) b4(1) b5(0) b5(1)
        u[5] = __builtin_amdgcn_perm(hi[3], hi[2], 0x05010400u);
      }
      if constexpr (c == 4) {  // second stage; bytes of Q + 0x8080808080 minus 128 (one XOR) are the balanced digits, the top one is signed as it is
        p[q][5] = __builtin_amdgcn_perm(u[2], u[0], 0x05040100u) ^ 0x80808080u;
        p[q][4] = __builtin_amdgcn_perm(u[2], u[0], 0x07060302u) ^ 0x80808080u;
        p[q][3] = __builtin_amdgcn_perm(u[3], u[1], 0x05040100u) ^ 0x80808080u;
        p[q][2] = __builtin_amdgcn_perm(u[3], u[1], 0x07060302u) ^ 0x80808080u;
        p[q][1] = __builtin_amdgcn_perm(u[5], u[4], 0x05040100u) ^ 0x80808080u;
        p[q][0] = __builtin_amdgcn_perm(u[5], u[4], 0x07060302u);
      }
      if constexpr (c == 5) {  // sum of squares of digit 3 (the diagonal's share of the dropped pair (3, 3))
        if constexpr (WITH_SQ3) {
          st.sq3 = __builtin_amdgcn_sdot4((int)p[q][3], (int)p[q][3], st.sq3, false);
          asm volatile("" : "+v"(st.sq3));
        }
      }
    } else {  // C == 13: digit planes out.  Fragment (s, I = r >> 5): lane (half = cq >> 1, row r & 31) at byte 16 (32 half + (r & 31));
              // this thread's 8 columns are bytes 8 (cq & 1) .. + 8 of the lane's 16
#ifdef BLR_I8_DIG_PLANES  /* measured, not shipped: the fragment as TWO planes of 8 bytes per lane, 512 bytes apart -- a wave's 32 rows write
                             consecutive words, where the 16-byte slots of a single plane put lanes l and l + 16 on the same banks (2-way
                             conflicts: the kernel's 1.0e8 SQ_LDS_BANK_CONFLICT cycles per launch, 190 LDS cycles of a 2900-cycle k-step).
                             Conflict-free on both sides with one ds_read2_b64 per fragment -- and 1 % SLOWER (4.62 against 4.56 ms per
                             4096 updates, same box, twice): the LDS pipe is not what the k-step waits for. */
      char* dst = dig + (r >> 5) * 1024 + (cq & 1) * 512 + (((cq >> 1) * 32 + (r & 31)) * 8);
#else
      char* dst = dig + (r >> 5) * 1024 + (((cq >> 1) * 32 + (r & 31)) * 16) + (cq & 1) * 8;
#endif
#pragma unroll
      for (int s = 0; s < 6; ++s) {
        uint2 v;
        v.x = p[0][s]; v.y = p[1][s];
        *reinterpret_cast<uint2*>(dst + s * 4096) = v;
      }
    }
  }
};
constexpr int kI8SliceChunks = 14;

// ---- who issues the LDS-DMA pieces of a k-step (32 columns of 1 KiB + the y values [+ 1 / sqrt(s) under diagonal noise]) ----------------
// BLR_I8_DMA_WAVES = 8 (rounds 4, 5): every wave its four columns.  = 4: waves 0 - 3 eight columns each, waves 4 - 7 none.  The two
// waves of a SIMD are arbitrated by age: the older one (waves 0 - 3) finishes its k-step ~ 600 cycles before its partner and waits at
// the barrier, so the ~ 60 issue cycles of a piece are free there and on the critical path on waves 4 - 7.  Waves that issue no piece
// have an empty vector-memory queue of their own and can run ahead of the stream with "touch" loads (BLR_I8_TOUCH = distance in
// k-steps beyond the LDS-DMA; one dword per 128-byte line, result discarded): the line travels HBM -> L2 then, and the LDS-DMA piece
// that follows finds it in the L2 -- an in-order queue would make the pieces wait for the touches issued before them (round 4 measured
// touches from the SAME waves: slower).
#ifndef BLR_I8_DMA_WAVES
#define BLR_I8_DMA_WAVES 4
#endif
#ifndef BLR_I8_TOUCH
#define BLR_I8_TOUCH 0
#endif
template <int W> struct I8Dma {
  static constexpr int COLS = BLR_I8_DMA_WAVES == 8 ? 4 : (W < 4 ? 8 : 0);  // column pieces of this wave per k-step
  static constexpr int SLOTS = COLS + 1;                                   // + the y piece(s) / the advance / the touch
};
// (Measured and not shipped: four pieces per M0 set-up -- consecutive KiB of the LDS through the instruction offset, 20 instead of 56
// instructions per k-step on a wave with eight pieces -- is SLOWER, 4.27 - 4.30 against 4.20 - 4.22 ms per 4096 updates on one box: the
// pieces then leave in bursts of four, and what the stream needs is an even trickle.)

// ---- one k-step, order pinned by hand ---------------------------------------------------------------------------------------------
// A 32 x 32 x 32 int8 MFMA holds the matrix pipe for 32 cycles; about five single-issue instructions fit in its shadow.  Left
// to itself hipcc emits the k-step's MFMAs back to back and the vector instructions of the slicing behind them (and
// sched_group_barrier does not move them: the slicing hangs off LDS reads the group solver leaves where they are).  So the
// k-step is a compile-time list: MFMA i, then (fenced with sched_barrier) the fragment reads MFMA i + 2 is the first to need
// and the slicing chunks whose turn it is -- reads of the raw columns first, their arithmetic two MFMAs later.
template <int NG, int W, bool SLICE, bool WITH_Q, bool DIAG, bool ROWV, typename IssueFn>
__device__ __forceinline__ void i8_kstep(const char* __restrict__ dig, const char* __restrict__ raw, const double* __restrict__ yb,
                                         const double* __restrict__ wb, char* __restrict__ dign, int lane, int r, int cq, i32x16 (&A)[I8Plan<NG, W>::NACC], I8Slice& st,
                                         IssueFn issue_next) {
  using PL = I8Plan<NG, W>;
  constexpr int NM = PL::NM;
  constexpr int NCH = kI8SliceChunks;
#ifndef BLR_I8_LEAD
#define BLR_I8_LEAD 2
#endif
  constexpr int LEAD = BLR_I8_LEAD;   // a fragment is requested this many MFMAs before its first use
  i32x4 F[4][6];            // fragment (row block, slice): only the ones this wave uses ever get registers
  I8SliceSteps<WITH_Q, DIAG, ROWV, NG == 6> sl;
  auto frag_load_one = [&](auto itag, auto qtag) {
    constexpr int i = decltype(itag)::value, q = decltype(qtag)::value, rb = q / 6, sidx = q % 6;
    constexpr int fu = PL::first_use(rb, sidx);
#ifdef BLR_I8_DIG_PLANES
    if constexpr (fu == i) {  // (one ds_read2_b64: the lane's 8 + 8 contraction bytes from the two planes)
      const uint2* pl = reinterpret_cast<const uint2*>(dig + (sidx * 4 + rb) * 1024 + lane * 8);
      const uint2 lo8 = pl[0], hi8 = pl[64];
      F[rb][sidx] = i32x4{(int)lo8.x, (int)lo8.y, (int)hi8.x, (int)hi8.y};
    }
#else
    if constexpr (fu == i) F[rb][sidx] = lds_read_b128(dig + (sidx * 4 + rb) * 1024 + lane * 16);
#endif
  };
  auto frag_loads_rec = [&](auto self, auto itag, auto qtag) -> void {  // the reads whose first use is MFMA i
    constexpr int q = decltype(qtag)::value;
    if constexpr (q < 24) {
      frag_load_one(itag, qtag);
      self(self, itag, std::integral_constant<int, q + 1>{});
    }
  };
  auto frag_loads = [&](auto itag) { frag_loads_rec(frag_loads_rec, itag, std::integral_constant<int, 0>{}); };
  auto chunk = [&](auto ctag) {
    constexpr int c = decltype(ctag)::value;
#if defined(BLR_I8_EXP) && BLR_I8_EXP == 3  /* timing experiment: no slicing */
    if constexpr (false) sl.template step<c>(raw, yb, wb, dign, r, cq, st);
#else
    if constexpr (SLICE && c >= 0 && c < NCH) sl.template step<c>(raw, yb, wb, dign, r, cq, st);
#endif
  };
  auto unit = [&](auto itag) {
    constexpr int i = decltype(itag)::value;
    if constexpr (i == 0) {  // at the head: everything the first LEAD MFMAs need
      frag_loads(std::integral_constant<int, 0>{});
      frag_loads(std::integral_constant<int, 1>{});
      if constexpr (LEAD >= 3) frag_loads(std::integral_constant<int, 2>{});
      if constexpr (LEAD >= 4) frag_loads(std::integral_constant<int, 3>{});
    }
    if constexpr (i + LEAD < NM) frag_loads(std::integral_constant<int, i + LEAD>{});
    __builtin_amdgcn_sched_barrier(0);
    constexpr int ai = PL::acc_index(i);
    constexpr int ra = PL::rowblock(i, 0), sa = PL::slice(i, 0), rbk = PL::rowblock(i, 1), sb = PL::slice(i, 1);
#if defined(BLR_I8_EXP) && BLR_I8_EXP == 2  /* timing experiment: no MFMA */
    A[ai][0] += F[ra][sa][0] + F[rbk][sb][1];
#else
    A[ai] = __builtin_amdgcn_mfma_i32_32x32x32_i8(F[ra][sa], F[rbk][sb], A[ai], 0, 0, 0);
#endif
    __builtin_amdgcn_sched_barrier(0);
    // the LDS-DMA pieces of the k-step three ahead (their slot was freed by the barrier that opened this k-step) go out one at a
    // time, every sixth MFMA (~60 cycles of issue each: beside the partner wave's MFMAs, not in a bunch behind the barrier)
    // (every sixth MFMA where the wave has 27 or more of them; waves of the 6-group plan carry 17 to 26: closer together)
    constexpr int NSLOTS = I8Dma<W>::SLOTS;  // issue slots of this wave per k-step (its LDS-DMA pieces, then the advance to the next k-step)
    constexpr int ISTEP = (NM - 2) / NSLOTS >= 6 ? 6 : ((NM - 2) / NSLOTS >= 1 ? (NM - 2) / NSLOTS : 1);
    static_assert(2 + (NSLOTS - 1) * ISTEP < NM, "all LDS-DMA pieces of a k-step must find a slot among the wave's MFMAs");
    if constexpr (i >= 2 && (i - 2) % ISTEP == 0 && (i - 2) / ISTEP < NSLOTS) issue_next(std::integral_constant<int, (i - 2) / ISTEP>{});
    // slicing chunks c with floor(c NM / NCH) == i
    constexpr int c_lo = (i * NCH + NM - 1) / NM, c_hi = ((i + 1) * NCH + NM - 1) / NM;
    if constexpr (c_lo < c_hi) chunk(std::integral_constant<int, c_lo>{});
    if constexpr (c_lo + 1 < c_hi) chunk(std::integral_constant<int, c_lo + 1>{});
    __builtin_amdgcn_sched_barrier(0);
  };
  auto run_units = [&](auto self, auto itag) -> void {
    constexpr int i = decltype(itag)::value;
    if constexpr (i < NM) {
      unit(itag);
      self(self, std::integral_constant<int, i + 1>{});
    }
  };
  run_units(run_units, std::integral_constant<int, 0>{});
}

// ---- the stream: on exit the wave's accumulators and the slicing state --------------------------------------------------------------
template <int NG, int W, bool DIAG, bool ROWV>
__device__ __forceinline__ void i8_gram_stream(char* smem, const BLR_GLOBAL double* X /*uniform*/, const BLR_GLOBAL double* y /*uniform*/,
                                               const BLR_GLOBAL double* rw /*uniform; DIAG: 1 / sqrt(s_n)*/, double rwmax, int64_t ldx, int N, int tid, i32x16 (&A)[I8Plan<NG, W>::NACC], I8Slice& st, int& ok) {
  using C = I8Cfg;
  const int lane = tid & 63;
  const int r = tid & 127, cq = tid >> 7;
  char* const ring = smem;
  char* const dig0 = smem + C::OFF_DIG;
  double* const yring = reinterpret_cast<double*>(smem + C::OFF_YB);
  double* const wring = reinterpret_cast<double*>(smem + C::OFF_RW);
  int* const xch = reinterpret_cast<int*>(smem + C::OFF_XCH);
  unsigned char* const oflag = reinterpret_cast<unsigned char*>(smem + C::OFF_OFLAG);
  const int nk = N / C::KC;
  // an entry of the block just sliced outgrew its row's capacity (or is Inf / NaN): its digits are those of a wrapped value; the block
  // is marked and put right in fp64 at the hand-over (i8_repair_block).  The running maximum starts afresh for the next block.
  auto mark_block = [&](int blk) {
    if (__builtin_amdgcn_ballot_w64(!i8_fits(st.hmin, st.hlo) || !i8_fits(st.hmax, st.hlo)) != 0ull) {
      if (lane == 0) oflag[blk] = 1;
    }
    st.hmin = 0xffffffffu;
    st.hmax = 0u;
  };
  if (tid < 136) reinterpret_cast<int*>(oflag)[tid] = 0;
  // LDS-DMA pieces per wave and k-step (I8Dma): its columns (+ the y piece of wave 0, + the 1 / sqrt(s) piece of wave 1 under diagonal noise)
  constexpr int NCOL = I8Dma<W>::COLS;
  constexpr int PW = NCOL + (W == 0 ? 1 : 0) + (DIAG && W == 1 ? 1 : 0);
  constexpr bool kTouch = BLR_I8_TOUCH > 0 && NCOL == 0;  // this wave issues no piece: it runs ahead of the stream with touch loads
  unsigned ring_addr = lds_addr_of(ring), y_addr = lds_addr_of(yring), w_addr = lds_addr_of(wring);
  asm volatile("" : "+v"(ring_addr), "+v"(y_addr), "+v"(w_addr));
  const uint64_t colbytes = (uint64_t)ldx * 8u;
  // ColVecs: this wave's columns of the k-step being issued.  RowVecs: the k-step's 32 observations of row 0; piece c of
  // this wave = rows 4 (NCOL W + c) .. + 3 (colbytes = the distance between two rows), lane -> (row lane & 3, pair of observations
  // ((lane >> 2) - piece) mod 16): the rotation of the LDS image (I8SliceSteps)
  uint64_t nextX = (uint64_t)(uintptr_t)X + (ROWV ? (uint64_t)0 : (uint64_t)(NCOL * W) * colbytes);
  unsigned voffR[NCOL > 0 ? NCOL : 1];
#pragma unroll
  for (int c = 0; c < NCOL; ++c) voffR[c] = (unsigned)(lane & 3) * (unsigned)colbytes + (unsigned)((((lane >> 2) - (NCOL * W + c)) & 15) * 16);
  uint64_t nextY = (uint64_t)(uintptr_t)y;
  uint64_t nextW = (uint64_t)(uintptr_t)rw;
  const unsigned voff = (unsigned)lane * 16u;
  // touch loads (waves without pieces): one dword per 128-byte line of block t + BLR_I8_TOUCH, a quarter of the block per wave --
  // ColVecs: columns 8 (W - 4) .. + 7, eight lines each; RowVecs: rows 32 (W - 4) .. + 31, two lines each
  uint64_t touchX = (uint64_t)(uintptr_t)X;
  unsigned voffT = 0, tdummy = 0;
  if constexpr (kTouch) {
    voffT = ROWV ? (unsigned)(32 * (W - 4) + (lane >> 1)) * (unsigned)colbytes + (unsigned)(lane & 1) * 128u
                 : (unsigned)(8 * (W - 4) + (lane >> 3)) * (unsigned)colbytes + (unsigned)(lane & 7) * 128u;
    const int t0 = (BLR_I8_TOUCH < nk) ? BLR_I8_TOUCH : nk - 1;  // the block touched when k-step 0 is issued (the prologue issues k-steps 0 .. 2)
    touchX += (uint64_t)t0 * (ROWV ? (uint64_t)C::KC * 8u : (uint64_t)C::KC * colbytes);
  }
  auto touch = [&](uint64_t base) {
    asm volatile("global_load_dword %0, %1, %2" : "+v"(tdummy) : "v"(voffT), "s"(uni((int64_t)base)) : "memory");
  };
  // piece c of k-step t -> ring slot t % 3: c = 0 .. NCOL - 1 this wave's columns, c = NCOL the y values (wave 0) and the advance to the
  // next k-step (pieces are issued in order: the global addresses just run on)
  auto issue_piece = [&](int t, auto ctag) {
    constexpr int c = decltype(ctag)::value;
#if defined(BLR_I8_EXP) && BLR_I8_EXP == 1  /* timing experiment: no DMA after the first three k-steps (the ring keeps them) */
    if (t >= 3) return;
#endif
    if constexpr (c < NCOL) {
      const unsigned slot = ring_addr + (unsigned)((t % C::NSLOT) * C::SLOT_BYTES) + (unsigned)((NCOL * W + c) * 1024);
      if constexpr (ROWV) glds_s<16, 64, true>(uni((int64_t)(nextX + (uint64_t)(4 * (NCOL * W + c)) * colbytes)), voffR[c], slot);
      else glds_s<16, 64, true>(uni((int64_t)(nextX + (uint64_t)c * colbytes)), voff, slot);
    } else if constexpr (c == NCOL) {
      if constexpr (W == 0) glds_s<4, 64>(uni((int64_t)nextY), (unsigned)lane * 4u, y_addr + (unsigned)((t % C::NSLOT) * C::KC * 8));
      if constexpr (DIAG && W == 1) glds_s<4, 64>(uni((int64_t)nextW), (unsigned)lane * 4u, w_addr + (unsigned)((t % C::NSLOT) * C::KC * 8));
      // (pieces are issued for three k-steps beyond the last one too -- a branch per piece in the k-step split its schedule: 4.73 -> 4.62 ms
      // per 4096 updates without them; they load the last block again, from L2: the addresses stop advancing there)
      const uint64_t adv = (t + 1 < nk) ? ~(uint64_t)0 : (uint64_t)0;
      nextX += (ROWV ? (uint64_t)C::KC * 8u : (uint64_t)C::KC * colbytes) & adv;
      nextY += ((uint64_t)C::KC * 8u) & adv;
      nextW += ((uint64_t)C::KC * 8u) & adv;
      if constexpr (kTouch) {
        touch(touchX);
        const uint64_t advT = (t + BLR_I8_TOUCH + 1 < nk) ? ~(uint64_t)0 : (uint64_t)0;
        touchX += (ROWV ? (uint64_t)C::KC * 8u : (uint64_t)C::KC * colbytes) & advT;
      }
    }
  };
  auto issue_rec = [&](auto self, int t, auto ctag) -> void {
    constexpr int c = decltype(ctag)::value;
    if constexpr (c <= NCOL) {
      issue_piece(t, ctag);
      self(self, t, std::integral_constant<int, c + 1>{});
    }
  };
  auto issue = [&](int t) { issue_rec(issue_rec, t, std::integral_constant<int, 0>{}); };
  auto wait_keep = [&](int groups) {  // all but the youngest `groups` issued k-steps of THIS wave have landed
    if constexpr (PW == 0) {  // (no pieces of its own; its touch loads are never waited for inside the stream)
      if (groups == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      if (groups >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PW) : "memory");
      else if (groups == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };
#pragma unroll
  for (int g = 0; g < I8Plan<NG, W>::NACC; ++g)
#pragma unroll
    for (int v = 0; v < 16; ++v) A[g][v] = 0;

  const int nissue0 = nk < 3 ? nk : 3;
  for (int t = 0; t < nissue0; ++t) issue(t);
  wait_keep(0);
  __syncthreads();  // the first blocks visible
  // ---- row scales from the blocks of the prologue (96 columns): e_r = exponent of the row's largest entry there + 1 + margin.
  // (From block 0 alone -- 32 columns -- a row of N(0,1) entries stays below 1 sigma once in 2e5 rows and then accepts |x| < 4 sigma
  // only: about one regressor in 7000 handed back, every other launch of 4096 paying a lone fp64 update, +10 %.  Over 96 columns
  // the same event is 1e-16.)
  {
    unsigned m = 0;
    for (int t = 0; t < nissue0; ++t) {
      if constexpr (ROWV) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        const int p = r >> 2, q = r & 3;
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) {
          const d2 v = *reinterpret_cast<const d2*>(ring + t * C::SLOT_BYTES + p * 1024 + ((((cq * 4 + hh + p) & 15) << 2) + q) * 16);
          const unsigned a0 = (unsigned)__double2hiint(v[0]) & 0x7fffffffu, a1 = (unsigned)__double2hiint(v[1]) & 0x7fffffffu;
          m = a0 > m ? a0 : m;
          m = a1 > m ? a1 : m;
        }
      } else {
        const double* col = reinterpret_cast<const double*>(ring + t * C::SLOT_BYTES) + (cq * 8) * 128 + r;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const unsigned ax = (unsigned)__double2hiint(col[j * 128]) & 0x7fffffffu;
          m = ax > m ? ax : m;
        }
      }
    }
    xch[cq * 128 + r] = (int)m;
  }
  __syncthreads();
  {
    unsigned m = (unsigned)xch[r];
#pragma unroll
    for (int c = 1; c < 4; ++c) { const unsigned o = (unsigned)xch[c * 128 + r]; m = o > m ? o : m; }
    if constexpr (DIAG) {
      // what gets sliced is x / sqrt(s_n): its bound is the row's bound times the LARGEST 1 / sqrt(s_n) of the regressor (known from
      // the preparation pass).  Taken from the scaled values of the first 96 columns instead -- tried again in round 6, now that an entry
      // beyond its row's capacity is repaired rather than handed back -- the bound does not survive variances with a spread: with
      // s_n = exp(N(0,1)) some fifty of a regressor's 4096 columns have 1 / sqrt(s_n) > 3, most of them hold an entry beyond a capacity
      // of 4 - 8 x the largest of 96 scaled values, and 4078 of 4096 regressors went back to the fp64 kernel (11 ms against 5.1).
      const double mv = __hiloint2double((int)m, -1) * rwmax;
      m = (unsigned)__double2hiint(mv) & 0x7fffffffu;
    }
    int E1 = (int)(m >> 20);           // biased exponent of the row maximum (0 for a zero / denormal row)
    if (E1 > 1023 + 400) ok = 0;       // Inf / NaN / out of the range the final scaling can represent: fp64 path
    if (E1 < 1023 - 400) E1 = 1023 - 400;
    // biased e_r = E + CAP: the capacity of the row.  Q = round(x 2^(47 - e_r)); what the mantissa holds is Q + 0x8080808080 (the
    // balancing offset rides in the magic constant), which must stay a signed 48-bit integer (i8_hlo / i8_fits: tested on the sum itself)
    const int eb = E1 + I8Mode<NG>::CAP;
    st.C = i8_magic(eb);
    st.hlo = i8_hlo(eb);
    st.hmin = 0xffffffffu;
    st.hmax = 0u;
    st.b = 0.0;
    st.q = 0.0;
    st.sq3 = 0;
  }
  constexpr bool kQ = (W & 1) == 0;  // (rows 64 (W & 1) + lane: row 0 lives in the even waves)
  {  // block 0 -> digit buffer 0 (nothing to overlap with yet)
    I8SliceSteps<kQ, DIAG, ROWV, NG == 6> sl;
    auto rec = [&](auto self, auto ctag) -> void {
      constexpr int c = decltype(ctag)::value;
      if constexpr (c < kI8SliceChunks) {
        sl.template step<c>(ring, yring, wring, dig0, r, cq, st);
        self(self, std::integral_constant<int, c + 1>{});
      }
    };
    rec(rec, std::integral_constant<int, 0>{});
  }
  mark_block(0);
  if (nk > 1) wait_keep(nissue0 - 2 > 0 ? nissue0 - 2 : 0);
  __syncthreads();  // digits of block 0 and raw block 1 visible; raw block 0 consumed by everyone
  I8_STAMP_DECL;
#ifdef BLR_I8_SETPRIO
  // the second-dispatched half of the workgroup (waves 4 - 7) loses the issue arbitration on its SIMD to its older partner in every
  // k-step (MI355X_MICROARCH.md, two waves per SIMD): one static priority for the whole stream
  if constexpr (W >= 4) __builtin_amdgcn_s_setprio(BLR_I8_SETPRIO);
#endif
  // (the last k-step has nothing to slice: peeled, so that the loop body is ONE basic block and the accumulators stay in place)
#pragma unroll 1
  for (int j = 0; j + 1 < nk; ++j) {
    const char* dig = dig0 + (j & 1) * C::DIG_BUF;
    char* dign = dig0 + ((j + 1) & 1) * C::DIG_BUF;
    const char* raw = ring + ((j + 1) % C::NSLOT) * C::SLOT_BYTES;
    const double* yb = yring + ((j + 1) % C::NSLOT) * C::KC;
    const double* wb = wring + ((j + 1) % C::NSLOT) * C::KC;
    // k-step j + 3 goes into the slot of block j, which everybody has sliced before the barrier that ended k-step j - 1
    i8_kstep<NG, W, true, kQ, DIAG, ROWV>(dig, raw, yb, wb, dign, lane, r, cq, A, st, [&](auto ctag) { issue_piece(j + 3, ctag); });
    mark_block(j + 1);
    I8_STAMP(0);
    // end of k-step j: raw block j + 2 must have landed (k-step j + 3 may stay in flight), then everybody's is visible
    wait_keep(1);
    __syncthreads();
    I8_STAMP(2);
  }
  i8_kstep<NG, W, false, kQ, DIAG, ROWV>(dig0 + ((nk - 1) & 1) * C::DIG_BUF, ring, yring, wring, dig0, lane, r, cq, A, st, [](auto) {});
  wait_keep(0);  // (the pieces issued beyond the last k-step land in a ring that is about to be reused)
  if constexpr (kTouch) asm volatile("" ::"v"(tdummy));  // (the touch loads' destination register stays reserved until they have all returned)
#ifdef BLR_I8_SETPRIO
  if constexpr (W >= 4) __builtin_amdgcn_s_setprio(0);
#endif
  I8_STAMP_FLUSH(W);
}

// ---- back substitution m = L^-T u for D = 128, blocked by 16 ------------------------------------------------------------------------------
// phase_backsolve (blr_fused_small.hpp) walks the 128 pivots one after the other on one wave: 312 cycles per pivot, 40 k cycles per
// regressor with the matrix pipe and seven waves idle.  Here the chain is 8 block steps: the inverses W_J = L_JJ^-1 of the eight 16 x 16
// diagonal blocks first (all at once: one column per lane, 128 lanes), then for J = 7 .. 0
//     m_J = W_J' r_J      (16 x 16 product: lane = column, r_J by v_readlane)
//     r_K -= L_JK' m_J    for the rows K < J (lane = column of L, m_J by v_readlane)
// on wave 0, while waves 1 - 3 write T = L'.  Same interface as phase_backsolve: on entry P = L (packed), bvec = u; on exit bvec = m,
// scr[6] = |u|^2, scr[7] = logdet A.  Four-wave code (the waves that survive the hand-over).
static __device__ __attribute__((noinline, not_tail_called)) void i8_backsolve_blocked(char* smem, double* Tout_in, int64_t ldt_in) {
  using SC = SmallCfg<double, 8>;
  constexpr int D = 128;
  double* const P = reinterpret_cast<double*>(smem);
  double* const bvec = reinterpret_cast<double*>(smem + SC::OFF_B);
  double* const dinv = reinterpret_cast<double*>(smem + SC::OFF_DINV);
  double* const scr = reinterpret_cast<double*>(smem + SC::OFF_SCR);
  double* const Wst = reinterpret_cast<double*>(smem + I8Cfg::OFF_UW);  // [8 blocks][16 rows i][16 columns c]
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int wave = uni(tid >> 6);
  BLR_GLOBAL double* const Tout = as_global(uni(Tout_in));
  const int64_t ldt = uni(ldt_in);
  BLR_BS_STAMP_DECL;
  if (tid < D) dinv[tid] = 1.0 / P[pidx(tid, tid)];
  __syncthreads();
  if (tid < D) {  // W = L_JJ^-1, column c per lane: forward substitution on e_c, sixteen steps in lockstep
    const int blk = tid >> 4, c = tid & 15;
    double w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const double* row = P + pidx(16 * blk + i, 16 * blk);
      double s2[2] = {(i == c) ? 1.0 : 0.0, 0.0};
#pragma unroll
      for (int k = 0; k < i; ++k) s2[k & 1] = __builtin_fma(-row[k], w[k], s2[k & 1]);
      w[i] = (s2[0] + s2[1]) * dinv[16 * blk + i];
      Wst[(blk * 16 + i) * 16 + c] = w[i];
    }
  }
  __syncthreads();
  BLR_BS_STAMP(12);
  if (wave != 0) {
    // T = L' (upper, column-major; strictly-lower part zero) goes out while wave 0 runs the substitution (:67, chol(...).U)
    if (Tout != nullptr) {
      for (int c = wave - 1; c < D; c += kWaves - 1) {
        const double* row = P + pidx(c, 0);
        BLR_GLOBAL double* out = Tout + (int64_t)c * ldt;
        const int r0 = lane, r1 = lane + 64;
        out[r0] = (r0 <= c) ? row[r0] : 0.0;
        out[r1] = (r1 <= c) ? row[r1] : 0.0;
      }
    }
  } else {
    double b0 = bvec[lane], b1 = bvec[lane + 64];
    double uu = wave_allreduce(b0 * b0 + b1 * b1);
    const int c = lane & 15, grp = lane >> 4;
#pragma unroll 1
    for (int J = 7; J >= 0; --J) {
      const bool hi = J >= 4;   // (uniform) block J lives in b1 (rows 64 ..) or in b0
      const int jb = J & 3;     // its 16-lane group there
      const double rs = hi ? b1 : b0;
      // (1) m_J = W_J' r_J: column c of W_J (zero above the diagonal: a term with i < c vanishes)
      const double* wc = Wst + J * 256 + c;
      double a4[4] = {0.0, 0.0, 0.0, 0.0};  // (four independent chains: a dependent fp64 FMA waits ~4 issue slots for its predecessor)
#pragma unroll
      for (int i = 0; i < 16; ++i) a4[i & 3] = __builtin_fma(wc[i * 16], readlane(rs, 16 * jb + i), a4[i & 3]);
      const double acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
      if (grp == jb) { if (hi) b1 = acc; else b0 = acc; }
      if (J == 0) break;
      // (2) the rows above: r_c -= sum_i L(16 J + i, c) m_i for c < 16 J
      const double ms = hi ? b1 : b0;
      const double* rowJ = P + pidx(16 * J, 0);
      const bool act0 = lane < 16 * J, act1 = lane + 64 < 16 * J;
      double d0[2] = {0.0, 0.0}, d1[2] = {0.0, 0.0};
      int ro = 0;  // offset of row 16 J + i within the packed triangle, relative to row 16 J
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const double mi = readlane(ms, 16 * jb + i);
        d0[i & 1] = __builtin_fma(rowJ[ro + lane], mi, d0[i & 1]);
        if (hi) d1[i & 1] = __builtin_fma(rowJ[ro + lane + 64], mi, d1[i & 1]);  // (uniform; J <= 4: nothing of b1 lies above block J)
        ro += 16 * J + i + 1;
      }
      if (act0) b0 -= d0[0] + d0[1];
      if (act1) b1 -= d1[0] + d1[1];
    }
    bvec[lane] = b0;
    bvec[lane + 64] = b1;
    BLR_BS_STAMP(13);
    double ld = log(P[pidx(lane, lane)]) + log(P[pidx(lane + 64, lane + 64)]);
    ld = 2.0 * wave_allreduce(ld);  // logdet A
    if (lane == 0) { scr[6] = uu; scr[7] = ld; }
  }
  BLR_BS_STAMP(14);
  __syncthreads();
  BLR_BS_STAMP(15);
}

// =========================================================================================================
// the kernel
// =========================================================================================================
template <bool DIAG, bool ROWV = false, int NG = (DIAG ? kI8GroupsDiag : kI8GroupsIso)>
__global__ __launch_bounds__(kI8Threads, 2) void fused_i8_kernel(PosteriorArgs<double> a) {
  using T = double;
  using C = I8Cfg;
  using SC = SmallCfg<double, 8>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* const P = reinterpret_cast<T*>(smem);
  T* const bvec = reinterpret_cast<T*>(smem + SC::OFF_B);
  double* const scr = reinterpret_cast<double*>(smem + SC::OFF_SCR);
  int* const iscr = reinterpret_cast<int*>(smem + SC::OFF_SCR + 64);
  int* const xch = reinterpret_cast<int*>(smem + C::OFF_XCH);
  int* const flag = reinterpret_cast<int*>(smem + C::OFF_FLAG);
  double* const sctab = reinterpret_cast<double*>(smem + C::OFF_SC);
  double* const bred = reinterpret_cast<double*>(smem + C::OFF_BRED);
  double* const gdiag = reinterpret_cast<double*>(smem + C::OFF_GD);
  double* const tdtab = reinterpret_cast<double*>(smem + C::OFF_TD);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int reg = blockIdx.x;
  const int N = a.N;
  const int N32 = N & ~(C::KC - 1);  // whole 32-column k-steps go through the int8 stream; the last N % 32 columns join in fp64 at the hand-over
  constexpr int D = 128;
  const double kNaN = __longlong_as_double(0x7ff8000000000000LL);
  const BLR_GLOBAL T* X = as_global(a.X + (int64_t)reg * a.strideX);
  // diagonal noise: x / sqrt(s_n), y / sqrt(s_n) take the places of x, y (unit noise from there on); i8_noise_prep_kernel has
  // written y / sqrt(s) and 1 / sqrt(s), summed log s and flagged a variance that is not positive and finite
  const BLR_GLOBAL T* y = DIAG ? as_global(a.i8_yt + (int64_t)reg * a.i8_stride) : as_global(a.y + (int64_t)reg * a.stridey);
  const BLR_GLOBAL T* rwp = DIAG ? as_global(a.i8_rw + (int64_t)reg * a.i8_stride) : y;
  const BLR_GLOBAL T* mw = as_global(a.mw + (int64_t)reg * a.stridemw);
  const BLR_GLOBAL T* Lw = as_global(a.Lw + (int64_t)reg * a.strideLw);
  const T s_iso = DIAG ? T(1) : as_global(a.s + (int64_t)reg * a.strides)[0];
  const double rwmax = DIAG ? a.i8_rwmax[reg] : 1.0;
  const bool fac = a.prior_kind == PRIOR_UPPER_FACTOR;  // Lw is the upper factor U of the prior precision (PDMat / a carried-forward posterior): U'U joins at the hand-over
  // Lw is a dense symmetric precision (upper triangle read; the reference's own toy priors, test/test_utils.jl:6-8): it joins the finished
  // matrix at the hand-over like a factor does; its Cholesky -- only logdet Lw and the positive-definiteness check of reference :78 come
  // from it -- has been done by i8_prior_logdet_kernel before this launch (once for a prior shared by the batch)
  const bool dns = a.prior_kind == PRIOR_DENSE;
  if (blockIdx.x == 0 && tid == 0 && a.i8_handed_slice != nullptr) *a.i8_handed_slice = 0ull;  // (this slice's retry launch counts into it)
  if (blockIdx.x == 0 && tid == 0 && a.i8_call_base != nullptr) *a.i8_call_base = *a.i8_handed_tot;  // (stream order: every earlier call's retry launch has finished)
  if (DIAG && a.i8_bad[reg] != 0) {  // (uniform)  reference :79: the fp64 kernel reports the index
    if (tid == 0) a.info[reg] = kI8Retry;
    return;
  }
  // the slice before this one handed back more than a quarter of its regressors (heavy-tailed rows: bounds from the first columns do
  // not hold): streaming these twice costs more than the fp64 kernel alone
  if (a.i8_prev_n > 0 && __builtin_nontemporal_load(a.i8_prev_handed) * 4ull > (unsigned long long)a.i8_prev_n) {
    if (tid == 0) a.info[reg] = kI8Retry;
    return;
  }

  // A prior mean costs the stream nothing: with G = X X' exact, b = X (y - X'mw) / s = X y / s - (G / s) mw and
  // delta'delta / s = y'y / s - 2 mw'X y / s + mw'(G / s) mw come out of the finished matrix after the hand-over (below; G / s is
  // A off the diagonal and kept next to it on the diagonal: A_ii - Lw_i would lose the data term under a strong prior).
#ifdef BLR_I8_STAGGER  /* diagnostic builds: the first round's workgroups start BLR_I8_STAGGER x (blockIdx & 7) us apart, so that the CUs' stream and tail phases interleave */
  if (blockIdx.x < 256) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long dt = (unsigned long long)(BLR_I8_STAGGER) * 100ull * (unsigned long long)(blockIdx.x & 7);
    while (__builtin_amdgcn_s_memrealtime() - t0 < dt) __builtin_amdgcn_s_sleep(32);
  }
#endif
  int ok = 1;
  const int has_mw = __syncthreads_or(tid < D && mw[tid] != T(0));
  if (tid == 0) flag[0] = 1;

  I8Slice st;
  I8_KSTAMP_DECL;
  // one copy of the stream per wave: the tile / group table is static
  auto run = [&](auto wtag) {
    constexpr int W = decltype(wtag)::value;
    using PL = I8Plan<NG, W>;
    using Items = I8Items<NG, W>;
    i32x16 A[PL::NACC];
    i8_gram_stream<NG, W, DIAG, ROWV>(smem, X, y, rwp, rwmax, a.ldx, N32, tid, A, st, ok);
    I8_KSTAMP(4);
    // ---- hand-over: validity, digit row sums, b partials, row scales (all through the exchange area / the dead digit area)
    // (Inf / NaN in X fail the capacity test and are seen by the repair; in y they show up in b = sum x y)
    if (!ok || !(__builtin_fabs(st.b) < __longlong_as_double(0x7ff0000000000000LL))) flag[0] = 0;  // (benign race: everybody writes the same value)
    xch[(tid >> 7) * 128 + (tid & 127)] = st.sq3;
    // a diagonal prior joins the diagonal after the conversion (a factor or a dense prior after the prior-mean terms): in flight meanwhile
    double lwd = 0.0;
    if (tid < D && !(fac || dns)) lwd = (double)Lw[tid];
    __syncthreads();  // ring and digit buffers are dead from here on
    bred[(tid >> 7) * 128 + (tid & 127)] = st.b;
    if (tid < D) {
      // 2^(e_i - 47) from the magic constant: C = 1.5 2^(e + 5) (+ the balancing offset)  ->  exponent field - 52
      const int ef = (int)(((unsigned)__double2hiint(st.C) >> 20) & 0x7ffu);
      sctab[tid] = __hiloint2double((ef - 52) << 20, 0);
      // The products that are dropped (s + t >= NG) are sums over the columns of balanced digits: zero-mean (header) -- except the pair
      // (3, 3) of the 6-group plan on the diagonal, a sum of squares (5461 N for uniform digits: 8e-14 of G_ii).  sum_n b_3^2 is exact
      // from one more v_dot4 per quad (I8Slice::sq3): TD = sum b_3^2 2^32 joins G_ii.
      const long long s33 = ((long long)xch[tid] + xch[128 + tid]) + ((long long)xch[256 + tid] + xch[384 + tid]);
      tdtab[tid] = NG == 6 ? (double)s33 * __hiloint2double((1023 + 80 - 48) << 20, 0) : 0.0;
    }
    double qsum = 0.0;
    if ((tid & 127) == 0) qsum = st.q;  // the four threads of row 0 hold the four column octets' shares
    __syncthreads();
    if ((tid & 127) == 0) reinterpret_cast<double*>(xch)[tid >> 7] = qsum;
    const int valid = flag[0];
    I8_KSTAMP(8);
    // ---- accumulators -> fp64 -> packed lower triangle of A = Lw + G / sigma^2 (diagonal prior), phase-0 items store, the others add
    const T winv = T(1) / s_iso;
    auto convert = [&](auto ittag) {
      constexpr int IT = decltype(ittag)::value;
      constexpr I8Item it = Items::it[IT];
      constexpr int base = PL::acc_base(IT);
      constexpr bool symt = i8_sym_tile<NG>(it.I, it.K);
      // (a symmetric diagonal tile's item has Q slots -- pairs s < t, whose transposes belong to the same entries -- unless it holds R slots only)
      constexpr bool hasq = symt && i8_slot_kind<NG>(it.I, it.K, it.q0) == 1;
      const int jl = lane & 31, j = 32 * it.K + jl;
      const double scj = sctab[j] * winv;
      // Mirror through a scratch tile of the wave's own (32 x 32 doubles, row stride 33: written with consecutive lanes = consecutive
      // words, read transposed without bank conflicts; one tile per diagonal block, and the phases never give two waves the same block):
      // entry (i, j) of the result takes Q(i, j) + Q(j, i) -- on the diagonal that is 2 Q(i, i) by itself.  (First version: lanes with
      // i < j added their Q to P[(j, i)] in a second sweep -- triangular-number strides, 4-way bank conflicts, two dependent LDS round
      // trips per entry: 7 k cycles per item against 2.7 k for an off-diagonal one.)
      double* const mir = reinterpret_cast<double*>(smem + C::OFF_TAIL) + it.I * (32 * 33);
      double glo[hasq ? 16 : 1];
      if constexpr (hasq) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int il = 8 * (v >> 2) + 4 * (lane >> 5) + (v & 3), i = 32 * it.I + il;
          double lo2[2] = {0.0, 0.0}, qs2[2] = {0.0, 0.0};
#pragma unroll
          for (int k = 10; k >= 0; --k) {
#pragma unroll
            for (int q = it.q0; q <= it.q1; ++q)
              if (i8_slot_k<NG>(it.I, it.K, q) == k) {
                const double term = (double)A[base + q - it.q0][v];
                const double sc2 = __hiloint2double((1023 + 80 - 8 * k) << 20, 0);
                if (i8_slot_kind<NG>(it.I, it.K, q) == 1) qs2[q & 1] = __builtin_fma(term, sc2, qs2[q & 1]);
                else lo2[q & 1] = __builtin_fma(term, sc2, lo2[q & 1]);
              }
          }
          const double sc = sctab[i] * scj;
          const double gqv = (qs2[0] + qs2[1]) * sc;
          glo[v] = (lo2[0] + lo2[1]) * sc + gqv;
          mir[il * 33 + jl] = gqv;
          if ((v & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int v0 = 0; v0 < 16; v0 += 8) {
          double tr[8], cur[8];
#pragma unroll
          for (int v = v0; v < v0 + 8; ++v) {
            const int il = 8 * (v >> 2) + 4 * (lane >> 5) + (v & 3), i = 32 * it.I + il;
            tr[v - v0] = mir[jl * 33 + il];
            cur[v - v0] = (!it.first && i >= j) ? P[pidx(i, j)] : 0.0;
          }
#pragma unroll
          for (int v = v0; v < v0 + 8; ++v) {
            const int i = 32 * it.I + 8 * (v >> 2) + 4 * (lane >> 5) + (v & 3);
            if (i >= j) P[pidx(i, j)] = cur[v - v0] + glo[v] + tr[v - v0];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int il = 8 * (v >> 2) + 4 * (lane >> 5) + (v & 3), i = 32 * it.I + il;
          // sum over the slots of (exact integer) x 2^(80 - 8 k): two interleaved partial sums -- a dependent fp64 operation waits for its
          // predecessor
          double lo2[2] = {0.0, 0.0};
#pragma unroll
          for (int k = 10; k >= 0; --k) {  // smallest scale first
#pragma unroll
            for (int q = it.q0; q <= it.q1; ++q)
              if (i8_slot_k<NG>(it.I, it.K, q) == k) {
                const double term = (double)A[base + q - it.q0][v];
                const double sc2 = __hiloint2double((1023 + 80 - 8 * k) << 20, 0);
                lo2[q & 1] = __builtin_fma(term, sc2, lo2[q & 1]);
              }
          }
          const double sc = sctab[i] * scj;
          if (it.I != it.K || i >= j) {  // (off-diagonal tiles lie below the diagonal as a whole)
            const double g = (lo2[0] + lo2[1]) * sc;
            if constexpr (it.first) P[pidx(i, j)] = g;
            else P[pidx(i, j)] += g;
          }
          if ((v & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // four entries at a time (their LDS round trips overlap; more would spill)
        }
      }
    };
    auto convert_all = [&](auto self, auto ittag, auto phase_tag) -> void {
      constexpr int IT = decltype(ittag)::value;
      if constexpr (IT < PL::NI) {
        if constexpr (Items::it[IT].phase == decltype(phase_tag)::value) convert(ittag);
        self(self, std::integral_constant<int, IT + 1>{}, phase_tag);
      }
    };
    if (valid) convert_all(convert_all, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    I8_KSTAMP(9);
    __syncthreads();
    if (valid) convert_all(convert_all, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
    I8_KSTAMP(10);
    __syncthreads();
    if (valid) convert_all(convert_all, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
    I8_KSTAMP(11);
    __syncthreads();
    I8_KSTAMP(16);
    // the diagonal: TD_i (the dropped pair (3, 3), a sum of squares); diag(G) / s to `gdiag` BEFORE the diagonal prior joins (the data term
    // alone: A_ii - Lw_i would lose it under a strong prior)
    if (valid && tid < D) {
      const double sci = sctab[tid];
      const double e = __builtin_fma(tdtab[tid], sci * sci * winv, P[pidx(tid, tid)]);
      gdiag[tid] = e;
      P[pidx(tid, tid)] = e + lwd;
    }
    I8_KSTAMP(17);
  };
  switch (wave) {
    case 0: run(std::integral_constant<int, 0>{}); break;
    case 1: run(std::integral_constant<int, 1>{}); break;
    case 2: run(std::integral_constant<int, 2>{}); break;
    case 3: run(std::integral_constant<int, 3>{}); break;
    case 4: run(std::integral_constant<int, 4>{}); break;
    case 5: run(std::integral_constant<int, 5>{}); break;
    case 6: run(std::integral_constant<int, 6>{}); break;
    default: run(std::integral_constant<int, 7>{}); break;
  }
  // b = X y / sigma^2 (fixed order over the four octet threads of a row), q = y'y / sigma^2
  const T winv = T(1) / s_iso;
  double bsum = 0.0;
  if (tid < D) bsum = ((bred[tid] + bred[128 + tid]) + (bred[256 + tid] + bred[384 + tid])) * winv;
  const double* qx = reinterpret_cast<const double*>(xch);
  double quad = ((qx[0] + qx[1]) + (qx[2] + qx[3])) * winv;
  const int valid = flag[0];
  __syncthreads();  // P complete; the tables in the digit area have been read
  // ---- repair: blocks in which an entry outgrew its row's capacity.  Its digits were those of c = x - m 2^(e_r + 1) (the integer wraps; beyond
  // 16 capacities they are whatever the magic addition left in the low mantissa bits -- still a definite number c), so the stream added
  // c c' for that column where x x' was meant.  Per marked block: read its 32 columns again, recompute c exactly as the slicing did, and add
  // x x' - c c' = d v' + v d' + d d' (d = x - c, v = the column as the matrix has it) entry by entry in fp64, in a fixed order.  b and y'y
  // were formed from the fp64 values and need nothing.  Inf / NaN, or more than kI8MaxRepair marked blocks: the fp64 kernel.
  I8_KSTAMP(5);
  if (valid) {  // (uniform)
    const unsigned char* const oflag = reinterpret_cast<const unsigned char*>(smem + C::OFF_OFLAG);
    unsigned long long* const omask = reinterpret_cast<unsigned long long*>(smem + C::OFF_OMASK);
    unsigned* const cmask = reinterpret_cast<unsigned*>(smem + C::OFF_OMASK + 64);      // [32 columns][4 words of 32 rows]
    unsigned* const oscr = reinterpret_cast<unsigned*>(smem + C::OFF_OMASK + 64 + 512);  // [0] columns with an entry to correct; [2..3] d
    const int nk = N32 / C::KC;
    const int nmarked = __syncthreads_count(tid < nk && oflag[tid] != 0);
    // (at most one marked block in 16, and never more than kI8MaxRepair: beyond, the rows' scales do not fit the data -- heavy tails, a
    //  feature that wakes up late -- and reading the blocks again costs more than the fp64 kernel)
    if (nmarked > (nk / 16 > kI8MaxRepair ? kI8MaxRepair : (nk / 16 < 2 ? 2 : nk / 16))) {
      if (tid == 0) a.info[reg] = kI8Retry;
      return;
    }
    if (nmarked > 0) {  // (uniform; rare)
      if (wave == 0) {
#pragma unroll
        for (int q8 = 0; q8 < 8; ++q8) {
          const int blk = 64 * q8 + lane;
          const unsigned long long m = __builtin_amdgcn_ballot_w64(blk < nk && oflag[blk] != 0);
          if (lane == 0) omask[q8] = m;
        }
      }
      __syncthreads();
      double* const vt = reinterpret_cast<double*>(smem + C::OFF_TAIL);  // the block as the matrix has it: [32 columns][128 rows]
      const int r = tid & 127, cq = tid >> 7;
      // this row's magic constant, capacity test and grid, from 2^(e_r - 47) (sctab)
      const int ef = (int)(((unsigned)__double2hiint(sctab[r]) >> 20) & 0x7ffu);  // biased e_r - 47
      const double Cr = i8_magic(ef + 47);
      const unsigned hlo = i8_hlo(ef + 47);
      const double grid = sctab[r];
      for (int q8 = 0; q8 < 8; ++q8) {
        unsigned long long bm = omask[q8];
        bm = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(bm >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)bm);
        while (bm != 0ull) {
          const int jb = 64 * q8 + __builtin_ctzll(bm);
          bm &= bm - 1ull;
          if (tid < 128) cmask[tid] = 0u;
          if (tid == 0) oscr[0] = 0u;
          __syncthreads();
          double dreg[8];
          int bad = 0;
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            const int n = C::KC * jb + 8 * cq + c;
            double xv = ROWV ? X[(int64_t)r * a.ldx + n] : X[(int64_t)n * a.ldx + r];
            if constexpr (DIAG) xv = __dmul_rn(xv, rwp[n]);
            const unsigned ax = (unsigned)__double2hiint(xv) & 0x7fffffffu;
            if (ax >= 0x7ff00000u) bad = 1;
            double cur = xv;
            dreg[c] = 0.0;
            const double t = __dadd_rn(xv, Cr);
            if (!i8_fits((unsigned)__double2hiint(t), hlo)) {  // the test of the slicing, the arithmetic of the slicing
              const long long q48 = ((long long)(short)((unsigned)__double2hiint(t) & 0xffffu) << 32) | (long long)(unsigned)__double2loint(t);
              cur = (double)(q48 - kI8Balance) * grid;  // (the digits of the wrapped Q' stand for Q' - 0x8080808080)
              dreg[c] = xv - cur;
              atomicOr(&cmask[(8 * cq + c) * 4 + (r >> 5)], 1u << (r & 31));
              atomicOr(&oscr[0], 1u << (8 * cq + c));
            }
            vt[(8 * cq + c) * 128 + r] = cur;
          }
          if (__syncthreads_or(bad)) {  // Inf / NaN: the fp64 kernel reports what the reference would
            if (tid == 0) a.info[reg] = kI8Retry;
            return;
          }
          unsigned cols = (unsigned)__builtin_amdgcn_readfirstlane((int)oscr[0]);
          while (cols != 0u) {
            const int cl = __builtin_ctz(cols);
            cols &= cols - 1u;
            for (int wq = 0; wq < 4; ++wq) {
              unsigned rm = (unsigned)__builtin_amdgcn_readfirstlane((int)cmask[cl * 4 + wq]);
              while (rm != 0u) {
                const int rr = 32 * wq + __builtin_ctz(rm);
                rm &= rm - 1u;
                if (tid == rr + 128 * (cl >> 3)) {  // the thread that holds this entry's d publishes it
                  const int c8 = cl & 7;
                  double dv = dreg[0];
#pragma unroll
                  for (int c = 1; c < 8; ++c) dv = (c8 == c) ? dreg[c] : dv;
                  reinterpret_cast<double*>(oscr)[1] = dv;
                }
                __syncthreads();
                const double dv = reinterpret_cast<const double*>(oscr)[1];
                if (tid < 128) {
                  const double vj = vt[cl * 128 + tid];
                  if (tid != rr) {
                    P[tid > rr ? pidx(tid, rr) : pidx(rr, tid)] += winv * dv * vj;
                  } else {
                    const double g = winv * (2.0 * dv * vj + dv * dv);
                    P[pidx(rr, rr)] += g;
                    gdiag[rr] += g;
                    vt[cl * 128 + rr] = vj + dv;  // the column now holds the true entry
                  }
                }
                __syncthreads();
              }
            }
          }
          __syncthreads();
        }
      }
    }
  }
  I8_KSTAMP(3);
  if (N32 < N && valid) {  // (uniform)  the last r = N % 32 columns: a rank-r term in fp64, all eight waves
    const int r = N - N32;
    double* const tb = reinterpret_cast<double*>(smem + C::OFF_TAIL);  // [r][128], then y[r]
    int bad = 0;
    for (int e = tid; e < r * D; e += kI8Threads) {
      double v = ROWV ? X[(int64_t)(e & 127) * a.ldx + N32 + (e >> 7)] : X[(int64_t)(N32 + (e >> 7)) * a.ldx + (e & 127)];
      if constexpr (DIAG) v *= rwp[N32 + (e >> 7)];
      tb[e] = v;
      if (!(fabs(v) < __longlong_as_double(0x7ff0000000000000LL))) bad = 1;  // Inf / NaN: as in the stream, the fp64 kernel reports it
    }
    if (tid < r) tb[31 * D + tid] = y[N32 + tid];
    if (__syncthreads_or(bad)) {
      if (tid == 0) a.info[reg] = kI8Retry;
      return;
    }
    for (int e = tid; e < SC::PACKED; e += kI8Threads) {
      int i = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);  // packed index -> (i, j), j <= i
      while ((i + 1) * (i + 2) / 2 <= e) ++i;
      while (i * (i + 1) / 2 > e) --i;
      const int j = e - i * (i + 1) / 2;
      double acc = 0.0;
      for (int c = 0; c < r; ++c) acc = __builtin_fma(tb[c * D + i], tb[c * D + j], acc);
      P[e] += acc * winv;
      if (i == j) gdiag[i] += acc * winv;
    }
    double bt = 0.0, qt = 0.0;
    for (int c = 0; c < r; ++c) {
      const double yc = tb[31 * D + c];
      if (tid < D) bt = __builtin_fma(tb[c * D + tid], yc, bt);
      qt = __builtin_fma(yc, yc, qt);
    }
    bsum += bt * winv;
    quad += qt * winv;
    __syncthreads();
  }
  I8_KSTAMP(5);
  if (wave >= 4) return;  // the phases below are four-wave code (their barriers count the surviving waves only)
  if (!valid) {
    if (tid == 0) a.info[reg] = kI8Retry;
    return;
  }
  double quad_mw = quad;
  if (has_mw) {  // (uniform)  prior mean != 0: fold it into the right-hand side and the quadratic form
    T* const mwl = reinterpret_cast<T*>(smem + SC::OFF_MW);
    if (tid < D) mwl[tid] = mw[tid];
    __syncthreads();
    double gm = 0.0, mi = 0.0;
    if (tid < D) {
      mi = (double)mwl[tid];
      double am = 0.0;
#pragma unroll 4
      for (int j = 0; j < D; ++j) {
        const double aij = (j < tid) ? P[pidx(tid, j)] : ((j > tid) ? P[pidx(j, tid)] : gdiag[tid]);  // off the diagonal A = G / s
        am += aij * (double)mwl[j];
      }
      gm = am;  // (G mw)_i / s
    }
    const double s1 = block_allreduce(mi * bsum, scr, tid);
    const double s2 = block_allreduce(mi * gm, scr, tid);
    quad_mw = quad - 2.0 * s1 + s2;
    // Both are differences of y'y / s-sized numbers.  When the prior mean already explains the data (a carried-forward posterior
    // conditioned on more of the same stream) they cancel: the error of delta'delta / s is eps y'y / s, not eps delta'delta / s
    // (reference :82 forms delta = y - X'mw first, and so does the fp64 kernel).  Three digits of cancellation are the most this
    // route keeps for itself -- inside the 1e-10 the header promises for the evidence; beyond, the fp64 kernel redoes the regressor.
    const double bn_old = block_allreduce(bsum * bsum, scr, tid);
    bsum -= gm;
    const double bn_new = block_allreduce(bsum * bsum, scr, tid);
    if (!(quad_mw > 1e-3 * quad) || !(bn_new >= 1e-6 * bn_old)) {  // (uniform)
      if (tid == 0) a.info[reg] = kI8Retry;
      return;
    }
  }
  if (tid < D) bvec[tid] = (T)bsum;
  // prior: SPD check + logdet (reference :78; a factor: its diagonal, logdet = 2 sum log U_kk), noise variance (reference :79)
  int info = 0;
  double logdet_Lw = 0.0;
  if (dns) {  // (uniform)
    info = a.i8_prior_info[(int64_t)reg * a.i8_prior_stride];
    logdet_Lw = a.i8_prior_logdet[(int64_t)reg * a.i8_prior_stride];
    if (info == 0) {
      // A += Lw, after the prior-mean terms have been taken from the pure data matrix: rows over the four waves, a row's entries
      // j <= i over the lanes (element (j, i) of the caller's upper triangle: consecutive lanes, consecutive words), four rows in flight
      const int w4 = tid >> 6, l = tid & 63;
#pragma unroll 1
      for (int i0 = 4 * w4; i0 < D; i0 += 16) {
        double v0[4], v1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u;
          v0[u] = (l <= i) ? (double)Lw[(int64_t)i * a.ldl + l] : 0.0;
          v1[u] = (l + 64 <= i) ? (double)Lw[(int64_t)i * a.ldl + l + 64] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u;
          if (l <= i) P[pidx(i, l)] += v0[u];
          if (l + 64 <= i) P[pidx(i, l + 64)] += v1[u];
        }
      }
      __syncthreads();
    }
  } else {
    double v = 0.0;
    int bad = 0x7fffffff;
    if (tid < D) {
      const T dv = fac ? Lw[(int64_t)tid * a.ldl + tid] : Lw[tid];
      if (dv > T(0)) v = log((double)dv);
      else bad = tid + 1;
    }
    bad = block_min_int(bad, iscr, tid);
    if (bad != 0x7fffffff) info = bad;
    logdet_Lw = block_allreduce(v, scr, tid) * (fac ? 2.0 : 1.0);
  }
  if (fac && info == 0) {  // (uniform)
    // A = U'U + G / s: the prior's part in fp64, after the prior-mean terms have been taken from the pure data matrix.  4 x 4 blocks
    // of the lower triangle (528 of them on 256 threads), U staged sixteen rows at a time (zero above its diagonal, so that a
    // term with k > min(i, j) vanishes), the next chunk's loads in flight while this one is multiplied.  ~7 % of an update.
    double* const ub = reinterpret_cast<double*>(smem + C::OFF_TAIL);
    constexpr int LU = 130;  // row stride of the staged chunk (16 x 128, padded)
    int bi[3], bj[3];
    double acc[3][16];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int blk = tid + kThreads * q;
      int i = -1, j = 0;
      if (blk < 528) {
        i = (int)((sqrtf(8.0f * (float)blk + 1.0f) - 1.0f) * 0.5f);
        while ((i + 1) * (i + 2) / 2 <= blk) ++i;
        while (i * (i + 1) / 2 > blk) --i;
        j = blk - i * (i + 1) / 2;
      }
      bi[q] = i; bj[q] = j;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[q][e] = 0.0;
    }
    const int skk = tid & 15, sc0 = tid >> 4;  // staging: rows k0 + skk of columns sc0 + 16 u (16 consecutive doubles of a column per 16 threads)
    double pre[8];
    auto fetch = [&](int k0) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = sc0 + 16 * u, k = k0 + skk;
        pre[u] = (k <= c) ? (double)Lw[(int64_t)c * a.ldl + k] : 0.0;
      }
    };
    fetch(0);
#pragma unroll 1
    for (int k0 = 0; k0 < D; k0 += 16) {
      __syncthreads();  // the previous chunk has been consumed
#pragma unroll
      for (int u = 0; u < 8; ++u) ub[skk * LU + sc0 + 16 * u] = pre[u];
      if (k0 + 16 < D) fetch(k0 + 16);
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        if (bi[q] < 0 || k0 > 4 * bj[q] + 3) continue;  // (rows k > j of column j are zero)
        const double* pa = ub + 4 * bi[q];
        const double* pb = ub + 4 * bj[q];
#pragma unroll 4
        for (int kk = 0; kk < 16; ++kk) {
          double av[4], bv[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) { av[e] = pa[kk * LU + e]; bv[e] = pb[kk * LU + e]; }
#pragma unroll
          for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) acc[q][4 * rr + cc] = __builtin_fma(av[rr], bv[cc], acc[q][4 * rr + cc]);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      if (bi[q] < 0) continue;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const int i = 4 * bi[q] + rr, j = 4 * bj[q] + cc;
          if (j <= i) P[pidx(i, j)] += acc[q][4 * rr + cc];
        }
    }
  }
  if (info == 0 && !(s_iso > T(0))) info = 1;
  if (info != 0) {
    if (tid == 0) {
      a.info[reg] = info;
      if (a.logpdf) a.logpdf[reg] = kNaN;
    }
    return;
  }
  const double logdet_Sy = DIAG ? a.i8_logdet[reg] : (double)N * log((double)s_iso);
  __syncthreads();
  if (a.Lw_post) {  // posterior precision Lw' = A, full symmetric (:92)
    T* out = a.Lw_post + (int64_t)reg * a.strideLp;
    for (int c = tid >> 6; c < D; c += kWaves)
      for (int rr = tid & 63; rr < D; rr += kWave) out[(int64_t)c * a.ldlp + rr] = (rr >= c) ? P[pidx(rr, c)] : P[pidx(c, rr)];
  }
  info = phase_chol<T, 8>(smem, D, 1);  // :86; T = L' is chol(Lw + G).U (:67)
  I8_KSTAMP(6);
  if (info != 0) {
    if (tid == 0) {
      a.info[reg] = info;
      if (a.logpdf) a.logpdf[reg] = kNaN;
    }
    return;
  }
  i8_backsolve_blocked(smem, a.T_post ? a.T_post + (int64_t)reg * a.strideT : (T*)nullptr, a.ldt);
  if (a.mw_post && tid < D) a.mw_post[(int64_t)reg * a.stride_mwpost + tid] = mw[tid] + bvec[tid];  // :68
  if (tid == 0) {
    a.info[reg] = 0;
    if (a.logpdf) {
      const double LOG2PI = 1.8378770664093454835606594728112;
      a.logpdf[reg] = -0.5 * ((double)N * LOG2PI + logdet_Sy + quad_mw + scr[7] - logdet_Lw - scr[6]);  // :84 + :57
    }
  }
  I8_KSTAMP(7);
  I8_CLKSTAMP;
}

// ---- dense prior: what the int8 route needs from Lw, once per call ------------------------------------------------------------------
// One 256-thread workgroup per prior (one prior in all when the batch shares it): the blocked Cholesky of fused_small_kernel's prior
// phase on the packed triangle -> logdet Lw (fixed order) and the LAPACK-style index of a failing leading minor (reference :78).
static __global__ __launch_bounds__(kThreads, 2) void i8_prior_logdet_kernel(const double* __restrict__ Lw, int64_t ldl, int64_t strideLw, int nprior,
                                                                          double* __restrict__ logdet, int32_t* __restrict__ info_out) {
  using SC = SmallCfg<double, 8>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* const P = reinterpret_cast<double*>(smem);
  double* const scr = reinterpret_cast<double*>(smem + SC::OFF_SCR);
  const int tid = threadIdx.x;
  constexpr int D = 128;
  for (int p = blockIdx.x; p < nprior; p += gridDim.x) {
    const double* A = Lw + (int64_t)p * strideLw;
    __syncthreads();
    for (int i = tid >> 6; i < D; i += kWaves)
      for (int k = tid & 63; k <= i; k += kWave) P[pidx(i, k)] = A[(int64_t)i * ldl + k];  // upper entry (k, i) -> lower (i, k)
    __syncthreads();
    const int info = phase_chol<double, 8>(smem, D, 0);
    const double v = (info == 0 && tid < D) ? log(P[pidx(tid, tid)]) : 0.0;
    const double ld = 2.0 * block_allreduce(v, scr, tid);
    if (tid == 0) { logdet[p] = ld; info_out[p] = info; }
  }
}

// ---- diagonal noise: what the int8 route needs from s, once per call --------------------------------------------------------------
// One workgroup per regressor: yt_n = y_n / sqrt(s_n), rw_n = 1 / sqrt(s_n), logdet = sum_n log s_n (fixed order), bad = some s_n is
// not positive and finite (reference :79 cholesky(Sigma_y) throws: the fp64 kernel, which the regressor is handed to, reports it).
static __global__ __launch_bounds__(kThreads) void i8_noise_prep_kernel(const double* __restrict__ s, int64_t strides, const double* __restrict__ y,
                                                                 int64_t stridey, int N, double* __restrict__ yt, double* __restrict__ rw,
                                                                 int64_t stride, double* __restrict__ logdet, int32_t* __restrict__ bad,
                                                                 double* __restrict__ rwmax) {
  __shared__ double scr[8];
  __shared__ int iscr[8];
  const int reg = blockIdx.x, tid = threadIdx.x;
  s += (int64_t)reg * strides; y += (int64_t)reg * stridey; yt += (int64_t)reg * stride; rw += (int64_t)reg * stride;
  double ld = 0.0, rmax = 0.0;
  int b = 0;
  for (int n = tid; n < N; n += kThreads) {
    const double sn = s[n];
    const bool okn = sn > 0.0 && sn < __longlong_as_double(0x7ff0000000000000LL);
    if (!okn) b = 1;
    const double r = fast_rsqrt(okn ? sn : 1.0);
    rw[n] = r;
    rmax = r > rmax ? r : rmax;
    yt[n] = y[n] * r;
    ld += log(okn ? sn : 1.0);
  }
  ld = block_allreduce(ld, scr, tid);
  b = -block_min_int(-b, iscr, tid);
  // (max over the block through the integer minimum: positive doubles order like their bit patterns)
  const int hi = -block_min_int(-__double2hiint(rmax), iscr, tid);
  if (tid == 0) { logdet[reg] = ld; bad[reg] = b; rwmax[reg] = __hiloint2double(hi, -1); }
}

// The four instantiations of fused_i8_kernel are most of the library's build time, so they live in translation units of their own
// (blr_i8_kernels.hip, compiled once per noise kind, next to blr_abi.hip: make -j3).  The host side reaches them through these:
// the kernel's address (hipFuncSetAttribute) and one launch of `grid` workgroups, RowVecs or ColVecs inputs.
const void* i8_kernel_ptr_iso(bool rowv);
const void* i8_kernel_ptr_diag(bool rowv);
void i8_kernel_launch_iso(bool rowv, unsigned grid, hipStream_t stream, const PosteriorArgs<double>& a);
void i8_kernel_launch_diag(bool rowv, unsigned grid, hipStream_t stream, const PosteriorArgs<double>& a);
// the OTHER plan of each noise kind (handle option I8_GROUPS): seven groups under isotropic noise, six under diagonal noise
const void* i8_kernel_ptr_alt_iso(bool rowv);
const void* i8_kernel_ptr_alt_diag(bool rowv);
void i8_kernel_launch_alt_iso(bool rowv, unsigned grid, hipStream_t stream, const PosteriorArgs<double>& a);
void i8_kernel_launch_alt_diag(bool rowv, unsigned grid, hipStream_t stream, const PosteriorArgs<double>& a);
inline const void* i8_kernel_ptr_alt(bool diag, bool rowv) { return diag ? i8_kernel_ptr_alt_diag(rowv) : i8_kernel_ptr_alt_iso(rowv); }
inline void i8_kernel_launch_alt(bool diag, bool rowv, unsigned grid, hipStream_t stream, const PosteriorArgs<double>& a) {
  if (diag) i8_kernel_launch_alt_diag(rowv, grid, stream, a);
  else i8_kernel_launch_alt_iso(rowv, grid, stream, a);
}

}  // namespace blr
