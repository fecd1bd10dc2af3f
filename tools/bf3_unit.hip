// Unit check for the bf16 x 3 form of an fp32 product on gfx950: one wave, C(32 x 32) = A(32 x 16) B(32 x 16)' from operands held as
// v_mfma_f32_16x16x4_f32 fragments (lane (r16, q) holds row 16 I + r16, column 4 ks + q), via v_permlane16_swap_b32, an exact three-way
// split into bf16 and six v_mfma_f32_32x32x16_bf16.  Prints the error against a double product next to that of an fp32 product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 bf3_unit.hip -o bf3_unit
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

struct Planes { u4 h, m, l; };  // eight bf16 k slots per lane and plane
// x[ks], y[ks]: the two registers of a row-block pair after the swap (k = 4 ks + 2 h and 4 ks + 2 h + 1)
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {  // v_cvt_pk_bf16_f32: round to nearest even, lo -> bits 15:0
  bf2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ Planes split_pack(const float (&x)[4], const float (&y)[4]) {
  Planes p;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
#ifdef BF3_TRUNCATE
    const unsigned xb = __float_as_uint(x[ks]), yb = __float_as_uint(y[ks]);
    const float xr = x[ks] - __uint_as_float(xb & 0xffff0000u), yr = y[ks] - __uint_as_float(yb & 0xffff0000u);
    const unsigned xrb = __float_as_uint(xr), yrb = __float_as_uint(yr);
    const float xl = xr - __uint_as_float(xrb & 0xffff0000u), yl = yr - __uint_as_float(yrb & 0xffff0000u);
    p.h[ks] = __builtin_amdgcn_perm(yb, xb, 0x07060302u);   // (y.hi16 << 16) | x.hi16: slot 2 ks = x, slot 2 ks + 1 = y
    p.m[ks] = __builtin_amdgcn_perm(yrb, xrb, 0x07060302u);
    p.l[ks] = __builtin_amdgcn_perm(__float_as_uint(yl), __float_as_uint(xl), 0x07060302u);
#else
    // round to nearest at every level: the residuals are signed and zero-mean, so the three dropped products do not add up coherently
    const unsigned h = pk_bf16(x[ks], y[ks]);
    const float xr = x[ks] - __uint_as_float(h << 16), yr = y[ks] - __uint_as_float(h & 0xffff0000u);
    const unsigned m = pk_bf16(xr, yr);
    const float xl = xr - __uint_as_float(m << 16), yl = yr - __uint_as_float(m & 0xffff0000u);
    p.h[ks] = h; p.m[ks] = m; p.l[ks] = pk_bf16(xl, yl);
#endif
  }
  return p;
}
__device__ __forceinline__ bf8 as_bf8(u4 v) { return __builtin_bit_cast(bf8, v); }

__global__ __launch_bounds__(64) void k(const float* A, const float* B, float* C) {
  const int lane = threadIdx.x, r16 = lane & 15, q = lane >> 4;
  float fa[2][4], fb[2][4];  // fragments [row block][k-step]
  for (int I = 0; I < 2; ++I) for (int ks = 0; ks < 4; ++ks) { fa[I][ks] = A[(16 * I + r16) * 16 + 4 * ks + q]; fb[I][ks] = B[(16 * I + r16) * 16 + 4 * ks + q]; }
  float ax[4], ay[4], bx[4], by[4];
  for (int ks = 0; ks < 4; ++ks) {
    auto sa = __builtin_amdgcn_permlane16_swap(__float_as_uint(fa[0][ks]), __float_as_uint(fa[1][ks]), false, false);
    ax[ks] = __uint_as_float(sa[0]); ay[ks] = __uint_as_float(sa[1]);
    auto sb = __builtin_amdgcn_permlane16_swap(__float_as_uint(fb[0][ks]), __float_as_uint(fb[1][ks]), false, false);
    bx[ks] = __uint_as_float(sb[0]); by[ks] = __uint_as_float(sb[1]);
  }
  const Planes pa = split_pack(ax, ay), pb = split_pack(bx, by);
  f16v acc;
  for (int v = 0; v < 16; ++v) acc[v] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(pa.l), as_bf8(pb.h), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(pa.h), as_bf8(pb.l), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(pa.m), as_bf8(pb.m), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(pa.m), as_bf8(pb.h), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(pa.h), as_bf8(pb.m), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(pa.h), as_bf8(pb.h), acc, 0, 0, 0);
  // C layout of the 32 x 32 form: column lane & 31, rows 8 (v / 4) + 4 (lane / 32) + v % 4
  for (int v = 0; v < 16; ++v) C[(8 * (v >> 2) + 4 * (lane >> 5) + (v & 3)) * 32 + (lane & 31)] = acc[v];
}

int main() {
  std::mt19937_64 g(7);
  std::normal_distribution<double> nd;
  std::vector<float> A(32 * 16), B(32 * 16), C(32 * 32);
  for (auto& v : A) v = (float)nd(g);
  for (auto& v : B) v = (float)(nd(g) * std::exp(2.0 * nd(g)));
  float *dA, *dB, *dC;
  CK(hipMalloc((void**)&dA, A.size() * 4)); CK(hipMalloc((void**)&dB, B.size() * 4)); CK(hipMalloc((void**)&dC, C.size() * 4));
  CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  k<<<1, 64>>>(dA, dB, dC);
  CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
  double e3 = 0, e32 = 0, scale = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    double ref = 0, mag = 0; float f = 0;
    for (int kk = 0; kk < 16; ++kk) { ref += (double)A[i * 16 + kk] * B[j * 16 + kk]; mag += std::fabs((double)A[i * 16 + kk] * B[j * 16 + kk]); f = std::fmaf(A[i * 16 + kk], B[j * 16 + kk], f); }
    e3 = std::fmax(e3, std::fabs(C[i * 32 + j] - ref) / mag); e32 = std::fmax(e32, std::fabs((double)f - ref) / mag);
    scale = std::fmax(scale, mag);
  }
  printf("bf16 x 3 (six products): max |C - ref| / sum|a b| = %.3e   fp32 fma chain: %.3e   (2^-24 = 5.96e-08)\n", e3, e32);
  // the coherent case: B = A with positive entries -- the MEAN signed error of the diagonal (a truncating split is biased there)
  for (auto& v : A) v = (float)(0.5 + std::fabs(nd(g)));
  CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, A.data(), A.size() * 4, hipMemcpyHostToDevice));
  k<<<1, 64>>>(dA, dB, dC);
  CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
  double bias = 0;
  for (int i = 0; i < 32; ++i) { double ref = 0; for (int kk = 0; kk < 16; ++kk) ref += (double)A[i * 16 + kk] * A[i * 16 + kk]; bias += (C[i * 32 + i] - ref) / ref / 32; }
  printf("  positive operands, diagonal of A A': mean signed relative error %.3e\n", bias);
  return 0;
}
