// How accurate are the hardware seeds v_rcp_f64 / v_rsq_f64 on gfx950, and what do one / two Newton steps leave?  (phase_chol's pivot chain
// pays ~40 cycles per dependent fp64 operation: a Newton step less is two of the ~9 operations per column.)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__global__ void k(const double* x, double* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = x[i];
  double r0 = __builtin_amdgcn_rcp(v);
  double r1 = __builtin_fma(__builtin_fma(-v, r0, 1.0), r0, r0);
  double r2 = __builtin_fma(__builtin_fma(-v, r1, 1.0), r1, r1);
  double y0 = __builtin_amdgcn_rsq(v);
  double h = 0.5 * v * y0;
  double y1 = __builtin_fma(__builtin_fma(-h, y0, 0.5), y0, y0);
  h = 0.5 * v * y1;
  double y2 = __builtin_fma(__builtin_fma(-h, y1, 0.5), y1, y1);
  out[6 * i + 0] = r0; out[6 * i + 1] = r1; out[6 * i + 2] = r2; out[6 * i + 3] = y0; out[6 * i + 4] = y1; out[6 * i + 5] = y2;
}
int main() {
  const int n = 1 << 20;
  double *hx = new double[n], *ho = new double[6 * n], *dx, *dout;
  unsigned long long st = 88172645463325252ULL;
  for (int i = 0; i < n; ++i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; hx[i] = std::ldexp(1.0 + (double)(st >> 11) / 9007199254740992.0, (int)(st % 41) - 20); }
  hipMalloc((void**)&dx, n * 8); hipMalloc((void**)&dout, 6 * n * 8);
  hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, dout, n);
  hipMemcpy(ho, dout, 6 * n * 8, hipMemcpyDeviceToHost);
  double e[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const long double tr = 1.0L / (long double)hx[i], ts = 1.0L / sqrtl((long double)hx[i]);
    for (int q = 0; q < 3; ++q) e[q] = std::fmax(e[q], (double)fabsl(((long double)ho[6 * i + q] - tr) / tr));
    for (int q = 3; q < 6; ++q) e[q] = std::fmax(e[q], (double)fabsl(((long double)ho[6 * i + q] - ts) / ts));
  }
  printf("max relative error over 2^20 values: v_rcp_f64 %.3g | + 1 Newton step %.3g | + 2 steps %.3g || v_rsq_f64 %.3g | + 1 step %.3g | + 2 steps %.3g   (2^-53 = 1.11e-16)\n",
         e[0], e[1], e[2], e[3], e[4], e[5]);
  return 0;
}
