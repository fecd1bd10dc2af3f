// fused_i8_kernel (blr_fused_i8.hpp) instantiated away from the rest of the library: one translation unit per noise kind
// (-DBLR_I8_TU_DIAG=0 / 1), each with the ColVecs and the RowVecs form of the stream, and one with the four forms of the seven-group
// plan per noise kind (-DBLR_I8_TU_G7=0 / 1); the Makefile compiles this file four times.
// Host side: blr_abi.hip (launch_fused_i8).  Development / sanitizer builds (-DBLR_DEV_FAST, no BLR_I8_TU_DIAG): one object
// with the isotropic ColVecs kernel only.
#include <hip/hip_runtime.h>

#include "blr_fused_i8.hpp"

namespace blr {

#if defined(BLR_DEV_FAST)
const void* i8_kernel_ptr_g7_iso(bool) { return nullptr; }
const void* i8_kernel_ptr_g7_diag(bool) { return nullptr; }
void i8_kernel_launch_g7_iso(bool, unsigned, hipStream_t, const PosteriorArgs<double>&) {}
void i8_kernel_launch_g7_diag(bool, unsigned, hipStream_t, const PosteriorArgs<double>&) {}
const void* i8_kernel_ptr_iso(bool rowv) { return rowv ? nullptr : reinterpret_cast<const void*>(fused_i8_kernel<false, false>); }
const void* i8_kernel_ptr_diag(bool) { return nullptr; }
void i8_kernel_launch_iso(bool rowv, unsigned grid, hipStream_t stream, const PosteriorArgs<double>& a) {
  if (!rowv) hipLaunchKernelGGL((fused_i8_kernel<false, false>), dim3(grid), dim3(kI8Threads), I8Cfg::LDS_BYTES, stream, a);
}
void i8_kernel_launch_diag(bool, unsigned, hipStream_t, const PosteriorArgs<double>&) {}
#elif defined(BLR_I8_TU_G7)
// the seven-group plan (handle option I8_GROUPS=7: 260 instead of 174 MFMAs per k-step, A within 1e-14 instead of 3e-14 of its
// diagonal scale): one translation unit per noise kind here too (-DBLR_I8_TU_G7=0 / 1), both layouts each
constexpr bool kG7Diag = BLR_I8_TU_G7 != 0;
#if BLR_I8_TU_G7
const void* i8_kernel_ptr_g7_diag(bool rowv) {
#else
const void* i8_kernel_ptr_g7_iso(bool rowv) {
#endif
  return rowv ? reinterpret_cast<const void*>(fused_i8_kernel<kG7Diag, true, 7>) : reinterpret_cast<const void*>(fused_i8_kernel<kG7Diag, false, 7>);
}
#if BLR_I8_TU_G7
void i8_kernel_launch_g7_diag(bool rowv, unsigned grid, hipStream_t stream, const PosteriorArgs<double>& a) {
#else
void i8_kernel_launch_g7_iso(bool rowv, unsigned grid, hipStream_t stream, const PosteriorArgs<double>& a) {
#endif
  if (rowv) hipLaunchKernelGGL((fused_i8_kernel<kG7Diag, true, 7>), dim3(grid), dim3(kI8Threads), I8Cfg::LDS_BYTES, stream, a);
  else hipLaunchKernelGGL((fused_i8_kernel<kG7Diag, false, 7>), dim3(grid), dim3(kI8Threads), I8Cfg::LDS_BYTES, stream, a);
}
#elif BLR_I8_TU_DIAG
const void* i8_kernel_ptr_diag(bool rowv) {
  return rowv ? reinterpret_cast<const void*>(fused_i8_kernel<true, true>) : reinterpret_cast<const void*>(fused_i8_kernel<true, false>);
}
void i8_kernel_launch_diag(bool rowv, unsigned grid, hipStream_t stream, const PosteriorArgs<double>& a) {
  if (rowv) hipLaunchKernelGGL((fused_i8_kernel<true, true>), dim3(grid), dim3(kI8Threads), I8Cfg::LDS_BYTES, stream, a);
  else hipLaunchKernelGGL((fused_i8_kernel<true, false>), dim3(grid), dim3(kI8Threads), I8Cfg::LDS_BYTES, stream, a);
}
#else
const void* i8_kernel_ptr_iso(bool rowv) {
  return rowv ? reinterpret_cast<const void*>(fused_i8_kernel<false, true>) : reinterpret_cast<const void*>(fused_i8_kernel<false, false>);
}
void i8_kernel_launch_iso(bool rowv, unsigned grid, hipStream_t stream, const PosteriorArgs<double>& a) {
  if (rowv) hipLaunchKernelGGL((fused_i8_kernel<false, true>), dim3(grid), dim3(kI8Threads), I8Cfg::LDS_BYTES, stream, a);
  else hipLaunchKernelGGL((fused_i8_kernel<false, false>), dim3(grid), dim3(kI8Threads), I8Cfg::LDS_BYTES, stream, a);
}
#endif

}  // namespace blr
