// fused_i8_kernel (blr_fused_i8.hpp) instantiated away from the rest of the library: one translation unit per noise kind
// (-DBLR_I8_TU_DIAG=0 / 1; the Makefile compiles this file twice), each with the ColVecs and the RowVecs form of the stream.
// Host side: blr_abi.hip (launch_fused_i8).  Development / sanitizer builds (-DBLR_DEV_FAST, no BLR_I8_TU_DIAG): one object
// with the isotropic ColVecs kernel only.
#include <hip/hip_runtime.h>

#include "blr_fused_i8.hpp"

namespace blr {

#if defined(BLR_DEV_FAST)
const void* i8_kernel_ptr_iso(bool rowv) { return rowv ? nullptr : reinterpret_cast<const void*>(fused_i8_kernel<false, false>); }
const void* i8_kernel_ptr_diag(bool) { return nullptr; }
void i8_kernel_launch_iso(bool rowv, unsigned grid, hipStream_t stream, const PosteriorArgs<double>& a) {
  if (!rowv) hipLaunchKernelGGL((fused_i8_kernel<false, false>), dim3(grid), dim3(kI8Threads), I8Cfg::LDS_BYTES, stream, a);
}
void i8_kernel_launch_diag(bool, unsigned, hipStream_t, const PosteriorArgs<double>&) {}
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
