// fused_i8_kernel (blr_fused_i8.hpp) instantiated away from the rest of the library: one translation unit per noise kind
// (-DBLR_I8_TU_DIAG=0 / 1), each with the ColVecs and the RowVecs form of the stream, and one per noise kind with the plan the handle
// option I8_GROUPS switches to (-DBLR_I8_TU_ALT=0 / 1); the Makefile compiles this file four times.
// Host side: blr_abi.hip (launch_fused_i8).  Development / sanitizer builds (-DBLR_DEV_FAST, no BLR_I8_TU_DIAG): one object
// with the isotropic ColVecs kernel only.
#include <hip/hip_runtime.h>

#include "blr_fused_i8.hpp"

namespace blr {

#if defined(BLR_DEV_FAST)
const void* i8_kernel_ptr_alt_iso(bool) { return nullptr; }
const void* i8_kernel_ptr_alt_diag(bool) { return nullptr; }
void i8_kernel_launch_alt_iso(bool, unsigned, hipStream_t, const PosteriorArgs<double>&) {}
void i8_kernel_launch_alt_diag(bool, unsigned, hipStream_t, const PosteriorArgs<double>&) {}
const void* i8_kernel_ptr_iso(bool rowv) { return rowv ? nullptr : reinterpret_cast<const void*>(fused_i8_kernel<false, false>); }
const void* i8_kernel_ptr_diag(bool) { return nullptr; }
void i8_kernel_launch_iso(bool rowv, unsigned grid, hipStream_t stream, const PosteriorArgs<double>& a) {
  if (!rowv) hipLaunchKernelGGL((fused_i8_kernel<false, false>), dim3(grid), dim3(kI8Threads), I8Cfg::LDS_BYTES, stream, a);
}
void i8_kernel_launch_diag(bool, unsigned, hipStream_t, const PosteriorArgs<double>&) {}
#elif defined(BLR_I8_TU_ALT)
// the OTHER plan of each noise kind (handle option I8_GROUPS): seven digit groups under isotropic noise (-DBLR_I8_TU_ALT=0), six under
// diagonal noise (-DBLR_I8_TU_ALT=1), both layouts each
constexpr bool kAltDiag = BLR_I8_TU_ALT != 0;
constexpr int kAltNG = kAltDiag ? 6 : 7;
#if BLR_I8_TU_ALT
const void* i8_kernel_ptr_alt_diag(bool rowv) {
#else
const void* i8_kernel_ptr_alt_iso(bool rowv) {
#endif
  return rowv ? reinterpret_cast<const void*>(fused_i8_kernel<kAltDiag, true, kAltNG>) : reinterpret_cast<const void*>(fused_i8_kernel<kAltDiag, false, kAltNG>);
}
#if BLR_I8_TU_ALT
void i8_kernel_launch_alt_diag(bool rowv, unsigned grid, hipStream_t stream, const PosteriorArgs<double>& a) {
#else
void i8_kernel_launch_alt_iso(bool rowv, unsigned grid, hipStream_t stream, const PosteriorArgs<double>& a) {
#endif
  if (rowv) hipLaunchKernelGGL((fused_i8_kernel<kAltDiag, true, kAltNG>), dim3(grid), dim3(kI8Threads), I8Cfg::LDS_BYTES, stream, a);
  else hipLaunchKernelGGL((fused_i8_kernel<kAltDiag, false, kAltNG>), dim3(grid), dim3(kI8Threads), I8Cfg::LDS_BYTES, stream, a);
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
