// Host side of the C ABI declared in include/blr_mi355x.h: argument validation, host<->device
// staging for BLR_MEM_HOST calls, kernel dispatch.  No exception crosses the boundary.
#include "../../include/blr_mi355x.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types only: the symbols are resolved with dlopen / dlsym on first use
#include <dlfcn.h>

#include <algorithm>
#include <array>
#include <functional>
#include <map>
#include <queue>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "blr_aux_kernels.hpp"
#include "blr_fused_small.hpp"
#include "blr_large.hpp"
#include "blr_planes.hpp"
#include "blr_dense.hpp"
#include "blr_update.hpp"
#include "blr_fused_wave.hpp"
#include "blr_fused_i8.hpp"
#include "blr_marginals.hpp"

using namespace blr;

// Run-time switches (A/B experiments and tests; the defaults are the measured best).  Read from the environment ONCE, when the
// handle is created (BLR_MI355X_<KEY>), and settable per handle with blr_set_option: no getenv on any launch path.
struct BlrOptions {
  bool no_ldsdma = false, no_wave_kernel = false, no_gram_ring = false, no_diag_split = false, no_xcd_swizzle = false,
       no_mfma_project = false, plan_debug = false, no_i8_gram = false, no_marg_gemm = false, no_i8_diag = false, no_i8_factor = false, no_i8_rowvecs = false, no_grad_gemm = false, no_i8_dense = false, no_i8_fallback = false, no_bf16x3 = false, no_planes = false, no_fp16_planes = false, planes8 = false, no_spec_rowmax = false, no_multi_planes = false;
  int wave_split = 0;     // waves per regressor of the wave kernel: 0 = router, else 1 | 2 | 4
  int chain_batch = 0;    // regressors per shared launch at D > 128: 0 = as many as the workspace holds
  int i8_probe_min = 0;   // int8 route: batches beyond this many regressors start with a probe slice; 0 = kI8ProbeMin
  int i8_groups = 0;      // int8 route: digit groups kept, 0 = six under isotropic noise and seven under diagonal noise; 6 | 7 = that plan for both noise kinds
  long chain_ws_mb = 0;   // workspace bound of such a group in MiB: 0 = kChainWorkspace
  int sweep = 0;          // blr_update_factor_* route: 0 = router, 1 = always the Givens sweep, 2 = never
  int gs_fields = 0, gs_so = 0, gs_sd = 0, gs_nl = 0;  // GRAM_SPLITS = "off-diagonal,diagonal[,nlong]" (gs_fields = numbers parsed)
  // -> 0, -2 for an unknown key, -3 for a malformed value (the codes blr_set_option documents).  value NULL or "" = the built-in default
  static bool parse_long(const char* v, long& out) {  // the whole string must be a decimal number
    char* end = nullptr;
    const long x = strtol(v, &end, 10);
    if (end == v || *end != '\0') return false;
    out = x;
    return true;
  }
  int set(const char* key, const char* value) {
    if (!key) return -2;
    if (!strncmp(key, "BLR_MI355X_", 11)) key += 11;
    const bool on = value && *value;
    auto flag = [&](bool& f) { f = on; return 0; };
    if (!strcmp(key, "NO_LDSDMA")) return flag(no_ldsdma);
    if (!strcmp(key, "NO_WAVE_KERNEL")) return flag(no_wave_kernel);
    if (!strcmp(key, "NO_GRAM_RING")) return flag(no_gram_ring);
    if (!strcmp(key, "NO_DIAG_SPLIT")) return flag(no_diag_split);
    if (!strcmp(key, "NO_XCD_SWIZZLE")) return flag(no_xcd_swizzle);
    if (!strcmp(key, "NO_MFMA_PROJECT")) return flag(no_mfma_project);
    if (!strcmp(key, "PLAN_DEBUG")) return flag(plan_debug);
    if (!strcmp(key, "NO_I8_GRAM")) return flag(no_i8_gram);
    if (!strcmp(key, "NO_MARG_GEMM")) return flag(no_marg_gemm);
    if (!strcmp(key, "NO_GRAD_GEMM")) return flag(no_grad_gemm);
    if (!strcmp(key, "NO_I8_DIAG")) return flag(no_i8_diag);
    if (!strcmp(key, "NO_I8_FACTOR")) return flag(no_i8_factor);
    if (!strcmp(key, "NO_I8_ROWVECS")) return flag(no_i8_rowvecs);
    if (!strcmp(key, "NO_I8_DENSE")) return flag(no_i8_dense);
    if (!strcmp(key, "NO_BF16X3")) return flag(no_bf16x3);
    if (!strcmp(key, "NO_PLANES")) return flag(no_planes);
    if (!strcmp(key, "NO_MULTI_PLANES")) return flag(no_multi_planes);  // logpdf_multi at D > 128, fp32: the separate residual product + panel sweep
    if (!strcmp(key, "NO_SPEC_ROWMAX")) return flag(no_spec_rowmax);  // exact row maxima (one more pass over X) instead of the sampled ones
    if (!strcmp(key, "NO_FP16_PLANES")) return flag(no_fp16_planes);
    if (!strcmp(key, "PLANES8")) return flag(planes8);
    if (!strcmp(key, "NO_I8_FALLBACK")) return flag(no_i8_fallback);
    long v = 0;
    if (!strcmp(key, "WAVE_SPLIT")) {
      if (!on) { wave_split = 0; return 0; }
      if (!parse_long(value, v) || !(v == 1 || v == 2 || v == 4)) return -3;
      wave_split = (int)v;
      return 0;
    }
    if (!strcmp(key, "CHAIN_BATCH")) {
      if (!on) { chain_batch = 0; return 0; }
      if (!parse_long(value, v) || v < 1 || v > 128) return -3;
      chain_batch = (int)v;
      return 0;
    }
    if (!strcmp(key, "I8_PROBE_MIN")) {
      if (!on) { i8_probe_min = 0; return 0; }
      if (!parse_long(value, v) || v < 256 || v > (1 << 20)) return -3;
      i8_probe_min = (int)v;
      return 0;
    }
    if (!strcmp(key, "I8_GROUPS")) {
      if (!on) { i8_groups = 0; return 0; }
      if (!parse_long(value, v) || !(v == 6 || v == 7)) return -3;
      i8_groups = (int)v;
      return 0;
    }
    if (!strcmp(key, "CHAIN_WS_MB")) {
      if (!on) { chain_ws_mb = 0; return 0; }
      if (!parse_long(value, v) || v < 1) return -3;
      chain_ws_mb = v;
      return 0;
    }
    if (!strcmp(key, "SWEEP")) {
      if (!on || !strcmp(value, "auto")) { sweep = 0; return 0; }
      if (!strcmp(value, "always")) { sweep = 1; return 0; }
      if (!strcmp(value, "never")) { sweep = 2; return 0; }
      return -3;
    }
    if (!strcmp(key, "GRAM_SPLITS")) {
      gs_fields = gs_so = gs_sd = gs_nl = 0;
      if (on) gs_fields = sscanf(value, "%d,%d,%d", &gs_so, &gs_sd, &gs_nl);
      return (!on || gs_fields >= 2) ? 0 : -3;
    }
    return -2;
  }
  void from_environment() {
    // boolean flags: a variable that is set -- even to the empty string -- switches the flag on
    for (const char* k : {"NO_LDSDMA", "NO_WAVE_KERNEL", "NO_GRAM_RING", "NO_DIAG_SPLIT", "NO_XCD_SWIZZLE", "NO_MFMA_PROJECT", "PLAN_DEBUG",
                          "NO_I8_GRAM", "NO_MARG_GEMM", "NO_GRAD_GEMM", "NO_I8_DIAG", "NO_I8_FACTOR", "NO_I8_ROWVECS", "NO_I8_DENSE", "NO_I8_FALLBACK", "NO_BF16X3",
                          "NO_PLANES", "NO_FP16_PLANES", "PLANES8", "NO_SPEC_ROWMAX", "NO_MULTI_PLANES"}) {
      const std::string name = std::string("BLR_MI355X_") + k;
      if (const char* v = getenv(name.c_str())) (void)set(k, *v ? v : "1");
    }
    // valued options: an empty variable is ignored (the built-in default stays), a malformed one too
    for (const char* k : {"WAVE_SPLIT", "CHAIN_BATCH", "CHAIN_WS_MB", "SWEEP", "GRAM_SPLITS", "I8_PROBE_MIN", "I8_GROUPS"}) {
      const std::string name = std::string("BLR_MI355X_") + k;
      if (const char* v = getenv(name.c_str()))
        if (*v) (void)set(k, v);
    }
  }
};

struct blr_handle {
  BlrOptions opt;
  int device = 0;
  int cus = 256;  // compute units of the device (MI355X: 256; a partitioned part reports its share)
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  bool async = false;
  std::string err = "";
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::vector<void*> staged;  // device buffers of the current HOST-memspace call
  char* ws = nullptr;          // grow-only scratch (factors, info)
  size_t ws_bytes = 0;
  char* feat = nullptr;        // grow-only feature matrix of blr_posterior_rff_*
  size_t feat_bytes = 0;
  char* aux = nullptr;         // grow-only: triangular-inverse images of the marginal stream (blr_marginals.hpp), temporaries of logpdf_multi
  size_t aux_bytes = 0;
  char* i8side = nullptr;      // grow-only: what the int8 route prepares per call (y / sqrt(s), 1 / sqrt(s), ... -- launch_fused_i8); a buffer of
  size_t i8side_bytes = 0;     // its own because logpdf_multi carves ITS temporaries from `aux` around a nested update that may take that route
  // counters behind blr_get_stat: [0] regressors the int8 route handed back to the fp64 kernel (cumulative); [8 + 2 k + {0, 1}]:
  // hand-backs of slice k of the current call (two banks, alternating), read by the NEXT slice's launch (launch_fused_i8)
  unsigned long long* stats_dev = nullptr;
  unsigned long long i8_attempted = 0;   // regressors sent down the int8 route (host count)
  unsigned i8_slices = 0;                // parity = bank of the per-slice hand-back counter
  const char* route = "none";            // kernel family the most recent posterior dispatch launched (blr_last_route)
  struct RffSrc { const void *Xin, *Omega, *phase; int64_t ldxin, ldo; double scale; int Din; };
  const RffSrc* rff_src = nullptr;       // set by posterior_rff around its posterior_batched call: the basis is evaluated inside the planes pass
  // set by logpdf_multi around its update of column 0 (fp32, ColVecs, D > 128, S <= 128): the other columns of Y ride through the SAME
  // planes pass, Gram launch and factorisation as one more row block (blr_planes.hpp); the group function leaves what the finish needs
  struct MultiSrc { const void* Y; int64_t ldY; int S; void* Abar; int64_t lda; void* Tfull; int DP; double* qsp; int nq; bool done; };
  MultiSrc* multi_src = nullptr;
  int64_t route_i8_B = 0;                // > 0: that dispatch took the int8 route with this many regressors (blr_last_route looks at its hand-backs)
  std::string route_buf;
  // wavefront back substitution (D > 128): tagged exchange buffer, start-order ticket counter, launch epoch
  unsigned long long* xchg = nullptr;
  size_t xchg_bytes = 0;
  unsigned* ticket = nullptr;     // [0], [2], [3]: wavefront solve (tickets, done, launch count); [16 + 128 bank + g]: arrivals of panel_chain_kernel
  unsigned panel_launches = 0;    // parity = the bank of arrival words the next panel launch counts in (it clears the other one)
  // kernels whose dynamic-LDS limit has been raised on this handle's device (hipFuncSetAttribute is per device and costs a
  // driver call: once per (handle, kernel), not once per launch -- it sat on the launch path of the 5 us wave kernel)
  std::unordered_map<const void*, size_t> lds_limit;
  // multi-round Gram launches: (row blocks, N, slots) -> (off-diagonal ranges, diagonal ranges, tiles with one range less)
  std::map<std::array<int, 3>, std::array<int, 3>> gram_plans;
  // RCCL communicator of blr_comm_init (one rank per handle / GPU); NULL until then
  ncclComm_t comm = nullptr;
  int comm_size = 0, comm_rank = 0;
};

namespace {

// ---- RCCL, bound at run time -------------------------------------------------------------------------------------
// The library has no link-time dependency on librccl: a single-GPU host never loads it, and a process that already has a
// RCCL (PyTorch ships one) gets that one.  Only the calls of the path's single exchange are bound.
struct RcclApi {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
  std::string why;
};
RcclApi& rccl() {
  static RcclApi api = [] {
    RcclApi a;
    void* lib = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (lib) break;
    }
    if (!lib) { a.why = std::string("librccl not found: ") + dlerror(); return a; }
    auto sym = [&](const char* n) { return dlsym(lib, n); };
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(sym("ncclAllGather"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(sym("ncclAllReduce"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
    a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllGather && a.AllReduce && a.GetErrorString;
    if (!a.ok) a.why = "librccl lacks a required symbol";
    return a;
  }();
  return api;
}
int rccl_fail(blr_handle* h, ncclResult_t r, const char* what) {
  if (h) h->err = std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(r) : "RCCL error");
  return -(2000 + (int)r);
}

constexpr int kMaxSmallD = 128;
constexpr int kMaxLargeD = 8192;

int hip_fail(blr_handle* h, hipError_t e, const char* what) {
  if (h) {
    h->err = std::string(what) + ": " + hipGetErrorString(e);
  }
  return -(1000 + (int)e);
}
#define HIP_TRY(h, expr)                                   \
  do {                                                     \
    hipError_t e__ = (expr);                               \
    if (e__ != hipSuccess) return hip_fail(h, e__, #expr); \
  } while (0)

int set_lds_once(blr_handle* h, const void* kern, size_t bytes) {
  auto it = h->lds_limit.find(kern);
  if (it != h->lds_limit.end() && it->second >= bytes) return 0;
  HIP_TRY(h, hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  h->lds_limit[kern] = bytes;
  return 0;
}

int bad_arg(blr_handle* h, int pos, const char* why) {
  if (h) h->err = std::string("argument ") + std::to_string(pos) + ": " + why;
  return -pos;
}

struct Staging {  // RAII for the device temporaries of one call; nests (an inner guard frees only what it added)
  blr_handle* h;
  size_t base;
  explicit Staging(blr_handle* hh) : h(hh), base(hh->staged.size()) {}
  ~Staging() {
    for (size_t i = base; i < h->staged.size(); ++i) (void)hipFree(h->staged[i]);
    h->staged.resize(base);
  }
};

template <typename T>
int stage_in(blr_handle* h, const T* host, size_t count, const T** dev) {
  *dev = nullptr;
  if (!host || count == 0) return 0;
  void* p = nullptr;
  HIP_TRY(h, hipMalloc(&p, count * sizeof(T)));
  h->staged.push_back(p);
  HIP_TRY(h, hipMemcpyAsync(p, host, count * sizeof(T), hipMemcpyHostToDevice, h->stream));
  *dev = static_cast<const T*>(p);
  return 0;
}
template <typename T>
int stage_out_alloc(blr_handle* h, const T* host_initial, size_t count, T** dev) {
  // outputs with gaps (ld > rows, stride > extent) keep the caller's bytes in the gaps: copy the
  // current host contents in first.
  *dev = nullptr;
  if (!host_initial || count == 0) return 0;
  void* p = nullptr;
  HIP_TRY(h, hipMalloc(&p, count * sizeof(T)));
  h->staged.push_back(p);
  HIP_TRY(h, hipMemcpyAsync(p, host_initial, count * sizeof(T), hipMemcpyHostToDevice, h->stream));
  *dev = static_cast<T*>(p);
  return 0;
}

int ensure_ws(blr_handle* h, size_t bytes) {
  if (bytes <= h->ws_bytes) return 0;
  if (h->ws) {
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipFree(h->ws));
    h->ws = nullptr;
    h->ws_bytes = 0;
  }
  size_t want = std::max(bytes, (size_t)1 << 20);
  HIP_TRY(h, hipMalloc((void**)&h->ws, want));
  h->ws_bytes = want;
  return 0;
}

int ensure_i8side(blr_handle* h, size_t bytes) {
  if (bytes <= h->i8side_bytes) return 0;
  if (h->i8side) {
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipFree(h->i8side));
    h->i8side = nullptr;
    h->i8side_bytes = 0;
  }
  HIP_TRY(h, hipMalloc((void**)&h->i8side, bytes));
  h->i8side_bytes = bytes;
  return 0;
}
int ensure_stats(blr_handle* h) {
  if (h->stats_dev) return 0;
  HIP_TRY(h, hipMalloc((void**)&h->stats_dev, 16 * sizeof(unsigned long long)));
  HIP_TRY(h, hipMemsetAsync(h->stats_dev, 0, 16 * sizeof(unsigned long long), h->stream));
  return 0;
}

int ensure_aux(blr_handle* h, size_t bytes) {
  if (bytes <= h->aux_bytes) return 0;
  if (h->aux) {
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipFree(h->aux));
    h->aux = nullptr;
    h->aux_bytes = 0;
  }
  HIP_TRY(h, hipMalloc((void**)&h->aux, bytes));
  h->aux_bytes = bytes;
  return 0;
}

// Exchange buffer of backsolve_wave_kernel: zero at allocation, afterwards only written by that kernel with launch
// epochs that are never reused, so a stale granule can never carry the current tag.
int ensure_xchg(blr_handle* h, size_t bytes) {
  if (!h->ticket) {
    HIP_TRY(h, hipMalloc((void**)&h->ticket, 2048));
    HIP_TRY(h, hipMemsetAsync(h->ticket, 0, 2048, h->stream));
  }
  if (bytes <= h->xchg_bytes) return 0;
  if (h->xchg) {
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipFree(h->xchg));
    h->xchg = nullptr;
    h->xchg_bytes = 0;
  }
  size_t want = std::max(bytes, (size_t)1 << 18);
  HIP_TRY(h, hipMalloc((void**)&h->xchg, want));
  HIP_TRY(h, hipMemsetAsync(h->xchg, 0, want, h->stream));
  h->xchg_bytes = want;
  return 0;
}
template <typename T>
constexpr size_t wave_solve_lds() {
  return (((size_t)kPB * (kPB + 1) / 2 * sizeof(T) + 15) & ~(size_t)15) + 3 * kPB * sizeof(T) + 8 * sizeof(double) + 16;
}
// launches one wavefront solve of NC x S workgroups; fills the synchronisation fields of `b`
template <typename T>
int launch_wave_solve(blr_handle* h, WaveSolveArgs<T>& b, int NC, int64_t S) {
  int rc = ensure_xchg(h, (size_t)S * b.DP * 2 * sizeof(unsigned long long));
  if (rc) return rc;
  b.xchg = h->xchg; b.ticket = h->ticket;  // tickets, the launch count behind the granule tags: device-side (WaveSolveArgs)
  { const int rc_lds = set_lds_once(h, reinterpret_cast<const void*>(backsolve_wave_kernel<T>), (size_t)((int)wave_solve_lds<T>())); if (rc_lds) return rc_lds; }
  hipLaunchKernelGGL(backsolve_wave_kernel<T>, dim3(NC, (unsigned)S), dim3(kThreads), wave_solve_lds<T>(), h->stream, b);
  const hipError_t le = hipGetLastError();
  if (le != hipSuccess) return hip_fail(h, le, "launch of backsolve_wave_kernel");  // nothing ran: the counters did not move
  return 0;
}

size_t extent(int64_t B, int64_t stride, size_t one) {  // elements spanned by B items at `stride`
  if (B <= 0) return 0;
  return (size_t)((B - 1) * stride) + one;
}
size_t mat_extent(int64_t rows, int64_t cols, int64_t ld) {
  if (rows <= 0 || cols <= 0) return 0;
  return (size_t)((cols - 1) * ld + rows);
}

template <typename T>
bool aligned16(const T* p, int64_t ld, int64_t stride) {
  return ((uintptr_t)p % 16 == 0) && ((ld * (int64_t)sizeof(T)) % 16 == 0) && ((stride * (int64_t)sizeof(T)) % 16 == 0);
}

// ---- fused small-D dispatch -----------------------------------------------------------------------
template <typename T, int NB, int MODE>
int launch_fused_small(blr_handle* h, const PosteriorArgs<T>& a) {
  using C = SmallCfg<T, NB>;
  auto kern = fused_small_kernel<T, NB, MODE>;
  // every launch: the attribute is per DEVICE, and handles on different devices share this code (a process-wide
  // "already set" flag left the second device at the 64 KB default)
  { const int rc_lds = set_lds_once(h, reinterpret_cast<const void*>(kern), (size_t)(C::LDS_BYTES)); if (rc_lds) return rc_lds; }
  int grid = (int)std::min<int64_t>(a.B, 1 << 20);
  if (a.retry_only) grid = std::min(grid, 2 * h->cus);  // (one round of workgroups, each walking its share: see the kernel's head)
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), C::LDS_BYTES, h->stream, a);
  HIP_TRY(h, hipGetLastError());
  if (!a.retry_only) {
    static const std::string name = std::string("fused_small_kernel<") + (sizeof(T) == 8 ? "double" : "float") + ", " + std::to_string(NB) + ", " +
                                    std::to_string(MODE) + ">";
    h->route = name.c_str();
    h->route_i8_B = 0;
  }
  return 0;
}

template <typename T, int NB>
int launch_fused_small_mode(blr_handle* h, const PosteriorArgs<T>& a) {
  if (a.layout == BLR_LAYOUT_ROWVECS) return launch_fused_small<T, NB, 1>(h, a);
  if (a.vec_ok) {
    if (h->opt.no_ldsdma) return launch_fused_small<T, NB, 3>(h, a);  // A/B experiments only
    return launch_fused_small<T, NB, 4>(h, a);
  }
  return launch_fused_small<T, NB, 0>(h, a);
}

// D = 32 / 64, ColVecs with aligned columns: one wavefront per regressor (blr_fused_wave.hpp)
template <typename T, int NB, int NW>
int launch_fused_wave_nw(blr_handle* h, const PosteriorArgs<T>& a) {
  using C = WaveCfg<T, NB>;
  const int grid = (int)std::min<int64_t>(a.B, (int64_t)h->cus * 8 / NW);  // 8 waves per CU (18.6 KB of LDS each)
  if (NW * C::LDS_BYTES > 64 * 1024) {
    int rc = set_lds_once(h, reinterpret_cast<const void*>(fused_wave_kernel<T, NB, NW>), (size_t)NW * C::LDS_BYTES);
    if (rc) return rc;
  }
  hipLaunchKernelGGL((fused_wave_kernel<T, NB, NW>), dim3(grid), dim3(64 * NW), NW * C::LDS_BYTES, h->stream, a);
  HIP_TRY(h, hipGetLastError());
  static const std::string name = std::string("fused_wave_kernel<") + (sizeof(T) == 8 ? "double" : "float") + ", " + std::to_string(NB) + ", " +
                                  std::to_string(NW) + ">";
  h->route = name.c_str();
  h->route_i8_B = 0;
  return 0;
}
// The chip has 2048 wave slots for this kernel.  A batch that cannot fill them with one regressor per wave splits each
// regressor's observations over 2 or 4 waves instead (BLR_MI355X_WAVE_SPLIT=1|2|4 overrides; the tests run all three).
template <typename T, int NB>
int launch_fused_wave(blr_handle* h, const PosteriorArgs<T>& a) {
  int nw = a.B >= 2048 ? 1 : (a.B >= 1024 ? 2 : 4);
  if (h->opt.wave_split) nw = h->opt.wave_split;
  if (nw == 1) return launch_fused_wave_nw<T, NB, 1>(h, a);
  if (nw == 2) return launch_fused_wave_nw<T, NB, 2>(h, a);
  return launch_fused_wave_nw<T, NB, 4>(h, a);
}

// D = 128, fp64, aligned ColVecs, isotropic noise, diagonal prior, whole 32-column k-steps: the Gram matrix on the int8 matrix
// cores (blr_fused_i8.hpp), followed by the fp64 kernel in retry-only mode for the regressors the fast path handed back
// (a row bound broken, non-finite input): every regressor leaves with the status and the numbers of an
// fp64-accurate update, none is computed twice on the fast path.
constexpr size_t kSmall8Lds = SmallCfg<double, 8>::LDS_BYTES;
int launch_fused_i8(blr_handle* h, const PosteriorArgs<double>& a) {
  const bool diag = a.noise_kind == BLR_NOISE_DIAGONAL, rowv = a.layout == BLR_LAYOUT_ROWVECS;
  // (the kernel's four instantiations: blr_i8_kernels.hip)
  // (default plans: six digit groups under isotropic noise, seven under diagonal noise; option I8_GROUPS asks for the other one)
  const int groups_default = diag ? kI8GroupsDiag : kI8GroupsIso;
  const bool alt = h->opt.i8_groups != 0 && h->opt.i8_groups != groups_default && i8_kernel_ptr_alt(diag, rowv) != nullptr;
  const int groups = alt ? h->opt.i8_groups : groups_default;
  const void* const kern = alt ? i8_kernel_ptr_alt(diag, rowv) : (diag ? i8_kernel_ptr_diag(rowv) : i8_kernel_ptr_iso(rowv));
  if (kern == nullptr) return hip_fail(h, hipErrorInvalidValue, "this build lacks the requested form of the int8 kernel");
  int rc = set_lds_once(h, kern, (size_t)I8Cfg::LDS_BYTES);
  if (rc) return rc;
  if ((rc = ensure_stats(h))) return rc;
  // diagonal noise: y / sqrt(s), 1 / sqrt(s), sum log s and a validity flag per regressor, once per call, in a side buffer of the
  // handle (16 bytes per observation: slices of at most 1 GiB)
  const int64_t per_reg = 2 * (int64_t)a.N * (int64_t)sizeof(double);
  const int grid = (int)std::min<int64_t>(a.B, diag ? std::max<int64_t>(1, ((int64_t)1 << 30) / per_reg) : (1 << 20));
  double *yt = nullptr, *rw = nullptr, *ld = nullptr, *rmx = nullptr;
  int32_t* bad = nullptr;
  // dense prior: logdet Lw and the status of its Cholesky per prior, from i8_prior_logdet_kernel (ONE prior when the batch shares it)
  const bool dense = a.prior_kind == BLR_PRIOR_DENSE;
  const bool shared_prior = a.strideLw == 0 || a.B == 1;
  double* pld = nullptr;
  int32_t* pinfo = nullptr;
  size_t o_prior = 0;
  if (diag) {
    const size_t o_rw = (((size_t)grid * a.N * sizeof(double)) + 255) & ~(size_t)255;
    o_prior = 2 * o_rw + 2 * (((size_t)grid * sizeof(double) + 255) & ~(size_t)255) + (((size_t)grid * sizeof(int32_t) + 255) & ~(size_t)255);
  }
  if (dense) {
    const size_t np = shared_prior ? 1 : (size_t)grid;
    if ((rc = ensure_i8side(h, o_prior + ((np * sizeof(double) + 255) & ~(size_t)255) + np * sizeof(int32_t)))) return rc;
    if ((rc = set_lds_once(h, reinterpret_cast<const void*>(i8_prior_logdet_kernel), kSmall8Lds))) return rc;
  }
  if (diag) {
    const size_t o_rw = (((size_t)grid * a.N * sizeof(double)) + 255) & ~(size_t)255;
    const size_t o_ld = 2 * o_rw, o_mx = o_ld + (((size_t)grid * sizeof(double) + 255) & ~(size_t)255);
    const size_t o_bad = o_mx + (((size_t)grid * sizeof(double) + 255) & ~(size_t)255);
    if (!dense && (rc = ensure_i8side(h, o_bad + (size_t)grid * sizeof(int32_t)))) return rc;
    yt = reinterpret_cast<double*>(h->i8side); rw = reinterpret_cast<double*>(h->i8side + o_rw);
    ld = reinterpret_cast<double*>(h->i8side + o_ld); rmx = reinterpret_cast<double*>(h->i8side + o_mx); bad = reinterpret_cast<int32_t*>(h->i8side + o_bad);
  }
  // Slices: one workgroup per regressor (batches beyond 2^20, or beyond the side buffer, in several launches).  A batch of more than
  // kI8ProbeMin = 4096 regressors (option I8_PROBE_MIN) starts with a PROBE slice of kI8Probe (one round of workgroups on the chip): every later slice reads how
  // many regressors of the slice before it the fast path had to hand back, and when that was more than a quarter its workgroups
  // leave their regressors to the fp64 kernel at once instead of streaming them twice (heavy-tailed inputs: blr_get_stat
  // "i8_handed_back").  The decision depends on the data of the previous slice only: same inputs, same bits.
  if (dense) {
    const size_t np = shared_prior ? 1 : (size_t)grid;
    pld = reinterpret_cast<double*>(h->i8side + o_prior);
    pinfo = reinterpret_cast<int32_t*>(h->i8side + o_prior + ((np * sizeof(double) + 255) & ~(size_t)255));
    if (shared_prior) {
      hipLaunchKernelGGL(i8_prior_logdet_kernel, dim3(1), dim3(kThreads), kSmall8Lds, h->stream, a.Lw, a.ldl, (int64_t)0, 1, pld, pinfo);
      HIP_TRY(h, hipGetLastError());
    }
  }
  int64_t b0 = 0;
  int prev_n = 0;
  while (b0 < a.B) {
    PosteriorArgs<double> s = a;
    int64_t want = std::min<int64_t>(grid, a.B - b0);
    if (b0 == 0 && a.B > (h->opt.i8_probe_min > 0 ? h->opt.i8_probe_min : kI8ProbeMin) && !h->opt.no_i8_fallback) want = std::min<int64_t>(want, kI8Probe);
    const int nb = (int)want;
    s.B = nb;
    s.X += b0 * a.strideX; s.y += b0 * a.stridey; s.s += b0 * a.strides; s.mw += b0 * a.stridemw; s.Lw += b0 * a.strideLw;
    if (s.mw_post) s.mw_post += b0 * a.stride_mwpost;
    if (s.T_post) s.T_post += b0 * a.strideT;
    if (s.Lw_post) s.Lw_post += b0 * a.strideLp;
    if (s.logpdf) s.logpdf += b0;
    s.info += b0;
    s.i8_handed_tot = h->stats_dev;
    s.i8_call_base = b0 == 0 ? h->stats_dev + 2 : nullptr;
    s.i8_handed_slice = h->stats_dev + 8 + (h->i8_slices & 1u);         // zeroed by this slice's int8 launch, counted up by its retry launch
    s.i8_prev_handed = h->stats_dev + 8 + ((h->i8_slices & 1u) ^ 1u);   // final since the previous slice's retry launch
    s.i8_prev_n = h->opt.no_i8_fallback ? 0 : prev_n;
    ++h->i8_slices;
    if (diag) {
      s.i8_yt = yt; s.i8_rw = rw; s.i8_stride = a.N; s.i8_logdet = ld; s.i8_bad = bad; s.i8_rwmax = rmx;
      hipLaunchKernelGGL(i8_noise_prep_kernel, dim3(nb), dim3(kThreads), 0, h->stream, s.s, a.strides, s.y, a.stridey, (int)a.N, yt, rw, (int64_t)a.N, ld,
                         bad, rmx);
    }
    if (dense) {
      if (!shared_prior) {
        hipLaunchKernelGGL(i8_prior_logdet_kernel, dim3(std::min(nb, 2 * h->cus)), dim3(kThreads), kSmall8Lds, h->stream, s.Lw, a.ldl,
                           a.strideLw, nb, pld, pinfo);
        HIP_TRY(h, hipGetLastError());
      }
      s.i8_prior_logdet = pld; s.i8_prior_info = pinfo; s.i8_prior_stride = shared_prior ? 0 : 1;
    }
    if (alt) i8_kernel_launch_alt(diag, rowv, (unsigned)nb, h->stream, s);
    else if (diag) i8_kernel_launch_diag(rowv, (unsigned)nb, h->stream, s);
    else i8_kernel_launch_iso(rowv, (unsigned)nb, h->stream, s);
    HIP_TRY(h, hipGetLastError());
    s.retry_only = 1;
    if ((rc = launch_fused_small_mode<double, 8>(h, s))) return rc;
    h->i8_attempted += (unsigned long long)nb;
    prev_n = nb;
    b0 += nb;
  }
  h->route = !alt ? "fused_i8_kernel" : (groups == 7 ? "fused_i8_kernel (7 digit groups)" : "fused_i8_kernel (6 digit groups)");
  h->route_i8_B = a.B;
  return 0;
}

template <typename T>
int dispatch_fused_small(blr_handle* h, const PosteriorArgs<T>& a) {
  int NB = (a.D + 15) / 16;
  if constexpr (sizeof(T) == 8) {
    // (RowVecs: a feature's 32 observations of a k-step are 16 DMA lanes of 16 bytes: rows and regressors 16-byte aligned)
    const bool i8_layout = a.layout == BLR_LAYOUT_COLVECS
                               ? a.vec_ok
                               : (!h->opt.no_i8_rowvecs && ((uintptr_t)a.X & 15) == 0 && (a.ldx & 1) == 0 && (a.B == 1 || (a.strideX & 1) == 0));
    const bool i8_built = (a.noise_kind == BLR_NOISE_DIAGONAL ? i8_kernel_ptr_diag(a.layout == BLR_LAYOUT_ROWVECS) : i8_kernel_ptr_iso(a.layout == BLR_LAYOUT_ROWVECS)) != nullptr;
    if (i8_built && !h->opt.no_i8_gram && !h->opt.no_ldsdma && a.D == 128 && i8_layout &&
        (a.noise_kind == BLR_NOISE_ISOTROPIC || (a.noise_kind == BLR_NOISE_DIAGONAL && !h->opt.no_i8_diag)) &&
        (a.prior_kind == BLR_PRIOR_DIAGONAL || (a.prior_kind == BLR_PRIOR_UPPER_FACTOR && !h->opt.no_i8_factor) ||
         (a.prior_kind == BLR_PRIOR_DENSE && !h->opt.no_i8_dense)) && a.N - a.N % I8Cfg::KC >= kI8MinN && a.N - a.N % I8Cfg::KC <= kI8MaxN && a.ldx * 8 * I8Cfg::KC < ((int64_t)1 << 31))
      return launch_fused_i8(h, a);
  }
  if (!h->opt.no_wave_kernel && a.layout == BLR_LAYOUT_COLVECS && a.vec_ok && a.D == 16 * NB &&
      (3 * a.ldx + 64) * (int64_t)sizeof(T) < ((int64_t)1 << 31)) {
    if (NB == 4) return launch_fused_wave<T, 4>(h, a);
    if constexpr (sizeof(T) == 8) {
      if (NB == 2) return launch_fused_wave<T, 2>(h, a);
    }
  }
  switch (NB) {
    case 1: return launch_fused_small_mode<T, 1>(h, a);
    case 2: return launch_fused_small_mode<T, 2>(h, a);
    case 3: return launch_fused_small_mode<T, 3>(h, a);
    case 4: return launch_fused_small_mode<T, 4>(h, a);
    case 5: return launch_fused_small_mode<T, 5>(h, a);
    case 6: return launch_fused_small_mode<T, 6>(h, a);
    case 7: return launch_fused_small_mode<T, 7>(h, a);
    case 8: return launch_fused_small_mode<T, 8>(h, a);
    default: return bad_arg(h, 5, "D > 128 is not supported by this build");
  }
}

// ---- large-D path (D > 128): multi-kernel pipeline of blr_large.hpp ---------------------------------------------
constexpr int kChainBatchMaxWords = 128;  // = kChainBatchMax below
template <typename T>
int set_lds(blr_handle* h, const void* kern, size_t bytes) { return set_lds_once(h, kern, bytes); }

// In-place blocked (128) right-looking Cholesky of the lower triangle of M (nrows_total x DP, ld); rows beyond DP
// (the right-hand-side block of the augmented matrix) are carried through the TRSM and the trailing updates.
template <typename T, int ER>
int launch_panel(blr_handle* h, T* M, int64_t ld, int p, int nrows_total, int nbelow, int32_t* info_dev, int G, int64_t batch_stride,
                 int info_stride) {
  constexpr int NW = BLR_PANEL_WAVES;
  using CC = ChainCfg<T, NW, ER>;
  int rc;
  if ((rc = set_lds<T>(h, reinterpret_cast<const void*>(panel_chain_kernel<T, NW, ER>), CC::LDS_BYTES))) return rc;
  const int nwg = std::max(1, (nbelow + ER - 1) / ER);
  // arrival counters (one per factorisation of the launch): count up to nwg during the launch.  Two banks, used alternately
  // by the launches of this handle (= of its stream, in order); a launch clears the bank of its successor (blr_panel.hpp).
  // Forward progress of the wait inside the kernel: workgroup 0 waits for workgroups of ITS OWN launch only, every one of
  // which arrives after its loads without waiting for anybody -- so it holds for any dispatch order as long as each
  // workgroup is eventually scheduled, which a grid of at most one workgroup per CU (callers: G * nwg <= cus, or ER = 32
  // and D <= 8192) always is; beyond that the bounded spin reports -999 instead of a wrong factor.
  static_assert(kChainBatchMaxWords == kPanelArriveWords, "one arrival word per factorisation of a group");
  unsigned* const bank = h->ticket + 16 + kPanelArriveWords * (h->panel_launches & 1u);
  unsigned* const next = h->ticket + 16 + kPanelArriveWords * ((h->panel_launches & 1u) ^ 1u);
  ++h->panel_launches;
  hipLaunchKernelGGL((panel_chain_kernel<T, NW, ER>), dim3(nwg, G), dim3(64 * NW), CC::LDS_BYTES, h->stream, M, ld, p * kPB, nrows_total,
                     info_dev, bank, (unsigned)nwg, batch_stride, info_stride, next);
  HIP_TRY(h, hipGetLastError());
  return 0;
}

// ... as long as their workspaces fit this many bytes (per handle; option CHAIN_WS_MB lowers the bound, blr_release_workspace
// hands the memory back: the scratch buffers of a handle only ever grow otherwise)
constexpr size_t kChainWorkspace = (size_t)8 << 30;
constexpr int kChainBatchMax = kChainBatchMaxWords;  // factorisations that step through their panels in shared launches (one arrival word each per bank)

// Blocked Cholesky of G independent matrices M + g * batch_stride (status words info_dev + g * info_stride), panel by panel,
// every step ONE launch over all of them.
template <typename T>
int chol_large(blr_handle* h, T* M, int64_t ld, int DP, int nrows_total, int32_t* info_dev, int G = 1, int64_t batch_stride = 0,
               int info_stride = 0) {
  const int NC = DP / kPB;
  int rc;
  if (G < 1 || G > kChainBatchMax) return hip_fail(h, hipErrorInvalidValue, "chol_large: too many factorisations per launch");
  if ((rc = ensure_xchg(h, 0))) return rc;  // the handle's counter words (ticket[16 + g]: arrivals of panel_chain_kernel)
  if ((rc = set_lds<T>(h, reinterpret_cast<const void*>(trail_update_kernel<T>), TrailCfg<T>::LDS_BYTES))) return rc;
  for (int p = 0; p < NC; ++p) {
    // L_pp and X <- X L_pp^-T for the rows below.  Every workgroup factors L_pp and takes 16 or 32 of those rows along: the
    // fewer, the shorter the launch (the update waves are its bottleneck) -- as long as every workgroup has a CU to itself
    // (D <= 8192: at most 8128 rows below a block, 254 workgroups of 32)
    const int nbelow = nrows_total - (p + 1) * kPB;
    const int cus = h->cus;
    if ((int64_t)G * ((nbelow + 15) / 16) <= cus) rc = launch_panel<T, 16>(h, M, ld, p, nrows_total, nbelow, info_dev, G, batch_stride, info_stride);
    else rc = launch_panel<T, 32>(h, M, ld, p, nrows_total, nbelow, info_dev, G, batch_stride, info_stride);
    if (rc) return rc;
    const int m = NC - 1 - p;  // remaining column blocks
    if (m > 0) {
      const int ntri = 2 * m;                                   // 64-row sub-blocks of the remaining triangle
      const int nextra = (nrows_total - DP) / TrailCfg<T>::SB;  // rhs rows below the square part
      const int ntiles = ntri * (ntri + 1) / 2 + nextra * ntri;
      const int slots = h->cus * (TrailCfg<T>::LDS_BYTES <= 80 * 1024 ? 2 : 1);  // sub-tiles the chip holds at once
      const int gx = std::min(ntiles, std::max(1, slots / G));
      hipLaunchKernelGGL(trail_update_kernel<T>, dim3(gx, G), dim3(kThreads), TrailCfg<T>::LDS_BYTES, h->stream, M, ld,
                         p, ntri, (p + 1) * kPB, DP, (const int32_t*)info_dev, ntiles, batch_stride, info_stride);
      HIP_TRY(h, hipGetLastError());
    }
  }
  HIP_TRY(h, hipGetLastError());
  return 0;
}

// Multi-round Gram launch of ONE regressor (c5: 136 tiles on 512 slots): with one split factor the launch takes whole rounds of
// the off-diagonal workgroup length (952 workgroups = 2 rounds at 93 % fill).  Three kinds of work items, dispatched longest
// first -- diagonal tiles with `sd` column ranges, `nlong` strictly lower tiles with so - 1 ranges, the others with so -- let
// a second round of SHORTER workgroups follow a first round of longer ones (c5: 64 x 3205 + 448 x 2597, then 448 x 2304 column
// units instead of 2 x 2597 everywhere).  The plan is the makespan of list scheduling on `slots` slots over a small neighbourhood
// of the one-factor choice s0 (10 ms of host time, once per shape and handle); `unit` = cost of a diagonal column / off-diagonal.
struct GramPlan { int so, sd, nlong; double makespan; };
inline double gram_makespan(int n_off, int NC, int so, int sd, int nlong, int N, int nsc, int slots, double unit, bool interleaved) {
  auto cols = [&](int sp) { return (double)(((N + sp - 1) / sp + nsc - 1) / nsc * nsc); };
  std::priority_queue<double, std::vector<double>, std::greater<double>> free_at;
  for (int i = 0; i < slots; ++i) free_at.push(0.0);
  double end = 0.0;
  auto run = [&](double cost) { double t = free_at.top() + cost; free_at.pop(); free_at.push(t); end = std::max(end, t); };
  if (interleaved) {  // one factor: item w -> tile w % ntiles, diagonal tiles spread over the launch
    const int ntiles = n_off + NC;
    for (int w = 0; w < ntiles * so; ++w) {
      const int t = w % ntiles;
      int ii = 0;
      while ((ii + 1) * (ii + 2) / 2 <= t) ++ii;
      const bool diag = t - ii * (ii + 1) / 2 == ii;
      run((diag ? unit : 1.0) * cols(so) + 256.0);
    }
    return end;
  }
  for (int i = 0; i < NC * sd; ++i) run(unit * cols(sd) + 256.0);
  for (int i = 0; i < nlong * (so - 1); ++i) run(cols(so - 1) + 256.0);
  for (int i = 0; i < (n_off - nlong) * so; ++i) run(cols(so) + 256.0);
  return end;
}
inline GramPlan plan_gram_rounds(int n_off, int NC, int s0, int N, int nsc, int slots, int max_split, double unit) {
  const double base = gram_makespan(n_off, NC, s0, s0, 0, N, nsc, slots, unit, true);
  // Candidates are the family that measured well on c5 (tools/scan_splits.sh): one or two ranges more than the one-factor choice,
  // diagonal tiles with as many ranges as the short tiles or half as many, the long tiles in sixteenths of the triangle.
  // (Measured against the model at D = 2048, N = 16384, ms per update: 7 | 7 -> 1.045; 8 | 8, 64 long -> 0.999; 8 | 4, 64 ->
  // 0.997; 8 | 6, 56 -> 0.999; 8 | 8, 56 (1032 workgroups: a third round) -> 1.065; 8 | 7, 64 -> 1.076 against a modelled tie.)
  GramPlan best{s0, 0, 0, base};
  for (int so = std::max(2, s0); so <= std::min(max_split, s0 + 2); ++so)
    for (int sd : {so, so / 2}) {
      if (sd < 1) continue;
      // the run of nlong values that tie for the best makespan of this (so, sd): take its middle (both ends sit next to a cliff)
      double m_best = 1e300;
      int first = -1, last = -1;
      for (int f = 0; f <= 16; ++f) {
        const double m = gram_makespan(n_off, NC, so, sd, n_off * f / 16, N, nsc, slots, unit, false);
        if (m < m_best * 0.995) { m_best = m; first = last = f; }
        else if (m <= m_best * 1.005 && f == last + 1) last = f;
      }
      if (m_best < best.makespan * 0.995) best = GramPlan{so, sd, n_off * ((first + last) / 2) / 16, m_best};
    }
  if (best.sd == 0 || best.makespan > 0.96 * base) return GramPlan{s0, 0, 0, base};  // not worth leaving the one-factor launch
  return best;
}

// `G` regressors reg0 .. reg0 + G - 1 of the batch, each with its own copy of the workspace, in every launch of the update
// (regressor from blockIdx.y / .z): the ~2 dispatches per panel of the factorisation are latency, not throughput, and so are
// the small kernels around the Gram launch.  G = 1 is the single-regressor path.
template <typename T>
int posterior_large_group(blr_handle* h, const PosteriorArgs<T>& a, int64_t reg0, int G, int* G_done = nullptr) {
  // fp32 Gram on the bf16 matrix cores: only where the ring loop runs (aligned ColVecs -- use_dma == 1).  RowVecs / unaligned fp32 calls
  // take the plain-f32 instantiation, its split plan and its label (they used to launch the BF3K instantiation's f32 loops,
  // 15 % slower, with a plan costed for bf16 diagonal tiles and a route that named a kernel form that never ran).
  const bool bf3 = sizeof(T) == 4 && !h->opt.no_bf16x3 && !h->opt.no_gram_ring && a.layout == LAYOUT_COLVECS &&
                   ((uintptr_t)(a.X + reg0 * a.strideX) % 16 == 0) && ((a.ldx * (int64_t)sizeof(T)) % 16 == 0);
  // fp32, ColVecs (any alignment): the operands of the Gram product are split into their three bf16 planes ONCE, in the fragment order
  // of the matrix instruction, and the Gram launch only moves and multiplies them (blr_planes.hpp)
  const bool rff = a.rff_Omega != nullptr;
  const bool planes = sizeof(T) == 4 && !h->opt.no_bf16x3 && !h->opt.no_planes && a.layout == LAYOUT_COLVECS && a.N > 0;
  if (rff && !planes) return hip_fail(h, hipErrorInvalidValue, "a basis that is not materialised needs the planes path");
  h->route_i8_B = 0;
  const int NP = h->opt.no_fp16_planes ? 3 : 2;  // planes per operand: two fp16 (three products) or three bf16 (six)
  h->route = sizeof(T) == 8 ? "gram_tile_kernel<double>"
                            : (planes ? (NP == 2 ? (h->opt.planes8 ? "gram_planes_kernel<2>" : "gram_planes4_kernel") : "gram_planes_kernel<3>")
                                      : (bf3 ? "gram_tile_kernel<float, true>" : "gram_tile_kernel<float>"));  // (large-D pipeline: the Gram launch dominates; <float, true>: full tiles on the bf16 matrix cores)
  using LC = LargeCfg<T>;
  const int D = a.D, N = a.N;
  const int DP = (D + kPB - 1) / kPB * kPB, NC = DP / kPB;
  const int64_t lda = DP + kPB;
  const int ntiles = NC * (NC + 1) / 2;
  // multi-output evidence: one more row block of operand planes (the residuals of Y's columns), its macro tiles in the Gram launch
  blr_handle::MultiSrc* const ms = h->multi_src;
  if (ms && !(planes && !rff && !h->opt.no_fp16_planes && G == 1 && a.prior_kind != PRIOR_UPPER_FACTOR && ms->S >= 1 && ms->S <= kPB))
    return hip_fail(h, hipErrorInvalidValue, "multi-output rows need the fp16 planes path (internal)");
  const int NCA = NC + (ms ? 1 : 0), DPA = NCA * kPB;
  const int ntiles_g = NCA * (NCA + 1) / 2;
  const int nstage_cols = LC::NSC;
  // The split plan depends on how many regressors share the launch -- and the workspace one regressor needs depends on the
  // plan.  Plan for the requested group first, clamp the group to the workspace bound, then plan AGAIN for the group that will
  // really run (a clamped group used to run with the split of the larger one: under-split, fewer rounds than modelled).
  int nsplit = 1, nsplit_diag = 0, nlong = 0;
  const int max_split_cols = std::max(1, (N + nstage_cols - 1) / nstage_cols);
  const int max_split = std::min(64, max_split_cols);
  auto plan_splits = [&](int G) {
    // split-K factor: fill the 2 x 256 workgroup slots of the chip in whole rounds (576 workgroups on 512 slots
    // would take two rounds for 1.125 rounds of work)
    // model of the launch: rounds x (columns per workgroup + 256) -- the 256 stands for a workgroup's fixed costs (pipeline fill,
    // the 64 KB partial it writes and the reduction reads back), fitted on c5 (136 tiles: 15 splits 1.232 ms, 11: 1.204, 7: 1.210)
    nsplit = 1;
    double best = 1e300;
    for (int sp = 1; sp <= max_split; ++sp) {
      const int wgs = ntiles * sp * G;  // (the whole group's tiles are one launch)
      const int slots = h->cus * (sizeof(T) == 4 ? BLR_GRAM_WGS : 2);
      const int rounds = (wgs + slots - 1) / slots;
      const int cols_sp = ((N + sp - 1) / sp + nstage_cols - 1) / nstage_cols * nstage_cols;
      const double cost = (double)rounds * (cols_sp + 256.0);
      if (cost < best) { best = cost; nsplit = sp; }
    }
    // Diagonal macro tiles cost less per column than off-diagonal ones in the ring loop (only the 36 tiles of 16 x 16 on or below
    // the diagonal are computed: 10 MFMAs per k-step on the critical waves instead of 16; measured ~11.5 with the b partials
    // riding along), so in a single-round launch they get their OWN split factor: fewer, longer column ranges, and the
    // workgroup slots that frees go to the off-diagonal tiles.  Chosen so that the longest workgroup is shortest
    // (c3: 36 x 14 -> 28 x 15 + 8 x 11, i.e. 16 N / 14 -> 16 N / 15 per workgroup on 508 of 512 slots: 0.963 -> 0.935 ms;
    // tools/scan_splits.sh).  Multi-round launches (c5: 136 tiles x 15) keep one factor: there the dispatcher balances.
    // (on the bf16 matrix cores -- gram_tile_kernel<float, true> -- a diagonal tile is 18 matrix instructions and up to four operand
    // builds per half against 24 and four: the builds dominate and the tiles cost nearly the same per column; tools/scan_splits.sh at
    // c3: 15 | 11 0.80 ms, 14 | 14 0.75, 14 | 12 and 13 | 13 0.765)
    const double kDiagCost = bf3 ? 15.5 : 11.5;
    nsplit_diag = 0;  // 0: one factor for all tiles
    nlong = 0;        // strictly lower tiles with one column range less (multi-round launches)
    {
      const int slots = h->cus * (sizeof(T) == 4 ? BLR_GRAM_WGS : 2);
      // (only where the diagonal tiles will go through the ring loop: f32, LDS-DMA staging, whole row blocks)
      const T* X0 = a.X + reg0 * a.strideX;
      const bool dealt = sizeof(T) == 4 && a.layout == LAYOUT_COLVECS && ((uintptr_t)X0 % 16 == 0) &&
                         ((a.ldx * (int64_t)sizeof(T)) % 16 == 0) && !h->opt.no_gram_ring &&
                         !h->opt.no_diag_split && D % kPB == 0 && NC >= 2 && (ntiles * nsplit * G <= slots || h->opt.gs_fields >= 2);
      if (h->opt.gs_fields >= 2) {  // "off-diagonal,diagonal" (equal: one factor): measurements only
        const int so = h->opt.gs_so, sd = h->opt.gs_sd, nl = h->opt.gs_nl, nf = h->opt.gs_fields;
        if (nf >= 2 && so >= 1 && sd >= 1 && sd <= so && so <= max_split) {
          nsplit = so;
          nsplit_diag = ((sd < so || nf == 3) && dealt) ? sd : 0;
          if (nf == 3 && nsplit_diag > 0 && so >= 2) nlong = std::max(0, std::min(nl, ntiles - NC));
        }
      } else if (dealt) {
        auto cols = [&](int sp) { return (double)(((N + sp - 1) / sp + nstage_cols - 1) / nstage_cols * nstage_cols); };
        const int n_off = ntiles - NC;
        double best_t = 16.0 * cols(nsplit);  // today's longest workgroup (diagonal tiles shorter, off-diagonal ones set the time)
        int bo = 0, bd = 0;
        for (int so = 1; so <= max_split; ++so)
          for (int sd = 1; sd <= so; ++sd) {
            if ((n_off * so + NC * sd) * G > slots) break;
            const double t = std::max(16.0 * cols(so), kDiagCost * cols(sd)) * (1.0 + 0.002 * so);
            if (t < best_t * 0.995) { best_t = t; bo = so; bd = sd; }
          }
        if (bo > 0) { nsplit = bo; nsplit_diag = bd; }
      }
      // multi-round launch of one regressor: three kinds of work items (plan_gram_rounds)
      const bool ring_ok = sizeof(T) == 4 && a.layout == LAYOUT_COLVECS && ((uintptr_t)X0 % 16 == 0) &&
                           ((a.ldx * (int64_t)sizeof(T)) % 16 == 0) && !h->opt.no_gram_ring &&
                           !h->opt.no_diag_split && h->opt.gs_fields < 2 && D % kPB == 0 && NC >= 2;
      if (ring_ok && G == 1 && nsplit_diag == 0 && ntiles * nsplit > slots && nsplit >= 2) {
        const std::array<int, 3> key{NC, N, slots};
        auto it = h->gram_plans.find(key);
        if (it == h->gram_plans.end()) {
          const GramPlan pl = plan_gram_rounds(ntiles - NC, NC, nsplit, N, nstage_cols, slots, max_split, kDiagCost / 16.0);
          it = h->gram_plans.emplace(key, std::array<int, 3>{pl.so, pl.sd, pl.nlong}).first;
          if (h->opt.plan_debug)
            fprintf(stderr, "blr: Gram plan for %d row blocks, N = %d: one factor %d -> ranges %d (off-diagonal, %d tiles with %d) / %d (diagonal), "
                            "modelled makespan %.0f column units\n", NC, N, nsplit, pl.so, pl.nlong, pl.so - 1, pl.sd, pl.makespan);
        }
        if (it->second[1] > 0) { nsplit = it->second[0]; nsplit_diag = it->second[1]; nlong = it->second[2]; }
      }
    }
  };
  // the planes path: one 512-thread workgroup per CU, k-blocks of 16 columns, diagonal macro tiles as long as the others
  const bool planes4 = NP == 2 && !h->opt.planes8;  // 64 x 64 per wave, two 256-thread workgroups per CU (gram_planes4_kernel)
  const int kbs = NP == 2 ? PlanesCfg<2>::KBS : PlanesCfg<3>::KBS;
  const int NKB = ((N + 15) / 16 + kbs - 1) / kbs * kbs;  // k-blocks of 16 columns, padded to whole stages of the Gram launch (zero columns)
  auto plan_splits_planes = [&](int G) {
    nsplit_diag = 0; nlong = 0; nsplit = 1;
    double best = 1e300;
    const int slots = h->cus * (planes4 ? 2 : 1);
    for (int sp = 1; sp <= std::min(64, std::max(1, NKB)); ++sp) {
      const int rounds = (ntiles_g * sp * G + slots - 1) / slots;
      const double cost = (double)rounds * (16.0 * ((NKB + sp - 1) / sp) + 192.0);  // (192: a workgroup's pipeline fill and its 64 KB partial tile, in columns)
      if (cost < best) { best = cost; nsplit = sp; }
    }
    if (h->opt.gs_fields >= 2 && h->opt.gs_so >= 1 && h->opt.gs_so <= 64) nsplit = h->opt.gs_so;  // (GRAM_SPLITS: measurements only)
  };
  if (planes) plan_splits_planes(G); else plan_splits(G);
  // column chunks of the planes pass (one b partial each): ~ 2 workgroups per CU, and no chunk beyond kPlanesChunkKb k-blocks (its
  // per-column scalars live in LDS)
  const int nbchunks = planes ? std::max(1, std::min(NKB, std::max((512 + NC - 1) / NC, (NKB + kPlanesChunkKb - 1) / kPlanesChunkKb))) : 0;
  const bool prior_factor = a.prior_kind == PRIOR_UPPER_FACTOR;
  const int pf = prior_factor ? 1 : 0;
  size_t ws_cap = kChainWorkspace;
  if (h->opt.chain_ws_mb > 0) ws_cap = (size_t)h->opt.chain_ws_mb << 20;  // tests: small groups
  {
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    auto per_for = [&](int nsp) {  // the carve below, as a function of the split factor
      const size_t nst = (size_t)(nsp + pf);
      return al((size_t)lda * DP * sizeof(T)) + al(a.prior_kind == PRIOR_DENSE ? (size_t)DP * DP * sizeof(T) : 0) +
             al(planes ? (size_t)NKB * NCA * 4 * NP * 1024 : 0) +
             al(nst * ntiles_g * kPB * kPB * sizeof(T)) + al(std::max<size_t>(nst, (size_t)nbchunks) * NCA * kPB * sizeof(double) + (size_t)DPA * sizeof(unsigned) + 8) + al((size_t)std::max(N, 1) * sizeof(T)) +
             (ms ? al((size_t)std::max(N, 1) * sizeof(T)) + al((size_t)nbchunks * kPB * sizeof(double)) : 0) +
             al(a.noise_kind == NOISE_DIAGONAL ? (size_t)std::max(N, 1) * sizeof(T) : 0) + 2 * al((size_t)1024 * sizeof(double)) +
             al((size_t)DP * DP * sizeof(T)) + al(64);
    };
    const int Gc = (int)std::max<size_t>(1, std::min<size_t>((size_t)G, ws_cap / per_for(nsplit)));
    if (Gc != G) {
      G = Gc;
      if (planes) plan_splits_planes(G); else plan_splits(G);
    }
  }
  const int nsplit_total = nsplit + pf;  // (nsplit_diag <= nsplit: the partial workspace is laid out for the larger factor)
  const int gridc = 1024;

  const int64_t gp_tiles = (int64_t)nsplit_total * ntiles_g;

  // workspace carve
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
  const size_t o_abar = carve((size_t)lda * DP * sizeof(T));
  const size_t o_w = carve(a.prior_kind == PRIOR_DENSE ? (size_t)DP * DP * sizeof(T) : 0);
  const size_t o_xp = carve(planes ? (size_t)NKB * NCA * 4 * NP * 1024 : 0);
  const size_t o_gp = carve((size_t)gp_tiles * kPB * kPB * sizeof(T));
  // b partials, then (planes path) the rows' largest entries: both zeroed by the prior launch's scratch initialisation
  const int bslots = std::max(nsplit_total, nbchunks);
  const size_t o_bp = carve((size_t)bslots * NCA * kPB * sizeof(double) + (size_t)DPA * sizeof(unsigned) + 8);  // (+ the planes pass's redo flag)
  const size_t o_mu = carve(ms ? (size_t)std::max(N, 1) * sizeof(T) : 0);           // multi-output: x_n'mw
  const size_t o_qs = carve(ms ? (size_t)nbchunks * kPB * sizeof(double) : 0);      // multi-output: partial sums of q_s per column chunk
  const size_t o_r = carve((size_t)std::max(N, 1) * sizeof(T));
  const size_t o_wv = carve(a.noise_kind == NOISE_DIAGONAL ? (size_t)std::max(N, 1) * sizeof(T) : 0);  // 1 / s_n for the Gram launch
  const size_t o_q = carve((size_t)gridc * sizeof(double));
  const size_t o_l = carve((size_t)gridc * sizeof(double));
  const size_t o_m = carve((size_t)DP * DP * sizeof(T));  // transposed factor for the back substitution
  const size_t o_sc = carve(64);
  const size_t per = off;  // one regressor's workspace (a multiple of 256 bytes)
  G = (int)std::max<size_t>(1, std::min<size_t>((size_t)G, ws_cap / per));  // (the re-planned split may need a little more per regressor)
  int rc;
  for (;;) {  // (a device too full for the whole group's workspace: smaller groups, down to one regressor at a time)
    rc = ensure_ws(h, per * (size_t)G);
    if (rc == 0 || G == 1) break;
    (void)hipGetLastError();
    h->err.clear();
    G = (G + 1) / 2;
  }
  if (rc) return rc;
  if (G_done) *G_done = G;
  // Every launch below covers the whole group: regressor g from blockIdx.y / .z, its caller-side arrays by the batch strides
  // and its workspace `per` bytes after its predecessor's (the pointers here are regressor reg0's).
  const int64_t reg = reg0;
  char* ws = h->ws;
  T* Abar = reinterpret_cast<T*>(ws + o_abar);
  T* W = reinterpret_cast<T*>(ws + o_w);
  T* Gpart = reinterpret_cast<T*>(ws + o_gp);
  double* bpart = reinterpret_cast<double*>(ws + o_bp);
  T* rvec = reinterpret_cast<T*>(ws + o_r);
  T* wvec = a.noise_kind == NOISE_DIAGONAL ? reinterpret_cast<T*>(ws + o_wv) : nullptr;
  double* qpart = reinterpret_cast<double*>(ws + o_q);
  double* lpart = reinterpret_cast<double*>(ws + o_l);
  double* logdetLw = reinterpret_cast<double*>(ws + o_sc);
  int32_t* info_prior = reinterpret_cast<int32_t*>(ws + o_sc + 8);
  int32_t* info_chol = reinterpret_cast<int32_t*>(ws + o_sc + 12);
  unsigned* info_noise = reinterpret_cast<unsigned*>(ws + o_sc + 16);
  const int64_t wsb = (int64_t)per;                  // byte stride between the regressors' workspaces
  const int64_t wse = (int64_t)(per / sizeof(T));    // the same in elements (per is a multiple of 256)

  const T* X = a.X + reg * a.strideX;
  const T* y = a.y + reg * a.stridey;
  const T* s = a.s + reg * a.strides;
  const T* mw = a.mw + reg * a.stridemw;
  const T* Lw = a.Lw + reg * a.strideLw;

  // ---- prior: SPD check + logdet (reference :78).  One launch clears the scratch words, sets the noise flag and zeroes the
  // b partials; for a diagonal / factor prior it also checks the diagonal and seeds info_chol with the prior's status (a failed
  // prior short-circuits the factorisation), for a dense one the blocked factorisation of W does that.
  {
    const bool dense = a.prior_kind == PRIOR_DENSE;
    ScratchInit init;
    init.words16 = reinterpret_cast<unsigned*>(ws + o_sc);
    init.ones = info_noise;
    init.zeros = bpart;
    init.nzeros = (long long)bslots * NCA * kPB + DPA / 2 + 1;  // (+ the row maxima behind the b partials: DP words, + the redo flag)
    init.info_copy = dense ? nullptr : info_chol;
    const int gridp = (int)std::min<long long>(64, 1 + init.nzeros / (8 * kThreads));
    // (dense: the kernel's own look at Lw's diagonal is not the answer -- status and logdet go to spare scratch words)
    hipLaunchKernelGGL(prior_diag_kernel<T>, dim3(gridp, G), dim3(kThreads), 0, h->stream, Lw, a.ldl, a.prior_kind, D,
                       dense ? reinterpret_cast<double*>(ws + o_sc + 40) : logdetLw,
                       dense ? reinterpret_cast<int32_t*>(ws + o_sc + 32) : info_prior, init, a.strideLw, wsb);
    if (dense) {
      hipLaunchKernelGGL(prior_copy_kernel<T>, dim3(1024, G), dim3(kThreads), 0, h->stream, Lw, a.ldl, D, DP, W, (int64_t)DP, a.strideLw, wse);
      if ((rc = chol_large<T>(h, W, DP, DP, DP, info_prior, G, wse, (int)(per / sizeof(int32_t))))) return rc;
      hipLaunchKernelGGL(logdet_kernel<T>, dim3(G), dim3(kThreads), 0, h->stream, (const T*)W, (int64_t)DP, D, logdetLw, wsb,
                         (const int32_t*)info_prior, info_chol);
    }
  }

  // ---- column statistics (reference :82-84)
  {
    ColstatsArgs<T> c{};
    c.X = X; c.ldx = a.ldx; c.y = y; c.s = s; c.mw = mw; c.r = rvec; c.w = wvec; c.qpart = qpart; c.lpart = lpart;
    c.noise_info = info_noise;
    c.layout = a.layout; c.noise_kind = a.noise_kind; c.D = D; c.N = N;
    c.grp_X = a.strideX; c.grp_y = a.stridey; c.grp_s = a.strides; c.grp_mw = a.stridemw; c.grp_ws = wsb;
    c.w_sqrt = planes ? 1 : 0;
    c.mu = ms ? reinterpret_cast<T*>(ws + o_mu) : nullptr;
    if (rff) {
      c.X = nullptr;
      c.rff_Xin = a.rff_Xin; c.rff_ldxin = a.rff_ldxin; c.rff_Omega = a.rff_Omega; c.rff_ldo = a.rff_ldo; c.rff_phase = a.rff_phase;
      c.rff_scale = a.rff_scale; c.rff_Din = a.rff_Din;
    }
    size_t lds = (((size_t)D * sizeof(T) + 15) & ~(size_t)15) + 64;
    hipLaunchKernelGGL(colstats_kernel<T>, dim3(gridc, G), dim3(kThreads), lds, h->stream, c);
  }

  // ---- Gram (reference :86) : split-K partial tiles, then the prior factor as pseudo-observations
  if ((rc = set_lds<T>(h, reinterpret_cast<const void*>(gram_tile_kernel<T>), LC::LDS_BYTES))) return rc;
  if constexpr (sizeof(T) == 4) {
    if ((rc = set_lds<T>(h, reinterpret_cast<const void*>(gram_tile_kernel<T, true>), LC::LDS_BYTES))) return rc;
  }
  GramTileArgs<T> g{};
  g.X = X; g.ldx = a.ldx; g.layout = a.layout;
  const bool no_ring = h->opt.no_gram_ring;  // A/B experiments only
  g.use_dma = (a.layout == LAYOUT_COLVECS && ((uintptr_t)X % 16 == 0) && ((a.ldx * (int64_t)sizeof(T)) % 16 == 0)) ? (no_ring ? 2 : 1) : 0;
  g.bf3 = bf3 ? 1 : 0;  // (f32 + the LDS-DMA ring only: the bf16 x 3 code lives in the ring loop)
  g.s = s; g.noise_kind = a.noise_kind; g.r = rvec; g.wpre = wvec;
  g.D = D; g.n_begin = 0; g.n_end = N; g.nblocks = NC; g.bpart = bpart; g.mode_out = 0;
  g.grp_X = a.strideX; g.grp_s = a.strides; g.grp_ws = wsb;
  const bool no_swizzle = h->opt.no_xcd_swizzle;
  ReduceArgs<T> r{};
  r.bpart = bpart; r.nblocks = NC;
  r.Lw = Lw; r.ldl = a.ldl; r.prior_kind = a.prior_kind; r.D = D; r.DP = DP; r.Abar = Abar; r.lda = lda;
  r.Lw_post = a.Lw_post ? a.Lw_post + reg * a.strideLp : nullptr; r.ldlp = a.ldlp;
  r.grp_Lw = a.strideLw; r.grp_Lp = a.strideLp; r.grp_ws = wsb;
  // launches the tiles described by g (+ the prior-factor pseudo split) and their reduction on `st`
  auto gram_tiles = [&](hipStream_t st, int nsp, int nt, T* gp) {
    g.nsplit = nsp; g.ntiles = nt; g.Gpart = gp;
    g.nsplit_diag = nsplit_diag; g.nlong = nlong;
    g.xcd_swizzle = (nsp > 1 && !no_swizzle) ? (nlong == 0 ? 1 : 2) : 0;  // (three kinds of work items: remapped inside a kind, the dispatch order of the kinds IS the plan)
    const int nwg = nsplit_diag ? (nt - NC) * nsp - nlong + NC * nsplit_diag : nt * nsp;
    if constexpr (sizeof(T) == 4) {
      if (g.bf3) hipLaunchKernelGGL((gram_tile_kernel<T, true>), dim3(nwg, G), dim3(kThreads), LC::LDS_BYTES, st, g);
      else hipLaunchKernelGGL(gram_tile_kernel<T>, dim3(nwg, G), dim3(kThreads), LC::LDS_BYTES, st, g);
    } else {
      hipLaunchKernelGGL(gram_tile_kernel<T>, dim3(nwg, G), dim3(kThreads), LC::LDS_BYTES, st, g);
    }
    if (prior_factor) {
      GramTileArgs<T> u = g;
      u.xcd_swizzle = 0;
      u.X = Lw; u.ldx = a.ldl; u.layout = 2; u.use_dma = 0; u.s = nullptr; u.r = nullptr; u.wpre = nullptr;
      u.grp_X = a.strideLw; u.grp_s = 0;
      u.n_begin = 0; u.n_end = D; u.nsplit = 1; u.nsplit_diag = 0; u.nlong = 0;
      u.Gpart = gp + (int64_t)nsp * nt * kPB * kPB;
      u.bpart = bpart + (int64_t)nsp * NC * kPB;
      hipLaunchKernelGGL(gram_tile_kernel<T>, dim3(nt, G), dim3(kThreads), LC::LDS_BYTES, st, u);
    }
  };
  auto gram_reduce = [&](hipStream_t st, int nsp, int nt, T* gp, int reduce_blocks) {
    r.Gpart = gp; r.nsplit_total = nsp + pf; r.ntiles = nt;
    r.nsplit_diag = nsplit_diag; r.pseudo_split = pf; r.nlong = nlong;
    hipLaunchKernelGGL(gram_reduce_kernel<T>, dim3(nt + reduce_blocks, 16, G), dim3(kThreads), 0, st, r);
  };

  g.tile_i0 = 0; g.tile_j0 = 0; g.tri = 1;
  bool planes_done = false;
  if constexpr (sizeof(T) == 4) {
    if (planes) {
      if ((rc = set_lds_once(h, reinterpret_cast<const void*>(gram_planes_kernel<2>), (size_t)PlanesCfg<2>::LDS))) return rc;
      if ((rc = set_lds_once(h, reinterpret_cast<const void*>(gram_planes_kernel<3>), (size_t)PlanesCfg<3>::LDS))) return rc;
      unsigned short* Xp = reinterpret_cast<unsigned short*>(ws + o_xp);
      unsigned* rowmax = reinterpret_cast<unsigned*>(bpart + (int64_t)bslots * NCA * kPB);
      PlanesArgs pa{};
      pa.X = rff ? nullptr : X; pa.ldx = a.ldx;
      pa.Xin = a.rff_Xin; pa.ldxin = a.rff_ldxin; pa.Omega = a.rff_Omega; pa.ldo = a.rff_ldo; pa.phase = a.rff_phase; pa.scale = a.rff_scale; pa.Din = a.rff_Din;
      pa.wsq = wvec; pa.r = rvec; pa.Xp = Xp; pa.bpart = bpart; pa.rowmax = rowmax;
      pa.D = D; pa.N = N; pa.NC = NCA; pa.NKB = NKB; pa.nchunks = nbchunks;
      pa.grp_X = a.strideX; pa.grp_ws = wsb;
      if constexpr (sizeof(T) == 4) {
        if (ms) {
          pa.Y = static_cast<const float*>(ms->Y); pa.ldY = ms->ldY; pa.S = ms->S;
          pa.mu = reinterpret_cast<const float*>(ws + o_mu); pa.qsp = reinterpret_cast<double*>(ws + o_qs);
        }
      }
      // a basis of up to 8 input dimensions: the raw inputs of a workgroup's whole column chunk are staged once (blr_planes.hpp, xs_chunk)
      pa.xs_chunk = (rff && a.rff_Din <= 8) ? 1 : 0;
      const size_t plds = (size_t)(3 * 16 * kPlanesChunkKb + (rff ? (pa.xs_chunk ? 8 * 16 * kPlanesChunkKb : a.rff_Din * 16) : 0)) * sizeof(float);
      if (NP == 2) {  // the rows' power-of-two scales need (a bound of) the rows' largest entries first
        if (rff) {  // a basis: its bound
          hipLaunchKernelGGL(rowmax_kernel<true>, dim3(nbchunks, 1, G), dim3(kThreads), 0, h->stream, pa);
          hipLaunchKernelGGL((planes_kernel<2, true>), dim3(nbchunks, NC, G), dim3(kThreads), plds, h->stream, pa);
        } else if (h->opt.no_spec_rowmax) {  // the exact maxima: one more pass over X
          hipLaunchKernelGGL(rowmax_kernel<false>, dim3(nbchunks, NCA, G), dim3(kThreads), 0, h->stream, pa);
          hipLaunchKernelGGL((planes_kernel<2, false>), dim3(nbchunks, NCA, G), dim3(kThreads), plds, h->stream, pa);
        } else {
          // sampled maxima with head-room, the planes pass checks that every entry fits; the exact pass + the planes again only if one
          // did not (two launches that return at once otherwise: ~ 5 us against the 58 us of the exact pass at config 3)
          if ((rc = ensure_stats(h))) return rc;
          pa.redo = rowmax + DPA;
          pa.redo_total = h->stats_dev + 1;
          pa.sample_kb = 2;
          hipLaunchKernelGGL(rowmax_kernel<false>, dim3(nbchunks, NCA, G), dim3(kThreads), 0, h->stream, pa);
          hipLaunchKernelGGL((planes_kernel<2, false>), dim3(nbchunks, NCA, G), dim3(kThreads), plds, h->stream, pa);
          pa.sample_kb = 0;
          pa.redo_pass = 1;
          hipLaunchKernelGGL(rowmax_kernel<false>, dim3(nbchunks, NCA, G), dim3(kThreads), 0, h->stream, pa);
          hipLaunchKernelGGL((planes_kernel<2, false>), dim3(nbchunks, NCA, G), dim3(kThreads), plds, h->stream, pa);
        }
      } else {
        if (rff) hipLaunchKernelGGL((planes_kernel<3, true>), dim3(nbchunks, NC, G), dim3(kThreads), plds, h->stream, pa);
        else hipLaunchKernelGGL((planes_kernel<3, false>), dim3(nbchunks, NC, G), dim3(kThreads), plds, h->stream, pa);
      }
      GramPlanesArgs ga{};
      ga.Xp = Xp; ga.NC = NCA; ga.NKB = NKB; ga.Gpart = Gpart; ga.ntiles = ntiles_g; ga.nsplit = nsplit;
      ga.s_iso = a.noise_kind == NOISE_DIAGONAL ? nullptr : s;
      ga.rowmax = rowmax;
      ga.xcd_swizzle = (nsplit > 1 && !no_swizzle) ? 1 : 0;
      ga.grp_ws = wsb; ga.grp_s = a.strides;
      if (planes4) {
        if ((rc = set_lds_once(h, reinterpret_cast<const void*>(gram_planes4_kernel), (size_t)Planes4Cfg::LDS))) return rc;
        hipLaunchKernelGGL(gram_planes4_kernel, dim3(ntiles_g * nsplit, G), dim3(kThreads), (size_t)Planes4Cfg::LDS, h->stream, ga);
      } else if (NP == 2) hipLaunchKernelGGL(gram_planes_kernel<2>, dim3(ntiles_g * nsplit, G), dim3(kPlanesThreads), (size_t)PlanesCfg<2>::LDS, h->stream, ga);
      else hipLaunchKernelGGL(gram_planes_kernel<3>, dim3(ntiles_g * nsplit, G), dim3(kPlanesThreads), (size_t)PlanesCfg<3>::LDS, h->stream, ga);
      if (prior_factor) {  // the prior factor as pseudo-observations: one more partial per tile, from the f32 kernel
        GramTileArgs<T> u = g;
        u.nsplit = 1; u.ntiles = ntiles; u.nsplit_diag = 0; u.nlong = 0;
        u.xcd_swizzle = 0;
        u.X = Lw; u.ldx = a.ldl; u.layout = 2; u.use_dma = 0; u.bf3 = 0; u.s = nullptr; u.r = nullptr; u.wpre = nullptr;
        u.grp_X = a.strideLw; u.grp_s = 0;
        u.n_begin = 0; u.n_end = D;
        u.Gpart = Gpart + (int64_t)nsplit * ntiles * kPB * kPB;
        u.bpart = bpart;  // (never written: r == NULL)
        hipLaunchKernelGGL(gram_tile_kernel<T>, dim3(ntiles, G), dim3(kThreads), LC::LDS_BYTES, h->stream, u);
      }
      r.nsplit_b = nbchunks;
      r.nblocks = NCA;                 // (stride of the planes pass's b partials)
      r.multi_block = ms ? NC : 0;
      gram_reduce(h->stream, nsplit, ntiles_g, Gpart, NC);
      planes_done = true;
    }
  }
  if (!planes_done) {
    gram_tiles(h->stream, nsplit, ntiles, Gpart);
    gram_reduce(h->stream, nsplit, ntiles, Gpart, NC);
  }
  // ---- blocked Cholesky of Abar (rows DP.. = rhs) -> L, u  (reference :86, :57)
  // (only the first 64 of the 128 padding rows ride along: row DP is b', the others are zero and nobody reads them back --
  // half the right-hand-side sub-tiles of every trailing update, and c5's first trailing updates fit one round)
  // (multi-output: rows DP + s are b_s' for the columns s < S of Y; more than 64 of them take all 128 rows along)
  if ((rc = chol_large<T>(h, Abar, lda, DP, DP + ((ms && ms->S > TrailCfg<T>::SB) ? kPB : TrailCfg<T>::SB), info_chol, G, wse,
                          (int)(per / sizeof(int32_t))))) return rc;
  if (ms) {
    ms->Abar = Abar; ms->lda = lda; ms->Tfull = ws + o_m; ms->DP = DP;
    ms->qsp = reinterpret_cast<double*>(ws + o_qs); ms->nq = nbchunks; ms->done = true;
  }

  // ---- T = L' (for the caller and for the AXPY-form back substitution), then m, posterior mean, evidence: one launch each
  // over the group (blockIdx.z / WaveSolveArgs::group)
  {
    char* ws = h->ws;
    T* Abar = reinterpret_cast<T*>(ws + o_abar);
    T* Tfull = reinterpret_cast<T*>(ws + o_m);
    // logpdf alone (no posterior mean, no factor wanted): the evidence is complete with the factorisation -- no transpose, no
    // back substitution, the launch below only assembles the scalars
    const bool evidence_only = a.mw_post == nullptr && a.T_post == nullptr;
    if (!evidence_only) {
      dim3 grid((DP + 31) / 32, (DP + 31) / 32, G);
      hipLaunchKernelGGL(transpose_out_kernel<T>, grid, dim3(kThreads), 0, h->stream, (const T*)Abar, lda, DP, Tfull,
                         (int64_t)DP, a.T_post ? a.T_post + reg0 * a.strideT : (T*)nullptr, a.ldt, D, (int64_t)(per / sizeof(T)), a.strideT,
                         (const int32_t*)info_prior, (const unsigned*)info_noise, (const int32_t*)info_chol, (int64_t)per);
    }
    WaveSolveArgs<T> b{};
    b.Tf = Tfull; b.ldtf = DP; b.D = D; b.DP = DP;
    if (evidence_only) { b.Tf = Abar; b.ldtf = lda; b.evidence_only = 1; }  // (diagonal of L = diagonal of T)
    b.rhs = Abar + DP; b.ldrhs = G > 1 ? (int64_t)(per / sizeof(T)) : 0; b.rhs_inc = lda;  // u = row DP of the factored Abar
    b.add = a.mw + reg0 * a.stridemw; b.out = a.mw_post ? a.mw_post + reg0 * a.stride_mwpost : nullptr;
    b.ldout = G > 1 ? a.stride_mwpost : 0;
    b.qpart = reinterpret_cast<double*>(ws + o_q); b.lpart = reinterpret_cast<double*>(ws + o_l); b.nparts = gridc;
    b.logdet_Lw_dev = reinterpret_cast<double*>(ws + o_sc);
    b.noise_kind = a.noise_kind; b.s = a.s + reg0 * a.strides; b.N = N;
    b.logpdf = a.logpdf ? a.logpdf + reg0 : nullptr; b.info = a.info + reg0;
    b.chol_info = reinterpret_cast<int32_t*>(ws + o_sc + 12);
    b.prior_info = reinterpret_cast<int32_t*>(ws + o_sc + 8); b.noise_info = reinterpret_cast<unsigned*>(ws + o_sc + 16);
    if (G > 1) { b.group = G; b.ws_stride = (int64_t)per; b.add_stride = a.stridemw; b.s_stride = a.strides; }
    if ((rc = launch_wave_solve<T>(h, b, NC, G))) return rc;
  }
  HIP_TRY(h, hipGetLastError());
  return 0;
}

template <typename T>
int posterior_large_one(blr_handle* h, const PosteriorArgs<T>& a, int64_t reg) { return posterior_large_group<T>(h, a, reg, 1); }

template <typename T>
int dispatch_posterior(blr_handle* h, const PosteriorArgs<T>& a) {
  if (a.D <= kMaxSmallD) return dispatch_fused_small<T>(h, a);
  // regressors of a batch go through the blocked factorisation in groups (posterior_large_group); the regressors of a group
  // must see the same alignment of X (one split / staging decision per group)
  // (the more the better wherever measured -- D = 256 .. 4096, groups of up to 128, tools/chain_batch_scan.sh -- so the bound is
  // the workspace: kChainWorkspace)
  int gmax = kChainBatchMax;
  if (h->opt.chain_batch > 0) gmax = std::min(kChainBatchMax, h->opt.chain_batch);  // measurements only
  if ((a.strideX * (int64_t)sizeof(T)) % 16 != 0) gmax = 1;
  {  // even groups: 129 regressors run as 65 + 64, not 128 + 1
    const int64_t ngroups = std::max<int64_t>(1, (a.B + gmax - 1) / gmax);
    gmax = (int)((a.B + ngroups - 1) / ngroups);
  }
  for (int64_t reg = 0; reg < a.B;) {
    int done = 1;
    int rc = posterior_large_group<T>(h, a, reg, (int)std::min<int64_t>(gmax, a.B - reg), &done);
    if (rc) return rc;
    reg += done;
  }
  return 0;
}

template <typename T>
int posterior_batched(blr_handle* h, int memspace, int layout, int64_t B, int64_t D, int64_t N, const T* X,
                      int64_t ldx, int64_t strideX, const T* y, int64_t stridey, int noise_kind, const T* s,
                      int64_t strides, int prior_kind, const T* mw, int64_t stridemw, const T* Lw, int64_t ldl,
                      int64_t strideLw, T* mw_post, int64_t stride_mwpost, T* T_post, int64_t ldt, int64_t strideT,
                      T* Lw_post, int64_t ldlp, int64_t strideLp, double* logpdf, int32_t* info) {
  if (!h) return -1;
  h->err.clear();
  if (memspace != BLR_MEM_HOST && memspace != BLR_MEM_DEVICE) return bad_arg(h, 2, "memspace");
  if (layout != BLR_LAYOUT_COLVECS && layout != BLR_LAYOUT_ROWVECS) return bad_arg(h, 3, "unknown layout (reference :26-31)");
  if (B < 0 || B > (1 << 30)) return bad_arg(h, 4, "B out of range (0..2^30)");
  if (D < 1) return bad_arg(h, 5, "D < 1");
  if (D > kMaxLargeD) return bad_arg(h, 5, "D > 8192 is not supported by this build");
  if (N < 0 || N > (1 << 30)) return bad_arg(h, 6, "N out of range");
  if (B == 0) return 0;
  if (N > 0 && !X) return bad_arg(h, 7, "X is NULL");
  if (layout == BLR_LAYOUT_COLVECS ? ldx < D : ldx < std::max<int64_t>(N, 1)) return bad_arg(h, 8, "ldx too small");
  if (strideX < 0) return bad_arg(h, 9, "strideX < 0");
  if (N > 0 && !y) return bad_arg(h, 10, "y is NULL (reference :74 length check)");
  if (stridey < 0) return bad_arg(h, 11, "stridey < 0");
  if (noise_kind != BLR_NOISE_ISOTROPIC && noise_kind != BLR_NOISE_DIAGONAL)
    return bad_arg(h, 12, "noise_kind (dense Sigma_y is outside the GPU scope)");
  if (!s) return bad_arg(h, 13, "s is NULL");
  if (strides < 0) return bad_arg(h, 14, "strides < 0");
  if (prior_kind != BLR_PRIOR_DENSE && prior_kind != BLR_PRIOR_UPPER_FACTOR && prior_kind != BLR_PRIOR_DIAGONAL)
    return bad_arg(h, 15, "prior_kind");
  if (!mw) return bad_arg(h, 16, "mw is NULL");
  if (stridemw < 0) return bad_arg(h, 17, "stridemw < 0");
  if (!Lw) return bad_arg(h, 18, "Lw is NULL");
  if (prior_kind != BLR_PRIOR_DIAGONAL && ldl < D) return bad_arg(h, 19, "ldl < D");
  if (strideLw < 0) return bad_arg(h, 20, "strideLw < 0");
  if (mw_post && B > 1 && stride_mwpost < D) return bad_arg(h, 22, "stride_mwpost < D");
  if (T_post && ldt < D) return bad_arg(h, 24, "ldt < D");
  if (T_post && B > 1 && strideT < (int64_t)mat_extent(D, D, ldt)) return bad_arg(h, 25, "strideT too small");
  if (Lw_post && ldlp < D) return bad_arg(h, 27, "ldlp < D");
  if (Lw_post && B > 1 && strideLp < (int64_t)mat_extent(D, D, ldlp)) return bad_arg(h, 28, "strideLp too small");
  if (!info) return bad_arg(h, 30, "info is NULL");

  HIP_TRY(h, hipSetDevice(h->device));
  PosteriorArgs<T> a{};
  a.ldx = ldx; a.strideX = strideX; a.stridey = stridey; a.strides = strides; a.stridemw = stridemw;
  a.ldl = ldl; a.strideLw = strideLw; a.stride_mwpost = stride_mwpost; a.ldt = ldt; a.strideT = strideT;
  a.ldlp = ldlp; a.strideLp = strideLp;
  a.layout = layout; a.noise_kind = noise_kind; a.prior_kind = prior_kind;
  a.D = (int)D; a.N = (int)N; a.B = (int)B;

  if (memspace == BLR_MEM_DEVICE) {
    a.X = X; a.y = y; a.s = s; a.mw = mw; a.Lw = Lw;
    a.mw_post = mw_post; a.T_post = T_post; a.Lw_post = Lw_post; a.logpdf = logpdf; a.info = info;
    a.vec_ok = (layout == BLR_LAYOUT_COLVECS && D % Mfma<T>::VEC == 0 && aligned16(X, ldx, strideX)) ? 1 : 0;
    if (h->rff_src) {
      a.rff_Xin = static_cast<const T*>(h->rff_src->Xin); a.rff_ldxin = h->rff_src->ldxin;
      a.rff_Omega = static_cast<const T*>(h->rff_src->Omega); a.rff_ldo = h->rff_src->ldo;
      a.rff_phase = static_cast<const T*>(h->rff_src->phase); a.rff_scale = (T)h->rff_src->scale; a.rff_Din = h->rff_src->Din;
    }
    int rc = dispatch_posterior<T>(h, a);
    if (rc) return rc;
    if (!h->async) HIP_TRY(h, hipStreamSynchronize(h->stream));
    return 0;
  }

  // host pointers: stage through device copies
  Staging guard(h);
  const size_t x_one = layout == BLR_LAYOUT_COLVECS ? mat_extent(D, N, ldx) : mat_extent(N, D, ldx);
  const size_t lw_one = prior_kind == BLR_PRIOR_DIAGONAL ? (size_t)D : mat_extent(D, D, ldl);
  const size_t s_one = noise_kind == BLR_NOISE_DIAGONAL ? (size_t)N : 1;
  int rc;
  if ((rc = stage_in(h, X, extent(B, strideX, x_one), &a.X))) return rc;
  if ((rc = stage_in(h, y, extent(B, stridey, (size_t)N), &a.y))) return rc;
  if ((rc = stage_in(h, s, extent(B, strides, s_one), &a.s))) return rc;
  if ((rc = stage_in(h, mw, extent(B, stridemw, (size_t)D), &a.mw))) return rc;
  if ((rc = stage_in(h, Lw, extent(B, strideLw, lw_one), &a.Lw))) return rc;
  const size_t n_mw = extent(B, stride_mwpost, (size_t)D);
  const size_t n_T = extent(B, strideT, mat_extent(D, D, ldt));
  const size_t n_Lp = extent(B, strideLp, mat_extent(D, D, ldlp));
  if ((rc = stage_out_alloc(h, mw_post, n_mw, &a.mw_post))) return rc;
  if ((rc = stage_out_alloc(h, T_post, n_T, &a.T_post))) return rc;
  if ((rc = stage_out_alloc(h, Lw_post, n_Lp, &a.Lw_post))) return rc;
  if ((rc = stage_out_alloc(h, logpdf, (size_t)B, &a.logpdf))) return rc;
  if ((rc = stage_out_alloc(h, info, (size_t)B, &a.info))) return rc;
  if (N == 0) {  // no data: the kernel never dereferences X / y, but keep the pointers valid
    if (!a.X) a.X = a.mw;
    if (!a.y) a.y = a.mw;
  }
  a.vec_ok = (layout == BLR_LAYOUT_COLVECS && D % Mfma<T>::VEC == 0 && aligned16(a.X, ldx, strideX)) ? 1 : 0;
  if ((rc = dispatch_posterior<T>(h, a))) return rc;
  if (mw_post) HIP_TRY(h, hipMemcpyAsync(mw_post, a.mw_post, n_mw * sizeof(T), hipMemcpyDeviceToHost, h->stream));
  if (T_post) HIP_TRY(h, hipMemcpyAsync(T_post, a.T_post, n_T * sizeof(T), hipMemcpyDeviceToHost, h->stream));
  if (Lw_post) HIP_TRY(h, hipMemcpyAsync(Lw_post, a.Lw_post, n_Lp * sizeof(T), hipMemcpyDeviceToHost, h->stream));
  if (logpdf) HIP_TRY(h, hipMemcpyAsync(logpdf, a.logpdf, (size_t)B * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipMemcpyAsync(info, a.info, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return 0;
}

template <typename T>
int posterior_single(blr_handle* h, int layout, int64_t D, int64_t N, const T* X, int64_t ldx, const T* y,
                     int noise_kind, const T* s, int prior_kind, const T* mw, const T* Lw, int64_t ldl, T* mw_post,
                     T* T_post, int64_t ldt, T* Lw_post, int64_t ldlp, double* logpdf) {
  int32_t info = 0;
  int rc = posterior_batched<T>(h, BLR_MEM_HOST, layout, 1, D, N, X, ldx, 0, y, 0, noise_kind, s, 0, prior_kind, mw, 0,
                                Lw, ldl, 0, mw_post, D, T_post, ldt, ldt * D, Lw_post, ldlp, ldlp * D, logpdf, &info);
  if (rc) {
    // re-number argument positions of the batched form onto the single form
    static const int map[31] = {0, 1, 0, 2, 0, 3, 4, 5, 6, 0, 7, 0, 8, 9, 0, 10, 11, 0, 12, 13, 0,
                                14, 0, 15, 16, 0, 17, 18, 0, 19, 0};
    if (rc < 0 && rc > -31 && map[-rc]) return -map[-rc];
    return rc;
  }
  return info;
}

// ---- factor of the prior precision for marginals / draws ----------------------------------------------
// Returns a device pointer to U (upper factor, ld = D, stride D*D) or to d (diagonal) per regressor.
template <typename T>
int prior_factor(blr_handle* h, int64_t B, int64_t D, int prior_kind, const T* Lw_dev, int64_t ldl, int64_t strideLw,
                 const T** U, int64_t* ldu, int64_t* strideU, int* kind_out, int32_t** info_dev) {
  *info_dev = nullptr;
  if (prior_kind != BLR_PRIOR_DENSE) {
    *U = Lw_dev; *ldu = ldl; *strideU = strideLw; *kind_out = prior_kind;
    return 0;
  }
  size_t need = (size_t)B * D * D * sizeof(T) + (size_t)B * sizeof(int32_t) + 64;
  int rc = ensure_ws(h, need);
  if (rc) return rc;
  T* Uw = reinterpret_cast<T*>(h->ws);
  int32_t* inf = reinterpret_cast<int32_t*>(h->ws + (((size_t)B * D * D * sizeof(T) + 15) & ~(size_t)15));
  size_t lds = ((size_t)D * (D + 1) / 2 + D) * sizeof(T) + 16;
  auto kern = chol_small_kernel<T>;
  { const int rc_lds = set_lds_once(h, reinterpret_cast<const void*>(kern), (size_t)((int)lds)); if (rc_lds) return rc_lds; }
  hipLaunchKernelGGL(kern, dim3((unsigned)std::min<int64_t>(B, 1 << 20)), dim3(kThreads), lds, h->stream, Lw_dev, ldl,
                     strideLw, Uw, D, D * D, inf, (int)D, (int)B);
  HIP_TRY(h, hipGetLastError());
  *U = Uw; *ldu = D; *strideU = D * D; *kind_out = BLR_PRIOR_UPPER_FACTOR; *info_dev = inf;
  return 0;
}


// ---- large-D marginals with a factor or a dense prior: block forward substitution on LDS-resident tiles of 32 inputs ---------------
// (marg_blocksub_kernel: X is read once, nothing but the outputs is written).  A whole batch per launch (blockIdx.y = regressor;
// a dense prior's blocked Cholesky takes the batch through chol_large's groups).  Needs the tile in LDS (D <= 1152 in fp32, 576
// in fp64) and 16-byte loads along d from X and U.  Returns 1 when the call does not qualify (the caller takes the tall-matrix route).
template <typename T>
int marginals_large_group(blr_handle* h, int layout, int64_t B, int64_t D, int64_t N, const T* X, int64_t ldx, int64_t strideX,
                          int noise_kind, const T* s, int64_t strides, int prior_kind, const T* mw, int64_t stridemw, const T* Lw,
                          int64_t ldl, int64_t strideLw, T* mean, int64_t stridemean, T* var, int64_t stridevar, int32_t* info_dev) {
  using MB = MargBlockCfg<T>;
  using MG = MargGemmCfg<T>;
  constexpr int VEC = Mfma<T>::VEC;
  const int DP = (int)((D + kPB - 1) / kPB * kPB), NC = DP / kPB;
  const bool u_given = prior_kind == BLR_PRIOR_UPPER_FACTOR;
  auto vec_ok = [&](const T* p, int64_t ld, int64_t stride) {
    return ld % VEC == 0 && ((uintptr_t)p % 16) == 0 && (B == 1 || (stride * (int64_t)sizeof(T)) % 16 == 0);
  };
  if (!var || prior_kind == BLR_PRIOR_DIAGONAL || h->opt.no_marg_gemm || layout != BLR_LAYOUT_COLVECS || D % 16 != 0 || N < 1 ||
      MB::lds_bytes(DP) > MB::kMaxLds || !vec_ok(X, ldx, strideX) || (u_given && !vec_ok(Lw, ldl, strideLw)))
    return 1;
  int rc;
  if ((rc = set_lds_once(h, reinterpret_cast<const void*>(marg_image_kernel<T>), (size_t)TrsmCfg<T>::LDS_BYTES))) return rc;
  // Tile height: 32 inputs on eight waves, one workgroup per CU -- or, when such tiles leave CUs idle (short N), 16 inputs on four
  // waves, two workgroups per CU (twice the factor traffic per input: 1.02 against 0.77 ms at D = 1024, N = 65536, but 0.081
  // against 0.104 ms at N = 999; fp32 only: the fp64 instance does not fit the registers of two workgroups per CU)
  using MB16 = MargBlockCfg<T, 16>;
  const bool small_tiles = sizeof(T) == 4 && ((N + MB::RT - 1) / MB::RT) * B < h->cus;
  if ((rc = set_lds_once(h, reinterpret_cast<const void*>(marg_blocksub_kernel<T, 32>), (size_t)MB::kMaxLds))) return rc;
  if constexpr (sizeof(T) == 4) {
    if (small_tiles && (rc = set_lds_once(h, reinterpret_cast<const void*>(marg_blocksub_kernel<T, 16>), (size_t)MB16::lds_bytes(DP)))) return rc;
  }
  // regressors per launch: grid.y, the images (36 / 72 KB per diagonal block) within 256 MiB, a dense prior's two D x D copies
  // within 1 GiB and chol_large's group size
  const size_t mat = (((size_t)DP * DP * sizeof(T)) + 255) & ~(size_t)255;
  int64_t chunk = std::min<int64_t>(B, std::min<int64_t>(65535, ((int64_t)256 << 20) / ((int64_t)NC * MG::IMG_ELEMS * (int64_t)sizeof(T))));
  if (!u_given) chunk = std::min<int64_t>(chunk, std::min<int64_t>(kChainBatchMax, std::max<int64_t>(1, ((int64_t)1 << 30) / (int64_t)(2 * mat))));
  if ((rc = ensure_aux(h, (size_t)chunk * NC * MG::IMG_ELEMS * sizeof(T)))) return rc;
  if (!u_given && (rc = ensure_ws(h, (size_t)chunk * 2 * mat + 256))) return rc;
  T* const img = reinterpret_cast<T*>(h->aux);
  HIP_TRY(h, hipMemsetAsync(info_dev, 0, (size_t)B * sizeof(int32_t), h->stream));
  for (int64_t b0 = 0; b0 < B; b0 += chunk) {
    const int nb = (int)std::min<int64_t>(chunk, B - b0);
    const T* U = Lw + b0 * strideLw;
    int64_t ldu = ldl, strideU = strideLw;
    int32_t* const inf = info_dev + b0;
    if (u_given) {  // a factor with a non-positive diagonal entry is not a Cholesky factor: LAPACK-style index
      hipLaunchKernelGGL(prior_diag_kernel<T>, dim3(1, nb), dim3(kThreads), 0, h->stream, U, ldl, (int)PRIOR_UPPER_FACTOR, (int)D,
                         (double*)nullptr, inf, ScratchInit(), strideLw, (int64_t)sizeof(int32_t));
    } else {  // dense precision: L = chol(Lw) (reference :41), then U = L' with the contraction index contiguous
      T* Lf = reinterpret_cast<T*>(h->ws);
      T* Ut = reinterpret_cast<T*>(h->ws + (size_t)chunk * mat);
      const int64_t mstride = (int64_t)(mat / sizeof(T));
      hipLaunchKernelGGL(prior_copy_kernel<T>, dim3(256, nb), dim3(kThreads), 0, h->stream, Lw + b0 * strideLw, ldl, (int)D, DP, Lf, (int64_t)DP,
                         strideLw, mstride);
      if ((rc = chol_large<T>(h, Lf, DP, DP, DP, inf, nb, mstride, 1))) return rc;  // (status words: one int32 apart)
      dim3 tg((DP + 31) / 32, (DP + 31) / 32, nb);
      hipLaunchKernelGGL(transpose_out_kernel<T>, tg, dim3(kThreads), 0, h->stream, (const T*)Lf, (int64_t)DP, DP, Ut, (int64_t)DP,
                         (T*)nullptr, (int64_t)0, 0, mstride);
      U = Ut;
      ldu = DP;
      strideU = mstride;
    }
    const int Df = u_given ? (int)D : DP;  // order of the factor the kernels see (a dense prior's is padded with a unit diagonal)
    hipLaunchKernelGGL(marg_image_kernel<T>, dim3((unsigned)(NC * nb), 2), dim3(kThreads), TrsmCfg<T>::LDS_BYTES, h->stream, U, ldu,
                       (int64_t)kPB * (ldu + 1), (int)kPB, img, (const int32_t*)inf, 0, Df, NC, strideU);
    MargBlockArgs<T> m{};
    m.X = X + b0 * strideX; m.ldx = ldx; m.U = U; m.ldu = ldu; m.img = img; m.mw = mw + b0 * stridemw; m.s = s + b0 * strides;
    m.noise_kind = noise_kind; m.mean = mean ? mean + b0 * stridemean : nullptr; m.var = var + b0 * stridevar; m.info = inf;
    m.D = Df; m.Dx = (int)D; m.DP = DP; m.N = (int)N;
    m.strideX = strideX; m.strideU = strideU; m.stridemw = stridemw; m.strides = strides; m.stridemean = stridemean; m.stridevar = stridevar;
    if constexpr (sizeof(T) == 4) {
      if (small_tiles) {
        const int64_t ntiles = (N + MB16::RT - 1) / MB16::RT;
        const int64_t gx = std::min<int64_t>(ntiles, std::max<int64_t>(1, (2 * h->cus + nb - 1) / nb));  // two workgroups per CU
        hipLaunchKernelGGL((marg_blocksub_kernel<T, 16>), dim3((unsigned)gx, (unsigned)nb), dim3(MB16::THREADS), (size_t)MB16::lds_bytes(DP),
                           h->stream, m);
      }
    }
    if (!small_tiles) {
      const int64_t ntiles = (N + MB::RT - 1) / MB::RT;
      const int64_t gx = std::min<int64_t>(ntiles, std::max<int64_t>(1, (h->cus + nb - 1) / nb));  // one workgroup per CU
      hipLaunchKernelGGL((marg_blocksub_kernel<T, 32>), dim3((unsigned)gx, (unsigned)nb), dim3(MB::THREADS), (size_t)MB::lds_bytes(DP),
                         h->stream, m);
    }
  }
  HIP_TRY(h, hipGetLastError());
  return 0;
}

// ---- large-D marginals: mean stream + (var) tall TRSM through the factorisation's panel machinery -------------------
template <typename T>
int marginals_large_one(blr_handle* h, int layout, int64_t D, int64_t N, const T* X, int64_t ldx, int noise_kind,
                        const T* s, int prior_kind, const T* mw, const T* Lw, int64_t ldl, T* mean, T* var,
                        int32_t* info_dev) {
  using SC = SmallCfg<T, 8>;
  using TC = TrsmCfg<T>;
  using LC = LargeCfg<T>;
  const int DP = (int)((D + kPB - 1) / kPB * kPB), NC = DP / kPB;
  const int NP = (int)((N + kPB - 1) / kPB * kPB);
  const int64_t ldy = (int64_t)DP + NP;
  int rc;
  HIP_TRY(h, hipMemsetAsync(info_dev, 0, sizeof(int32_t), h->stream));
  const bool need_tall = var && prior_kind != BLR_PRIOR_DIAGONAL;
  T* Ybar = nullptr;
  if (need_tall) {
    const size_t bytes = (((size_t)ldy * DP * sizeof(T) + 255) & ~(size_t)255) + (size_t)NP * sizeof(double);
    if ((rc = ensure_ws(h, bytes + 256))) return rc;
    Ybar = reinterpret_cast<T*>(h->ws);
  }
  {
    MeanFillArgs<T> m{};
    m.X = X; m.ldx = ldx; m.layout = layout; m.mw = mw; m.mean = mean; m.Ybar = Ybar; m.ldy = ldy; m.row0 = DP;
    m.D = (int)D; m.DP = DP; m.N = (int)N;
    constexpr int VEC = Mfma<T>::VEC;
    const bool stream_ok = mean && !Ybar && layout == BLR_LAYOUT_COLVECS && (D % VEC) == 0 && (ldx % VEC) == 0 &&
                           ((uintptr_t)X % 16) == 0;
    if (stream_ok) {  // mean only: a pure GEMV stream
      const size_t lds = ((size_t)D * sizeof(T) + 15) & ~(size_t)15;
      hipLaunchKernelGGL(mean_stream_kernel<T>, dim3(1024), dim3(kThreads), lds, h->stream, X, ldx, mw, mean, (int)D, (int)N);
    } else if (mean || Ybar) {
      hipLaunchKernelGGL(mean_fill_kernel<T>, dim3((unsigned)(NP / 64)), dim3(kThreads), 0, h->stream, m);
    }
  }
  if (var && prior_kind == BLR_PRIOR_DIAGONAL) {
    hipLaunchKernelGGL(var_diag_prior_kernel<T>, dim3(2048), dim3(kThreads), 0, h->stream, X, ldx, layout, Lw, (int)D, (int)N, s,
                       noise_kind, var);
  } else if (var) {
    // top block: L = U' (upper factor given) or chol(Lw) (dense precision, reference :41 _cholesky(Lw))
    if (prior_kind == BLR_PRIOR_UPPER_FACTOR) {
      // a factor with a non-positive diagonal entry is not a Cholesky factor: LAPACK-style index instead of Inf / NaN variances
      hipLaunchKernelGGL(prior_diag_kernel<T>, dim3(1), dim3(kThreads), 0, h->stream, Lw, ldl, (int)PRIOR_UPPER_FACTOR, (int)D,
                         (double*)nullptr, info_dev);
      dim3 grid((DP + 31) / 32, (DP + 31) / 32);
      hipLaunchKernelGGL(factor_transpose_fill_kernel<T>, grid, dim3(kThreads), 0, h->stream, Lw, ldl, (int)D, DP, Ybar, ldy);
    } else {
      hipLaunchKernelGGL(prior_copy_kernel<T>, dim3(1024), dim3(kThreads), 0, h->stream, Lw, ldl, (int)D, DP, Ybar, ldy);
      if ((rc = chol_large<T>(h, Ybar, ldy, DP, DP, info_dev))) return rc;
    }
    if ((rc = set_lds<T>(h, reinterpret_cast<const void*>(trsm_block_kernel<T>), TC::LDS_BYTES))) return rc;
    if ((rc = set_lds<T>(h, reinterpret_cast<const void*>(gram_tile_kernel<T>), LC::LDS_BYTES))) return rc;
    const int nyb = NP / kPB;  // row blocks of the input part
    double* rowsq = reinterpret_cast<double*>(h->ws + (((size_t)ldy * DP * sizeof(T) + 255) & ~(size_t)255));
    for (int p = 0; p < NC; ++p) {
      const int nblk = (NP + TC::RB - 1) / TC::RB;
      RowSqArgs<T> rs{};  // the row sums of squares ride on the TRSM: block p of a row is final after panel p
      rs.acc = rowsq; rs.var = var; rs.s = s; rs.noise_kind = noise_kind; rs.N = (int)N; rs.first = p == 0; rs.last = p == NC - 1;
      hipLaunchKernelGGL(trsm_block_kernel<T>, dim3(nblk), dim3(kThreads), TC::LDS_BYTES, h->stream, Ybar, ldy, p, DP, DP + NP,
                         (const int32_t*)info_dev, rs);
      const int m = NC - 1 - p;
      if (m > 0) {
        GramTileArgs<T> g{};
        g.X = Ybar + (int64_t)p * kPB * ldy; g.ldx = ldy; g.layout = LAYOUT_COLVECS; g.use_dma = 1;
        g.s = nullptr; g.noise_kind = NOISE_ISOTROPIC; g.r = nullptr;
        g.D = DP + NP; g.n_begin = 0; g.n_end = kPB; g.nsplit = 1;
        g.tile_i0 = NC; g.tile_j0 = p + 1; g.tri = 3; g.ntile_rows = nyb; g.ntiles = nyb * m; g.nblocks = NC + nyb;
        g.C = Ybar; g.ldc = ldy; g.mode_out = 1;
        hipLaunchKernelGGL(gram_tile_kernel<T>, dim3(g.ntiles), dim3(kThreads), LC::LDS_BYTES, h->stream, g);
      }
    }
  }
  (void)sizeof(SC);
  HIP_TRY(h, hipGetLastError());
  return 0;
}

template <typename T>
int marginals_batched(blr_handle* h, int memspace, int layout, int64_t B, int64_t D, int64_t N, const T* X, int64_t ldx,
                      int64_t strideX, int noise_kind, const T* s, int64_t strides, int prior_kind, const T* mw,
                      int64_t stridemw, const T* Lw, int64_t ldl, int64_t strideLw, T* mean, int64_t stridemean, T* var,
                      int64_t stridevar, int32_t* info) {
  if (!h) return -1;
  h->err.clear();
  if (memspace != BLR_MEM_HOST && memspace != BLR_MEM_DEVICE) return bad_arg(h, 2, "memspace");
  if (layout != BLR_LAYOUT_COLVECS && layout != BLR_LAYOUT_ROWVECS) return bad_arg(h, 3, "unknown layout (reference :26-31)");
  if (B < 0 || B > (1 << 30)) return bad_arg(h, 4, "B out of range (0..2^30)");
  if (D < 1 || D > kMaxLargeD) return bad_arg(h, 5, "D out of range for this build (1..8192)");
  if (N < 0 || N > (1 << 30)) return bad_arg(h, 6, "N out of range");
  if (B == 0 || N == 0) return 0;
  if (!X) return bad_arg(h, 7, "X is NULL");
  if (layout == BLR_LAYOUT_COLVECS ? ldx < D : ldx < N) return bad_arg(h, 8, "ldx too small");
  if (noise_kind != BLR_NOISE_ISOTROPIC && noise_kind != BLR_NOISE_DIAGONAL) return bad_arg(h, 10, "noise_kind");
  if (var && !s) return bad_arg(h, 11, "s is NULL");
  if (prior_kind != BLR_PRIOR_DENSE && prior_kind != BLR_PRIOR_UPPER_FACTOR && prior_kind != BLR_PRIOR_DIAGONAL)
    return bad_arg(h, 13, "prior_kind");
  if (!mw) return bad_arg(h, 14, "mw is NULL");
  if (var && !Lw) return bad_arg(h, 16, "Lw is NULL");
  if (prior_kind != BLR_PRIOR_DIAGONAL && ldl < D) return bad_arg(h, 17, "ldl < D");
  if (mean && B > 1 && stridemean < N) return bad_arg(h, 20, "stridemean < N");
  if (var && B > 1 && stridevar < N) return bad_arg(h, 22, "stridevar < N");
  if (!info) return bad_arg(h, 23, "info is NULL");
  HIP_TRY(h, hipSetDevice(h->device));

  Staging guard(h);
  MarginalArgs<T> a{};
  a.ldx = ldx; a.strideX = strideX; a.strides = strides; a.stridemw = stridemw;
  a.stridemean = stridemean; a.stridevar = stridevar;
  a.layout = layout; a.noise_kind = noise_kind; a.D = (int)D; a.N = (int)N; a.B = (int)B;
  const T* Lw_dev = Lw;
  int32_t* info_out_dev = info;
  int rc;
  if (memspace == BLR_MEM_HOST) {
    const size_t x_one = layout == BLR_LAYOUT_COLVECS ? mat_extent(D, N, ldx) : mat_extent(N, D, ldx);
    const size_t lw_one = prior_kind == BLR_PRIOR_DIAGONAL ? (size_t)D : mat_extent(D, D, ldl);
    if ((rc = stage_in(h, X, extent(B, strideX, x_one), &a.X))) return rc;
    if ((rc = stage_in(h, s, s ? extent(B, strides, noise_kind == BLR_NOISE_DIAGONAL ? (size_t)N : 1) : 0, &a.s))) return rc;
    if ((rc = stage_in(h, mw, extent(B, stridemw, (size_t)D), &a.mw))) return rc;
    if ((rc = stage_in(h, Lw, Lw ? extent(B, strideLw, lw_one) : 0, &Lw_dev))) return rc;
    if ((rc = stage_out_alloc(h, mean, extent(B, stridemean, (size_t)N), &a.mean))) return rc;
    if ((rc = stage_out_alloc(h, var, extent(B, stridevar, (size_t)N), &a.var))) return rc;
    if ((rc = stage_out_alloc(h, info, (size_t)B, &info_out_dev))) return rc;
  } else {
    a.X = X; a.s = s; a.mw = mw; a.mean = mean; a.var = var;
  }
  if (D > kMaxSmallD) {  // large-D path: the whole batch on LDS-resident tiles if it qualifies, else one regressor at a time
    if (h->opt.chain_batch == 1 && B > 1) {  // measurements only (tools/group_scan.py): the same kernels, one regressor per set of launches
      rc = 0;
      for (int64_t reg = 0; rc == 0 && reg < B; ++reg)
        rc = marginals_large_group<T>(h, layout, 1, D, N, a.X + reg * strideX, ldx, strideX, noise_kind, a.s ? a.s + reg * strides : nullptr, strides,
                                      prior_kind, a.mw + reg * stridemw, stridemw, Lw_dev ? Lw_dev + reg * strideLw : nullptr, ldl, strideLw,
                                      a.mean ? a.mean + reg * stridemean : nullptr, stridemean, a.var ? a.var + reg * stridevar : nullptr, stridevar,
                                      info_out_dev + reg);
    } else {
      rc = marginals_large_group<T>(h, layout, B, D, N, a.X, ldx, strideX, noise_kind, a.s, strides, prior_kind, a.mw, stridemw, Lw_dev, ldl,
                                    strideLw, a.mean, stridemean, a.var, stridevar, info_out_dev);
    }
    if (rc < 0 || rc > 1) return rc;
    for (int64_t reg = 0; rc == 1 && reg < B; ++reg) {
      rc = marginals_large_one<T>(h, layout, D, N, a.X + reg * strideX, ldx, noise_kind, a.s ? a.s + reg * strides : nullptr,
                                  prior_kind, a.mw + reg * stridemw, Lw_dev ? Lw_dev + reg * strideLw : nullptr, ldl,
                                  a.mean ? a.mean + reg * stridemean : nullptr, a.var ? a.var + reg * stridevar : nullptr,
                                  info_out_dev + reg);
      if (rc) return rc;
      rc = 1;
    }
    if (memspace == BLR_MEM_HOST) {
      if (mean) HIP_TRY(h, hipMemcpyAsync(mean, a.mean, extent(B, stridemean, (size_t)N) * sizeof(T), hipMemcpyDeviceToHost, h->stream));
      if (var) HIP_TRY(h, hipMemcpyAsync(var, a.var, extent(B, stridevar, (size_t)N) * sizeof(T), hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(h, hipMemcpyAsync(info, info_out_dev, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(h, hipStreamSynchronize(h->stream));
    } else if (!h->async) {
      HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    return 0;
  }
  int32_t* chol_info = nullptr;
  int kind = prior_kind;
  if (var) {
    if ((rc = prior_factor<T>(h, B, D, prior_kind, Lw_dev, ldl, strideLw, &a.U, &a.ldu, &a.strideU, &kind, &chol_info)))
      return rc;
  } else {
    a.U = Lw_dev; a.ldu = ldl; a.strideU = strideLw;
  }
  a.prior_kind = kind;
  a.info = chol_info;
  if (chol_info) HIP_TRY(h, hipMemcpyAsync(info_out_dev, chol_info, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToDevice, h->stream));
  else HIP_TRY(h, hipMemsetAsync(info_out_dev, 0, (size_t)B * sizeof(int32_t), h->stream));
  // D = 128 with a factor, aligned ColVecs: the triangular inverse once per regressor, then a dependency-free product
  // (blr_marginals.hpp); regressors in chunks whose images fit 256 MiB
  const bool rowv = layout == BLR_LAYOUT_ROWVECS;  // (RowVecs: scalar loads, no alignment to ask for)
  if (var && kind == BLR_PRIOR_UPPER_FACTOR && !h->opt.no_marg_gemm && D == kPB && N >= 64 &&
      (rowv || ((ldx % Mfma<T>::VEC) == 0 && ((uintptr_t)a.X % 16) == 0 && ((strideX * (int64_t)sizeof(T)) % 16) == 0))) {
    using G = MargGemmCfg<T>;
    using TC = TrsmCfg<T>;
    const int64_t chunk = std::min<int64_t>(std::min<int64_t>(B, 65535), ((int64_t)256 << 20) / (G::IMG_ELEMS * (int64_t)sizeof(T)));
    if ((rc = ensure_aux(h, (size_t)chunk * G::IMG_ELEMS * sizeof(T)))) return rc;
    if ((rc = set_lds_once(h, reinterpret_cast<const void*>(marg_image_kernel<T>), (size_t)TC::LDS_BYTES))) return rc;
    void (*const gemm_kern)(MarginalArgs<T>, const T*) = rowv ? marginals_gemm_kernel<T, true> : marginals_gemm_kernel<T, false>;
    if ((rc = set_lds_once(h, reinterpret_cast<const void*>(gemm_kern), (size_t)G::LDS_BYTES))) return rc;
    T* const img = reinterpret_cast<T*>(h->aux);
    const int64_t ntiles = (N + 15) / 16;
    for (int64_t b0 = 0; b0 < B; b0 += chunk) {
      const int64_t nb = std::min<int64_t>(chunk, B - b0);
      // (the kernels index regressor reg0 + blockIdx; the images of a chunk start at its first regressor)
      hipLaunchKernelGGL(marg_image_kernel<T>, dim3((unsigned)nb, 2), dim3(kThreads), TC::LDS_BYTES, h->stream, a.U, a.ldu, a.strideU, (int)D,
                         img - b0 * G::IMG_ELEMS, a.info, (int)b0);
      // two workgroups per CU, ONE round of them (tools/marg128_bench, 64 x 4096 inputs, stream kernel alone: 8 workgroups per
      // regressor 104.8 us, 16 -- two rounds -- 116.3, 32: 126.9; 256 regressors: 2 per regressor); every workgroup copies the
      // 74 KB image once: at least four tiles per wave
      const int64_t per_reg = std::max<int64_t>(1, std::min<int64_t>((ntiles + 15) / 16, (2 * (int64_t)h->cus + nb - 1) / nb));
      a.reg0 = (int)b0;
      hipLaunchKernelGGL(gemm_kern, dim3((unsigned)per_reg, (unsigned)nb), dim3(kThreads), G::LDS_BYTES, h->stream, a,
                         (const T*)(img - b0 * G::IMG_ELEMS));
    }
  } else {
    // inputs as rows of an LDS block; with a factor: Y = X'L^-T by the TRSM sweep (MFMA), fused mean / row sum of squares;
    // mean-only and diagonal-prior calls are pure streams through the same tile loop (smaller LDS image, 2 workgroups/CU)
    using TC = TrsmCfg<T>;
    auto kern = marginals_mfma_kernel<T>;
    const bool use_factor = var && kind == BLR_PRIOR_UPPER_FACTOR;
    const int xs_bytes = (TC::RB * TC::LDX * (int)sizeof(T) + 15) & ~15;
    const int lds = use_factor ? TC::LDS_BYTES + kPB * (int)sizeof(T) : xs_bytes + 2 * kPB * (int)sizeof(T) + 16;
    { const int rc_lds = set_lds_once(h, reinterpret_cast<const void*>(kern), (size_t)(TC::LDS_BYTES + kPB * (int)sizeof(T))); if (rc_lds) return rc_lds; }
    // every workgroup amortises its set-up over several tiles: aim at ~2 rounds of the chip
    const int64_t ntiles = (N + TC::RB - 1) / TC::RB;
    const int64_t slots = use_factor ? 512 : 1024;
    const int64_t per_reg = std::max<int64_t>(1, std::min<int64_t>(ntiles, (slots + B - 1) / B));
    for (int64_t b0 = 0; b0 < B; b0 += 65535) {  // grid.y <= 65535
      a.reg0 = (int)b0;
      dim3 grid((unsigned)per_reg, (unsigned)std::min<int64_t>(65535, B - b0));
      hipLaunchKernelGGL(kern, grid, dim3(kThreads), lds, h->stream, a);
    }
  }
  HIP_TRY(h, hipGetLastError());
  if (memspace == BLR_MEM_HOST) {
    if (mean) HIP_TRY(h, hipMemcpyAsync(mean, a.mean, extent(B, stridemean, (size_t)N) * sizeof(T), hipMemcpyDeviceToHost, h->stream));
    if (var) HIP_TRY(h, hipMemcpyAsync(var, a.var, extent(B, stridevar, (size_t)N) * sizeof(T), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(info, info_out_dev, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  } else if (!h->async) {
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  }
  return 0;
}

// ---- N-sharded single regressor: additive statistics of a column block, and the finish from their sum ----
template <typename T>
int gram_stats(blr_handle* h, int layout, int64_t D64, int64_t N64, const T* X, int64_t ldx, const T* y, int noise_kind,
               const T* s, const T* mw, T* stats, int64_t lds, double* scal) {
  if (!h) return -1;
  h->err.clear();
  if (layout != BLR_LAYOUT_COLVECS && layout != BLR_LAYOUT_ROWVECS) return bad_arg(h, 2, "unknown layout (reference :26-31)");
  if (D64 < 1 || D64 > kMaxLargeD) return bad_arg(h, 3, "D out of range for this build (1..8192)");
  if (N64 < 1 || N64 > (1 << 30)) return bad_arg(h, 4, "N out of range (>= 1)");
  const int D = (int)D64, N = (int)N64;
  const int DP = (D + kPB - 1) / kPB * kPB, NC = DP / kPB;
  if (!X) return bad_arg(h, 5, "X is NULL");
  if (layout == BLR_LAYOUT_COLVECS ? ldx < D : ldx < N) return bad_arg(h, 6, "ldx too small");
  if (!y) return bad_arg(h, 7, "y is NULL");
  if (noise_kind != BLR_NOISE_ISOTROPIC && noise_kind != BLR_NOISE_DIAGONAL) return bad_arg(h, 8, "noise_kind");
  if (!s) return bad_arg(h, 9, "s is NULL");
  if (!mw) return bad_arg(h, 10, "mw is NULL");
  if (!stats) return bad_arg(h, 11, "stats is NULL");
  if (lds < DP + kPB) return bad_arg(h, 12, "lds < 128 ceil(D/128) + 128");
  if (!scal) return bad_arg(h, 13, "scal is NULL");
  HIP_TRY(h, hipSetDevice(h->device));
  using LC = LargeCfg<T>;
  const int ntiles = NC * (NC + 1) / 2;
  const int max_split = std::max(1, std::min(64, (N + LC::NSC - 1) / LC::NSC));
  int nsplit = 1;
  double best = 0.0;
  for (int sp = 1; sp <= max_split; ++sp) {
    const int wgs = ntiles * sp;
    const int rounds = (wgs + 511) / 512;
    const double eff = (double)wgs / (rounds * 512.0) - 0.002 * sp;
    if (eff > best) { best = eff; nsplit = sp; }
  }
  const int gridc = 1024;
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
  const size_t o_gp = carve((size_t)nsplit * ntiles * kPB * kPB * sizeof(T));
  const size_t o_bp = carve((size_t)nsplit * NC * kPB * sizeof(double));
  const size_t o_r = carve((size_t)N * sizeof(T));
  const size_t o_wv = carve(noise_kind == BLR_NOISE_DIAGONAL ? (size_t)N * sizeof(T) : 0);  // 1 / s_n for the Gram launch
  const size_t o_q = carve((size_t)gridc * sizeof(double));
  const size_t o_l = carve((size_t)gridc * sizeof(double));
  int rc = ensure_ws(h, off);
  if (rc) return rc;
  char* ws = h->ws;
  T* Gpart = reinterpret_cast<T*>(ws + o_gp);
  double* bpart = reinterpret_cast<double*>(ws + o_bp);
  T* rvec = reinterpret_cast<T*>(ws + o_r);
  T* wvec = noise_kind == BLR_NOISE_DIAGONAL ? reinterpret_cast<T*>(ws + o_wv) : nullptr;
  double* qpart = reinterpret_cast<double*>(ws + o_q);
  double* lpart = reinterpret_cast<double*>(ws + o_l);
  HIP_TRY(h, hipMemsetAsync(bpart, 0, (size_t)nsplit * NC * kPB * sizeof(double), h->stream));
  {
    ColstatsArgs<T> c{};
    c.X = X; c.ldx = ldx; c.y = y; c.s = s; c.mw = mw; c.r = rvec; c.w = wvec; c.qpart = qpart; c.lpart = lpart;
    c.layout = layout; c.noise_kind = noise_kind; c.D = D; c.N = N;
    size_t ldsb = (((size_t)D * sizeof(T) + 15) & ~(size_t)15) + 64;
    hipLaunchKernelGGL(colstats_kernel<T>, dim3(gridc), dim3(kThreads), ldsb, h->stream, c);
    hipLaunchKernelGGL(stats_scalars_kernel<T>, dim3(1), dim3(kThreads), 0, h->stream, (const double*)qpart, (const double*)lpart, gridc,
                       noise_kind, s, N, scal);
  }
  if ((rc = set_lds<T>(h, reinterpret_cast<const void*>(gram_tile_kernel<T>), LC::LDS_BYTES))) return rc;
  {
    GramTileArgs<T> g{};
    g.X = X; g.ldx = ldx; g.layout = layout;
    g.use_dma = (layout == LAYOUT_COLVECS && ((uintptr_t)X % 16 == 0) && ((ldx * (int64_t)sizeof(T)) % 16 == 0)) ? 1 : 0;
    g.s = s; g.noise_kind = noise_kind; g.r = rvec; g.wpre = wvec;
    g.D = D; g.n_begin = 0; g.n_end = N; g.nsplit = nsplit;
    g.tile_i0 = 0; g.tile_j0 = 0; g.tri = 1; g.ntiles = ntiles; g.nblocks = NC;
    g.Gpart = Gpart; g.bpart = bpart; g.mode_out = 0; g.xcd_swizzle = nsplit > 1 ? 1 : 0;
    hipLaunchKernelGGL(gram_tile_kernel<T>, dim3(ntiles * nsplit), dim3(kThreads), LC::LDS_BYTES, h->stream, g);
    ReduceArgs<T> r{};
    r.Gpart = Gpart; r.bpart = bpart; r.nsplit_total = nsplit; r.ntiles = ntiles; r.nblocks = NC;
    r.Lw = nullptr; r.ldl = 0; r.prior_kind = 3 /* none: the prior is added after the cross-rank sum */;
    r.D = D; r.DP = DP; r.Abar = stats; r.lda = lds; r.Lw_post = nullptr; r.ldlp = 0;
    hipLaunchKernelGGL(gram_reduce_kernel<T>, dim3(ntiles + NC, 16), dim3(kThreads), 0, h->stream, r);
  }
  HIP_TRY(h, hipGetLastError());
  if (!h->async) HIP_TRY(h, hipStreamSynchronize(h->stream));
  return 0;
}

template <typename T>
int posterior_from_stats(blr_handle* h, int64_t D64, int64_t N_total, T* stats, int64_t lds, const double* scal, int prior_kind,
                         const T* mw, const T* Lw, int64_t ldl, T* mw_post, T* T_post, int64_t ldt, T* Lw_post, int64_t ldlp,
                         double* logpdf, int32_t* info) {
  if (!h) return -1;
  h->err.clear();
  if (D64 < 1 || D64 > kMaxLargeD) return bad_arg(h, 2, "D out of range for this build (1..8192)");
  if (N_total < 0 || N_total > ((int64_t)1 << 40)) return bad_arg(h, 3, "N_total out of range");
  const int D = (int)D64;
  const int DP = (D + kPB - 1) / kPB * kPB, NC = DP / kPB;
  if (!stats) return bad_arg(h, 4, "stats is NULL");
  if (lds < DP + kPB) return bad_arg(h, 5, "lds < 128 ceil(D/128) + 128");
  if (!scal) return bad_arg(h, 6, "scal is NULL");
  if (prior_kind != BLR_PRIOR_DENSE && prior_kind != BLR_PRIOR_DIAGONAL)
    return bad_arg(h, 7, "prior_kind (dense or diagonal precision; pass a carried-forward factor as U'U)");
  if (!mw) return bad_arg(h, 8, "mw is NULL");
  if (!Lw) return bad_arg(h, 9, "Lw is NULL");
  if (prior_kind == BLR_PRIOR_DENSE && ldl < D) return bad_arg(h, 10, "ldl < D");
  if (T_post && ldt < D) return bad_arg(h, 13, "ldt < D");
  if (Lw_post && ldlp < D) return bad_arg(h, 15, "ldlp < D");
  if (!info) return bad_arg(h, 17, "info is NULL");
  HIP_TRY(h, hipSetDevice(h->device));
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
  const size_t o_w = carve(prior_kind == BLR_PRIOR_DENSE ? (size_t)DP * DP * sizeof(T) : 0);
  const size_t o_m = carve((size_t)DP * DP * sizeof(T));
  const size_t o_sc = carve(64);
  int rc = ensure_ws(h, off);
  if (rc) return rc;
  char* ws = h->ws;
  T* W = reinterpret_cast<T*>(ws + o_w);
  T* Tfull = reinterpret_cast<T*>(ws + o_m);
  double* logdetLw = reinterpret_cast<double*>(ws + o_sc);
  int32_t* info_prior = reinterpret_cast<int32_t*>(ws + o_sc + 8);
  int32_t* info_chol = reinterpret_cast<int32_t*>(ws + o_sc + 12);
  HIP_TRY(h, hipMemsetAsync(ws + o_sc, 0, 64, h->stream));
  if (prior_kind == BLR_PRIOR_DENSE) {
    hipLaunchKernelGGL(prior_copy_kernel<T>, dim3(1024), dim3(kThreads), 0, h->stream, Lw, ldl, D, DP, W, (int64_t)DP);
    if ((rc = chol_large<T>(h, W, DP, DP, DP, info_prior))) return rc;
    hipLaunchKernelGGL(logdet_kernel<T>, dim3(1), dim3(kThreads), 0, h->stream, (const T*)W, (int64_t)DP, D, logdetLw);
  } else {
    hipLaunchKernelGGL(prior_diag_kernel<T>, dim3(1), dim3(kThreads), 0, h->stream, Lw, ldl, prior_kind, D, logdetLw, info_prior);
  }
  hipLaunchKernelGGL(stats_add_prior_kernel<T>, dim3(1024), dim3(kThreads), 0, h->stream, stats, lds, D, DP, Lw, ldl, prior_kind,
                     Lw_post, ldlp);
  HIP_TRY(h, hipMemcpyAsync(info_chol, info_prior, sizeof(int32_t), hipMemcpyDeviceToDevice, h->stream));
  if ((rc = chol_large<T>(h, stats, lds, DP, DP + TrailCfg<T>::SB, info_chol))) return rc;  // (see posterior_large_one)
  {
    dim3 grid((DP + 31) / 32, (DP + 31) / 32);
    hipLaunchKernelGGL(transpose_out_kernel<T>, grid, dim3(kThreads), 0, h->stream, (const T*)stats, lds, DP, Tfull, (int64_t)DP,
                       T_post, ldt, D, (int64_t)0, (int64_t)0, (const int32_t*)nullptr, (const unsigned*)nullptr, (const int32_t*)info_chol,
                       (int64_t)0);  // (info_chol carries the prior's status too)
  }
  {
    WaveSolveArgs<T> b{};
    b.Tf = Tfull; b.ldtf = DP; b.D = D; b.DP = DP;
    b.rhs = stats + DP; b.ldrhs = 0; b.rhs_inc = lds;
    b.add = mw; b.out = mw_post; b.ldout = 0;
    b.qpart = scal; b.lpart = scal + 1; b.nparts = 1; b.logdet_Lw_dev = logdetLw;
    b.noise_kind = BLR_NOISE_DIAGONAL; b.s = mw /* unused */; b.N = (int)std::min<int64_t>(N_total, 0x7fffffff);
    b.logpdf = logpdf; b.info = info; b.chol_info = info_chol;
    b.n_total = (double)N_total;
    if ((rc = launch_wave_solve<T>(h, b, NC, 1))) return rc;
  }
  HIP_TRY(h, hipGetLastError());
  if (!h->async) HIP_TRY(h, hipStreamSynchronize(h->stream));
  return 0;
}

// ---- gradient for D > 128: forward + backward panels over the tall matrix [F; X'; I], G regressors per launch ------------
// Regressor g of the group: caller's arrays at their element strides, the tall matrix and the per-observation vectors in ITS slice
// of the workspace (blockIdx.y of every launch; posterior_large_group's scheme).
struct GradGroupStrides {
  int64_t X, y, s, mwp, Tf, dX, dy, ds, dmw, Ai;
};
template <typename T>
size_t grad_large_ws_bytes(int64_t D, int64_t N, bool with_ainv, bool with_dmw) {
  const size_t DP = (size_t)((D + kPB - 1) / kPB * kPB), NP = (size_t)((N + kPB - 1) / kPB * kPB), DI = with_ainv ? DP : 0;
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  return al((DP + NP + DI) * DP * sizeof(T)) + al((NP + DI) * sizeof(double)) + 3 * al(NP * sizeof(T)) + al((NP + DI) * sizeof(T)) +
         al(with_dmw ? (NP / 64) * DP * sizeof(double) : 0);
}
template <typename T>
int logpdf_grad_large_group(blr_handle* h, int G, int layout, int64_t D, int64_t N, const T* X, int64_t ldx, const T* y, int noise_kind,
                            const T* s, const T* mwp, const T* Tfac, int64_t ldt, T* dX, int64_t lddx, T* dy, T* ds, T* dmw,
                            T* Ainv, int64_t ldai, int32_t* info_dev, const GradGroupStrides& gs) {
  using TC = TrsmCfg<T>;
  using LC = LargeCfg<T>;
  const int DP = (int)((D + kPB - 1) / kPB * kPB), NC = DP / kPB;
  const int NP = (int)((N + kPB - 1) / kPB * kPB);
  const int DI = Ainv ? DP : 0;
  const int R = DP + NP + DI;
  const int64_t ldy = R;
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
  const size_t o_y = carve((size_t)ldy * DP * sizeof(T));
  const size_t o_sq = carve((size_t)(NP + DI) * sizeof(double));
  const size_t o_mu = carve((size_t)NP * sizeof(T));
  const size_t o_var = carve((size_t)(NP + DI) * sizeof(T));
  const size_t o_r = carve((size_t)NP * sizeof(T));
  const size_t o_w = carve((size_t)NP * sizeof(T));
  const size_t o_part = carve(dmw ? (size_t)(NP / 64) * DP * sizeof(double) : 0);
  const int64_t wsb = (int64_t)off;  // one regressor's slice
  int rc = ensure_ws(h, off * (size_t)G);
  if (rc) return rc;
  T* Ybar = reinterpret_cast<T*>(h->ws + o_y);
  double* rowsq = reinterpret_cast<double*>(h->ws + o_sq);
  T* mu = reinterpret_cast<T*>(h->ws + o_mu);
  T* var = reinterpret_cast<T*>(h->ws + o_var);
  T* rvec = reinterpret_cast<T*>(h->ws + o_r);
  T* wvec = reinterpret_cast<T*>(h->ws + o_w);
  double* part = dmw ? reinterpret_cast<double*>(h->ws + o_part) : nullptr;
  const unsigned ug = (unsigned)G;

  {
    dim3 grid((DP + 31) / 32, (DP + 31) / 32, ug);
    hipLaunchKernelGGL(factor_sym_fill_kernel<T>, grid, dim3(kThreads), 0, h->stream, Tfac, ldt, (int)D, DP, Ybar, ldy, gs.Tf, wsb);
    MeanFillArgs<T> m{};
    m.X = X; m.ldx = ldx; m.layout = layout; m.mw = mwp; m.mean = mu; m.Ybar = Ybar; m.ldy = ldy; m.row0 = DP;
    m.D = (int)D; m.DP = DP; m.N = (int)N;
    m.grp_X = gs.X; m.grp_mw = gs.mwp; m.grp_ws = wsb;
    hipLaunchKernelGGL(mean_fill_kernel<T>, dim3((unsigned)(NP / 64), ug), dim3(kThreads), 0, h->stream, m);
    if (DI) hipLaunchKernelGGL(identity_rows_kernel<T>, dim3(G > 8 ? 128 : 1024, ug), dim3(kThreads), 0, h->stream, Ybar, ldy, DP + NP, DP, wsb);
  }
  if ((rc = set_lds<T>(h, reinterpret_cast<const void*>(trsm_block_kernel<T>), TC::LDS_BYTES))) return rc;
  if ((rc = set_lds<T>(h, reinterpret_cast<const void*>(trsm_back_block_kernel<T>), TC::LDS_BYTES))) return rc;
  if ((rc = set_lds<T>(h, reinterpret_cast<const void*>(gram_tile_kernel<T>), LC::LDS_BYTES))) return rc;
  const int nyb = (NP + DI) / kPB;                       // row blocks below the factor block
  const int nblk = (NP + DI + TC::RB - 1) / TC::RB;
  auto trailing = [&](int p, int j0, int ncolblocks) {   // C(I, J) -= Y(I, p) F(J, p)' for J = j0 .. j0 + ncolblocks - 1
    GramTileArgs<T> g{};
    g.X = Ybar + (int64_t)p * kPB * ldy; g.ldx = ldy; g.layout = LAYOUT_COLVECS; g.use_dma = 1;
    g.s = nullptr; g.noise_kind = NOISE_ISOTROPIC; g.r = nullptr;
    g.D = R; g.n_begin = 0; g.n_end = kPB; g.nsplit = 1;
    g.tile_i0 = NC; g.tile_j0 = j0; g.tri = 3; g.ntile_rows = nyb; g.ntiles = nyb * ncolblocks; g.nblocks = NC + nyb;
    g.C = Ybar; g.ldc = ldy; g.mode_out = 1;
    g.grp_X = wsb / (int64_t)sizeof(T); g.grp_s = 0; g.grp_ws = wsb;
    hipLaunchKernelGGL(gram_tile_kernel<T>, dim3(g.ntiles, ug), dim3(kThreads), LC::LDS_BYTES, h->stream, g);
  };
  for (int p = 0; p < NC; ++p) {  // forward: rows x' -> x'L^-T, with the row sums of squares riding along
    RowSqArgs<T> rs{};
    rs.acc = rowsq; rs.var = var; rs.s = s; rs.noise_kind = noise_kind; rs.N = (int)N; rs.first = p == 0; rs.last = p == NC - 1;
    rs.grp_ws = wsb; rs.grp_s = gs.s;
    hipLaunchKernelGGL(trsm_block_kernel<T>, dim3(nblk, ug), dim3(kThreads), TC::LDS_BYTES, h->stream, Ybar, ldy, p, DP, R,
                       (const int32_t*)info_dev, rs, wsb);
    if (p + 1 < NC) trailing(p, p + 1, NC - 1 - p);
  }
  {
    GradObsGroup og{gs.y, gs.s, gs.dy, gs.ds, wsb};
    hipLaunchKernelGGL(grad_obs_kernel<T>, dim3((unsigned)((N + kThreads - 1) / kThreads), ug), dim3(kThreads), 0, h->stream, y,
                       (const T*)mu, (const T*)var, s, noise_kind, (int)N, rvec, wvec, dy, ds, og);
  }
  for (int p = NC - 1; p >= 0; --p) {  // backward: x'L^-T -> x'L^-T L^-1 = x'A^-1
    hipLaunchKernelGGL(trsm_back_block_kernel<T>, dim3(nblk, ug), dim3(kThreads), TC::LDS_BYTES, h->stream, Ybar, ldy, p, DP, R,
                       (const int32_t*)info_dev, wsb);
    if (p > 0) trailing(p, 0, p);  // second operand: the UPPER triangle of the factor block (T = L')
  }
  if (dX || dmw) {
    GradOutArgs<T> o{};
    o.Ybar = Ybar; o.ldy = ldy; o.row0 = DP; o.X = X; o.ldx = ldx; o.layout = layout;
    o.rvec = rvec; o.wvec = wvec; o.mwp = mwp; o.dX = dX; o.lddx = lddx; o.dmw_part = part;
    o.D = (int)D; o.DP = DP; o.N = (int)N;
    o.grp_X = gs.X; o.grp_mwp = gs.mwp; o.grp_dX = gs.dX; o.grp_ws = wsb;
    hipLaunchKernelGGL(grad_out_large_kernel<T>, dim3((unsigned)(NP / 64), ug), dim3(kThreads), 0, h->stream, o);
    if (dmw)
      hipLaunchKernelGGL(grad_reduce_large_kernel<T>, dim3((unsigned)((D + kThreads - 1) / kThreads), ug), dim3(kThreads), 0, h->stream,
                         (const double*)part, NP / 64, DP, (int)D, dmw, wsb, gs.dmw);
  }
  if (Ainv)
    hipLaunchKernelGGL(ainv_copy_kernel<T>, dim3(G > 8 ? 128 : 1024, ug), dim3(kThreads), 0, h->stream, (const T*)Ybar, ldy, DP + NP, (int)D, Ainv, ldai,
                       wsb, gs.Ai);
  HIP_TRY(h, hipGetLastError());
  return 0;
}

// ---- gradient of the log marginal likelihood (D <= 128): fused posterior, then the two-sweep gradient kernel -----------
template <typename T>
int logpdf_grad_batched(blr_handle* h, int memspace, int layout, int64_t B, int64_t D, int64_t N, const T* X, int64_t ldx,
                        int64_t strideX, const T* y, int64_t stridey, int noise_kind, const T* s, int64_t strides,
                        int prior_kind, const T* mw, int64_t stridemw, const T* Lw, int64_t ldl, int64_t strideLw,
                        double* logpdf, T* dX, int64_t lddx, int64_t stridedX, T* dy, int64_t stridedy, T* ds,
                        int64_t strideds, T* dmw, int64_t stridedmw, T* mw_post, int64_t stride_mwpost, T* Ainv,
                        int64_t ldai, int64_t strideAi, int32_t* info) {
  if (!h) return -1;
  h->err.clear();
  if (memspace != BLR_MEM_HOST && memspace != BLR_MEM_DEVICE) return bad_arg(h, 2, "memspace");
  if (layout != BLR_LAYOUT_COLVECS && layout != BLR_LAYOUT_ROWVECS) return bad_arg(h, 3, "unknown layout (reference :26-31)");
  if (B < 0 || B > (1 << 30)) return bad_arg(h, 4, "B out of range (0..2^30)");
  if (D < 1 || D > kMaxLargeD) return bad_arg(h, 5, "D out of range for this build (1..8192)");
  if (N < 1 || N > (1 << 30)) return bad_arg(h, 6, "N out of range (>= 1)");
  if (B == 0) return 0;
  if (!X) return bad_arg(h, 7, "X is NULL");
  if (layout == BLR_LAYOUT_COLVECS ? ldx < D : ldx < N) return bad_arg(h, 8, "ldx too small");
  if (strideX < 0) return bad_arg(h, 9, "strideX < 0");
  if (!y) return bad_arg(h, 10, "y is NULL");
  if (noise_kind != BLR_NOISE_ISOTROPIC && noise_kind != BLR_NOISE_DIAGONAL) return bad_arg(h, 12, "noise_kind");
  if (!s) return bad_arg(h, 13, "s is NULL");
  if (prior_kind != BLR_PRIOR_DENSE && prior_kind != BLR_PRIOR_UPPER_FACTOR && prior_kind != BLR_PRIOR_DIAGONAL)
    return bad_arg(h, 15, "prior_kind");
  if (!mw) return bad_arg(h, 16, "mw is NULL");
  if (!Lw) return bad_arg(h, 18, "Lw is NULL");
  if (prior_kind != BLR_PRIOR_DIAGONAL && ldl < D) return bad_arg(h, 19, "ldl < D");
  if (dX && (layout == BLR_LAYOUT_COLVECS ? lddx < D : lddx < N)) return bad_arg(h, 23, "lddx too small");
  if (dy && B > 1 && stridedy < N) return bad_arg(h, 26, "stridedy < N");
  if (ds && B > 1 && strideds < N) return bad_arg(h, 28, "strideds < N");
  if (dmw && B > 1 && stridedmw < D) return bad_arg(h, 30, "stridedmw < D");
  if (mw_post && B > 1 && stride_mwpost < D) return bad_arg(h, 32, "stride_mwpost < D");
  if (Ainv && ldai < D) return bad_arg(h, 34, "ldai < D");
  if (!info) return bad_arg(h, 36, "info is NULL");
  HIP_TRY(h, hipSetDevice(h->device));

  Staging guard(h);
  PosteriorArgs<T> a{};
  a.ldx = ldx; a.strideX = strideX; a.stridey = stridey; a.strides = strides; a.stridemw = stridemw;
  a.ldl = ldl; a.strideLw = strideLw;
  a.layout = layout; a.noise_kind = noise_kind; a.prior_kind = prior_kind;
  a.D = (int)D; a.N = (int)N; a.B = (int)B;
  GradArgs<T> g{};
  const size_t x_one = layout == BLR_LAYOUT_COLVECS ? mat_extent(D, N, ldx) : mat_extent(N, D, ldx);
  const size_t dx_one = layout == BLR_LAYOUT_COLVECS ? mat_extent(D, N, lddx) : mat_extent(N, D, lddx);
  const size_t lw_one = prior_kind == BLR_PRIOR_DIAGONAL ? (size_t)D : mat_extent(D, D, ldl);
  const size_t s_one = noise_kind == BLR_NOISE_DIAGONAL ? (size_t)N : 1;
  int rc;
  T *dX_d = dX, *dy_d = dy, *ds_d = ds, *dmw_d = dmw, *mwp_d = mw_post, *Ai_d = Ainv;
  double* lp_d = logpdf;
  int32_t* info_d = info;
  if (memspace == BLR_MEM_HOST) {
    if ((rc = stage_in(h, X, extent(B, strideX, x_one), &a.X))) return rc;
    if ((rc = stage_in(h, y, extent(B, stridey, (size_t)N), &a.y))) return rc;
    if ((rc = stage_in(h, s, extent(B, strides, s_one), &a.s))) return rc;
    if ((rc = stage_in(h, mw, extent(B, stridemw, (size_t)D), &a.mw))) return rc;
    if ((rc = stage_in(h, Lw, extent(B, strideLw, lw_one), &a.Lw))) return rc;
    if ((rc = stage_out_alloc(h, dX, extent(B, stridedX, dx_one), &dX_d))) return rc;
    if ((rc = stage_out_alloc(h, dy, extent(B, stridedy, (size_t)N), &dy_d))) return rc;
    if ((rc = stage_out_alloc(h, ds, extent(B, strideds, (size_t)N), &ds_d))) return rc;
    if ((rc = stage_out_alloc(h, dmw, extent(B, stridedmw, (size_t)D), &dmw_d))) return rc;
    if ((rc = stage_out_alloc(h, mw_post, extent(B, stride_mwpost, (size_t)D), &mwp_d))) return rc;
    if ((rc = stage_out_alloc(h, Ainv, extent(B, strideAi, mat_extent(D, D, ldai)), &Ai_d))) return rc;
    if ((rc = stage_out_alloc(h, logpdf, (size_t)B, &lp_d))) return rc;
    if ((rc = stage_out_alloc(h, info, (size_t)B, &info_d))) return rc;
  } else {
    a.X = X; a.y = y; a.s = s; a.mw = mw; a.Lw = Lw;
  }
  if (D > kMaxSmallD) {
    // large-D path: the regressors go through the update (posterior_large_group) and through the sweeps over the tall matrix in
    // groups that share every launch; factors and posterior means of a group in buffers of their own
    int gmax = kChainBatchMax;
    if (h->opt.chain_batch > 0) gmax = std::min(kChainBatchMax, h->opt.chain_batch);  // measurements only
    if ((strideX * (int64_t)sizeof(T)) % 16 != 0) gmax = 1;
    {
      size_t ws_cap = kChainWorkspace;
      if (h->opt.chain_ws_mb > 0) ws_cap = (size_t)h->opt.chain_ws_mb << 20;
      const size_t one_ws = grad_large_ws_bytes<T>(D, N, Ai_d != nullptr, dmw_d != nullptr) + (size_t)D * D * sizeof(T);
      gmax = (int)std::max<size_t>(1, std::min<size_t>((size_t)gmax, ws_cap / one_ws));
      const int64_t ngroups = std::max<int64_t>(1, (B + gmax - 1) / gmax);
      gmax = (int)((B + ngroups - 1) / ngroups);  // even groups
    }
    T* Tf = nullptr;
    T* mp = nullptr;
    {
      void* p0 = nullptr;
      HIP_TRY(h, hipMalloc(&p0, (size_t)gmax * D * D * sizeof(T)));
      h->staged.push_back(p0);
      Tf = static_cast<T*>(p0);
      void* p1 = nullptr;
      HIP_TRY(h, hipMalloc(&p1, (size_t)gmax * D * sizeof(T)));
      h->staged.push_back(p1);
      mp = static_cast<T*>(p1);
    }
    a.vec_ok = 0;
    for (int64_t reg = 0; reg < B;) {
      PosteriorArgs<T> one = a;
      // posterior_large_group indexes the batched arrays by the regressor's number: shift the outputs that are NOT batched here
      one.stride_mwpost = mwp_d ? stride_mwpost : D;
      one.mw_post = mwp_d ? mwp_d : mp - reg * one.stride_mwpost;
      one.ldt = D; one.strideT = D * D; one.T_post = Tf - reg * one.strideT;
      one.Lw_post = nullptr; one.ldlp = D; one.strideLp = 0;
      one.logpdf = lp_d; one.info = info_d;
      int done = 1;
      if ((rc = posterior_large_group<T>(h, one, reg, (int)std::min<int64_t>(gmax, B - reg), &done))) return rc;
      GradGroupStrides gs{strideX, stridey, strides, one.stride_mwpost, one.strideT, stridedX, stridedy, strideds, stridedmw, strideAi};
      if ((rc = logpdf_grad_large_group<T>(h, done, layout, D, N, a.X + reg * strideX, ldx, a.y + reg * stridey, noise_kind,
                                           a.s + reg * strides, one.mw_post + reg * one.stride_mwpost, Tf, D,
                                           dX_d ? dX_d + reg * stridedX : nullptr, lddx, dy_d ? dy_d + reg * stridedy : nullptr,
                                           ds_d ? ds_d + reg * strideds : nullptr, dmw_d ? dmw_d + reg * stridedmw : nullptr,
                                           Ai_d ? Ai_d + reg * strideAi : nullptr, ldai, info_d + reg, gs)))
        return rc;
      reg += done;
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));  // Tf / mp are temporaries of this call
    if (memspace == BLR_MEM_HOST) {
      auto back = [&](void* dst, const void* src, size_t bytes) -> int {
        if (dst && bytes) HIP_TRY(h, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
        return 0;
      };
      if ((rc = back(dX, dX_d, extent(B, stridedX, dx_one) * sizeof(T)))) return rc;
      if ((rc = back(dy, dy_d, extent(B, stridedy, (size_t)N) * sizeof(T)))) return rc;
      if ((rc = back(ds, ds_d, extent(B, strideds, (size_t)N) * sizeof(T)))) return rc;
      if ((rc = back(dmw, dmw_d, extent(B, stridedmw, (size_t)D) * sizeof(T)))) return rc;
      if ((rc = back(mw_post, mwp_d, extent(B, stride_mwpost, (size_t)D) * sizeof(T)))) return rc;
      if ((rc = back(Ainv, Ai_d, extent(B, strideAi, mat_extent(D, D, ldai)) * sizeof(T)))) return rc;
      if ((rc = back(logpdf, lp_d, (size_t)B * sizeof(double)))) return rc;
      if ((rc = back(info, info_d, (size_t)B * sizeof(int32_t)))) return rc;
    }
    return 0;
  }
  // workspace: factor T [B][D x D], posterior mean (if the caller does not want it), dmw partials
  using TC = TrsmCfg<T>;
  const int64_t ntiles = (N + TC::RB - 1) / TC::RB + (Ainv ? (D + TC::RB - 1) / TC::RB : 0);
  int64_t per_reg = std::max<int64_t>(1, std::min<int64_t>(ntiles, (512 + B - 1) / B));
  // D = 128, aligned ColVecs: both sweeps as products with the explicit triangular inverse (grad_gemm_kernel, blr_marginals.hpp):
  // one workgroup per CU, a wave per 16 inputs; A^-1 itself (if wanted) = M M' from the same images
  using GG = GradGemmCfg<T>;
  using MG = MargGemmCfg<T>;
  const bool rowv = layout == BLR_LAYOUT_ROWVECS;  // (RowVecs: scalar loads, no alignment to ask for)
  const bool gemm = !h->opt.no_grad_gemm && D == kPB && N >= 64 && B <= 65535 &&
                    (rowv || ((ldx % Mfma<T>::VEC) == 0 && ((uintptr_t)a.X % 16) == 0 && ((strideX * (int64_t)sizeof(T)) % 16) == 0)) &&
                    (size_t)B * 2 * MG::IMG_ELEMS * sizeof(T) <= ((size_t)1 << 30);
  if (gemm) per_reg = std::max<int64_t>(1, std::min<int64_t>(((N + 15) / 16 + 4 * GG::WAVES - 1) / (4 * GG::WAVES), ((int64_t)h->cus + B - 1) / B));
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
  const size_t o_T = carve((size_t)B * D * D * sizeof(T));
  const size_t o_mp = carve(mwp_d ? 0 : (size_t)B * D * sizeof(T));
  const size_t o_part = carve(dmw_d ? (size_t)B * per_reg * kPB * sizeof(double) : 0);
  const size_t o_img = carve(gemm ? (size_t)B * 2 * MG::IMG_ELEMS * sizeof(T) : 0);
  if ((rc = ensure_ws(h, off))) return rc;
  T* Tf = reinterpret_cast<T*>(h->ws + o_T);
  int64_t smp = stride_mwpost;
  if (!mwp_d) { mwp_d = reinterpret_cast<T*>(h->ws + o_mp); smp = D; }
  double* part = dmw_d ? reinterpret_cast<double*>(h->ws + o_part) : nullptr;

  a.mw_post = mwp_d; a.stride_mwpost = smp;
  a.T_post = Tf; a.ldt = D; a.strideT = D * D;
  a.Lw_post = nullptr; a.ldlp = D; a.strideLp = 0;
  a.logpdf = lp_d; a.info = info_d;
  a.vec_ok = (layout == BLR_LAYOUT_COLVECS && D % Mfma<T>::VEC == 0 && aligned16(a.X, ldx, strideX)) ? 1 : 0;
  if ((rc = dispatch_posterior<T>(h, a))) return rc;

  g.X = a.X; g.ldx = ldx; g.strideX = strideX; g.y = a.y; g.stridey = stridey; g.s = a.s; g.strides = strides;
  g.mwp = mwp_d; g.stridemwp = smp; g.U = Tf; g.ldu = D; g.strideU = D * D;
  g.dX = dX_d; g.lddx = lddx; g.stridedX = stridedX; g.dy = dy_d; g.stridedy = stridedy; g.ds = ds_d; g.strideds = strideds;
  g.dmw_part = part; g.Ainv = Ai_d; g.ldai = ldai; g.strideAi = strideAi; g.info = info_d;
  g.layout = layout; g.noise_kind = noise_kind; g.D = (int)D; g.N = (int)N; g.B = (int)B;
  if (gemm) {
    if ((rc = set_lds_once(h, reinterpret_cast<const void*>(marg_image_kernel<T>), (size_t)TC::LDS_BYTES))) return rc;
    void (*const gg_kern)(GradArgs<T>, const T*, const T*) = rowv ? grad_gemm_kernel<T, true> : grad_gemm_kernel<T, false>;
    if ((rc = set_lds_once(h, reinterpret_cast<const void*>(gg_kern), (size_t)GG::LDS_BYTES))) return rc;
    T* const img = reinterpret_cast<T*>(h->ws + o_img);
    T* const img2 = img + B * MG::IMG_ELEMS;
    hipLaunchKernelGGL(marg_image_kernel<T>, dim3((unsigned)B, 2), dim3(kThreads), TC::LDS_BYTES, h->stream, (const T*)Tf, D, D * D, (int)D, img,
                       (const int32_t*)info_d, 0, 0, 1, (int64_t)0, img2);
    g.reg0 = 0;
    hipLaunchKernelGGL(gg_kern, dim3((unsigned)per_reg, (unsigned)B), dim3(GG::THREADS), GG::LDS_BYTES, h->stream, g, (const T*)img,
                       (const T*)img2);
    if (dmw_d)
      hipLaunchKernelGGL(grad_reduce_kernel<T>, dim3((unsigned)B), dim3(kPB), 0, h->stream, (const double*)part, (int)per_reg,
                         dmw_d, stridedmw, (int)D);
  } else {
    auto kern = logpdf_grad_kernel<T>;
    const int lds = TC::LDS_BYTES + (kPB + 3 * TC::RB) * (int)sizeof(T);
    { const int rc_lds = set_lds_once(h, reinterpret_cast<const void*>(kern), (size_t)(lds)); if (rc_lds) return rc_lds; }
    for (int64_t b0 = 0; b0 < B; b0 += 65535) {  // grid.y <= 65535
      g.reg0 = (int)b0;
      hipLaunchKernelGGL(kern, dim3((unsigned)per_reg, (unsigned)std::min<int64_t>(65535, B - b0)), dim3(kThreads), lds, h->stream, g);
    }
    if (dmw_d)
      hipLaunchKernelGGL(grad_reduce_kernel<T>, dim3((unsigned)B), dim3(kPB), 0, h->stream, (const double*)part, (int)per_reg,
                         dmw_d, stridedmw, (int)D);
  }
  HIP_TRY(h, hipGetLastError());
  if (memspace == BLR_MEM_HOST) {
    auto back = [&](void* dst, const void* src, size_t bytes) -> int {
      if (dst && bytes) HIP_TRY(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
      return 0;
    };
    if ((rc = back(dX, dX_d, extent(B, stridedX, dx_one) * sizeof(T)))) return rc;
    if ((rc = back(dy, dy_d, extent(B, stridedy, (size_t)N) * sizeof(T)))) return rc;
    if ((rc = back(ds, ds_d, extent(B, strideds, (size_t)N) * sizeof(T)))) return rc;
    if ((rc = back(dmw, dmw_d, extent(B, stridedmw, (size_t)D) * sizeof(T)))) return rc;
    if ((rc = back(mw_post, mwp_d, extent(B, stride_mwpost, (size_t)D) * sizeof(T)))) return rc;
    if ((rc = back(Ainv, Ai_d, extent(B, strideAi, mat_extent(D, D, ldai)) * sizeof(T)))) return rc;
    if ((rc = back(logpdf, lp_d, (size_t)B * sizeof(double)))) return rc;
    if ((rc = back(info, info_d, (size_t)B * sizeof(int32_t)))) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  } else if (!h->async) {
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  }
  return 0;
}

// ---- shared-X multi-output evidence: logpdf(fx, Y::Matrix), optional per-column posterior means ----
template <typename T>
int logpdf_multi(blr_handle* h, int memspace, int layout, int64_t D, int64_t N, int64_t S, const T* X, int64_t ldx, const T* Y,
                 int64_t ldY, int noise_kind, const T* s, int prior_kind, const T* mw, const T* Lw, int64_t ldl, double* logpdf,
                 T* mw_post, int64_t ldmp, int32_t* info) {
  if (!h) return -1;
  h->err.clear();
  if (memspace != BLR_MEM_HOST && memspace != BLR_MEM_DEVICE) return bad_arg(h, 2, "memspace");
  if (layout != BLR_LAYOUT_COLVECS && layout != BLR_LAYOUT_ROWVECS) return bad_arg(h, 3, "unknown layout (reference :26-31)");
  if (D < 1 || D > kMaxLargeD) return bad_arg(h, 4, "D out of range for this build (1..8192)");
  if (N < 1 || N > (1 << 30)) return bad_arg(h, 5, "N out of range (>= 1)");
  if (S < 0 || S > 65536) return bad_arg(h, 6, "S out of range (0..65536)");
  if (S == 0) return 0;
  if (!X) return bad_arg(h, 7, "X is NULL");
  if (layout == BLR_LAYOUT_COLVECS ? ldx < D : ldx < N) return bad_arg(h, 8, "ldx too small");
  if (!Y) return bad_arg(h, 9, "Y is NULL");
  if (ldY < N) return bad_arg(h, 10, "ldY < N (reference :74 length check)");
  if (noise_kind != BLR_NOISE_ISOTROPIC && noise_kind != BLR_NOISE_DIAGONAL) return bad_arg(h, 11, "noise_kind");
  if (!s) return bad_arg(h, 12, "s is NULL");
  if (prior_kind != BLR_PRIOR_DENSE && prior_kind != BLR_PRIOR_UPPER_FACTOR && prior_kind != BLR_PRIOR_DIAGONAL)
    return bad_arg(h, 13, "prior_kind");
  if (!mw) return bad_arg(h, 14, "mw is NULL");
  if (!Lw) return bad_arg(h, 15, "Lw is NULL");
  if (prior_kind != BLR_PRIOR_DIAGONAL && ldl < D) return bad_arg(h, 16, "ldl < D");
  if (!logpdf) return bad_arg(h, 17, "logpdf is NULL");
  if (mw_post && ldmp < D) return bad_arg(h, 19, "ldmp < D");
  if (!info) return bad_arg(h, 20, "info is NULL");
  HIP_TRY(h, hipSetDevice(h->device));

  using TC = TrsmCfg<T>;
  using LC = LargeCfg<T>;
  Staging guard(h);
  int rc;
  const T *X_d = X, *Y_d = Y, *s_d = s, *mw_d = mw, *Lw_d = Lw;
  double* lp_d = logpdf;
  T* mp_d = mw_post;
  int32_t* info_d = info;
  if (memspace == BLR_MEM_HOST) {
    const size_t x_one = layout == BLR_LAYOUT_COLVECS ? mat_extent(D, N, ldx) : mat_extent(N, D, ldx);
    const size_t lw_one = prior_kind == BLR_PRIOR_DIAGONAL ? (size_t)D : mat_extent(D, D, ldl);
    if ((rc = stage_in(h, X, x_one, &X_d))) return rc;
    if ((rc = stage_in(h, Y, mat_extent(N, S, ldY), &Y_d))) return rc;
    if ((rc = stage_in(h, s, noise_kind == BLR_NOISE_DIAGONAL ? (size_t)N : 1, &s_d))) return rc;
    if ((rc = stage_in(h, mw, (size_t)D, &mw_d))) return rc;
    if ((rc = stage_in(h, Lw, lw_one, &Lw_d))) return rc;
    if ((rc = stage_out_alloc(h, logpdf, (size_t)S, &lp_d))) return rc;
    if ((rc = stage_out_alloc(h, mw_post, mw_post ? mat_extent(D, S, ldmp) : 0, &mp_d))) return rc;
    if ((rc = stage_out_alloc(h, info, (size_t)1, &info_d))) return rc;
  }
  const int DP = (int)((D + kPB - 1) / kPB * kPB), NC = DP / kPB;
  // fp32, ColVecs, D > 128, up to 128 columns: the residuals of ALL columns ride through the update of column 0 as one more row block of
  // operand planes -- their b_s = X Sigma^-1 (y_s - mu) come out of the same Gram launch (NC + 1 more macro tiles), their u_s = L^-1 b_s
  // out of the same blocked factorisation (rows it carries along anyway), their q_s out of the planes pass.  No residual matrix, no
  // second product over X, no panel sweep of its own (round 6; until then: steps (2) - (5) below, 0.5 ms of a 1.09 ms call at config 3's
  // shape with 64 columns, 9.7 x the algorithmic bytes).
  if constexpr (sizeof(T) == 4) {
    if (D > kMaxSmallD && layout == BLR_LAYOUT_COLVECS && S <= kPB && prior_kind != BLR_PRIOR_UPPER_FACTOR && !h->opt.no_planes &&
        !h->opt.no_bf16x3 && !h->opt.no_fp16_planes && !h->opt.no_multi_planes) {
      void *vTf, *vlp0;
      {
        const size_t sz_tf = mp_d ? (((size_t)D * D * sizeof(T) + 255) & ~(size_t)255) : 0;
        if ((rc = ensure_aux(h, sz_tf + 256))) return rc;
        vTf = h->aux; vlp0 = h->aux + sz_tf;
      }
      T* Tf = static_cast<T*>(vTf);
      double* lp0 = static_cast<double*>(vlp0);
      blr_handle::MultiSrc ms{};
      ms.Y = Y_d; ms.ldY = ldY; ms.S = (int)S;
      PosteriorArgs<T> a{};
      a.X = X_d; a.ldx = ldx; a.strideX = 0; a.y = Y_d; a.stridey = 0; a.s = s_d; a.strides = 0; a.mw = mw_d; a.stridemw = 0;
      a.Lw = Lw_d; a.ldl = ldl; a.strideLw = 0;
      a.mw_post = nullptr; a.stride_mwpost = D; a.T_post = mp_d ? Tf : nullptr; a.ldt = D; a.strideT = D * D;  // (the means need T = L')
      a.Lw_post = nullptr; a.ldlp = D; a.strideLp = 0; a.logpdf = lp0; a.info = info_d;
      a.layout = layout; a.noise_kind = noise_kind; a.prior_kind = prior_kind; a.D = (int)D; a.N = (int)N; a.B = 1;
      a.vec_ok = (D % Mfma<T>::VEC == 0 && aligned16(X_d, ldx, 0)) ? 1 : 0;
      h->multi_src = &ms;
      rc = dispatch_posterior<T>(h, a);
      h->multi_src = nullptr;
      if (rc) return rc;
      if (!ms.done) return hip_fail(h, hipErrorInvalidValue, "multi-output rows were not taken along (internal)");
      hipLaunchKernelGGL(multi_rows_finish_kernel<T>, dim3((unsigned)S), dim3(kThreads), 0, h->stream, (const double*)lp0, (const double*)ms.qsp,
                         ms.nq, noise_kind == BLR_NOISE_ISOTROPIC ? s_d : (const T*)nullptr, (const T*)ms.Abar, ms.lda, ms.DP, (int)D, (int)S, lp_d);
      if (mp_d) {  // m_s = L^-T u_s, one wavefront of workgroups per column (as the weight draws of blr_sample_weights_*)
        WaveSolveArgs<T> b{};
        b.Tf = static_cast<const T*>(ms.Tfull); b.ldtf = ms.DP; b.D = (int)D; b.DP = ms.DP;
        b.rhs = static_cast<const T*>(ms.Abar) + ms.DP; b.ldrhs = 1; b.rhs_inc = ms.lda;
        b.add = mw_d; b.out = mp_d; b.ldout = ldmp;
        if ((rc = launch_wave_solve<T>(h, b, NC, S))) return rc;
      }
      HIP_TRY(h, hipGetLastError());
      if (memspace == BLR_MEM_HOST) {
        HIP_TRY(h, hipMemcpyAsync(logpdf, lp_d, (size_t)S * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        if (mw_post) HIP_TRY(h, hipMemcpyAsync(mw_post, mp_d, mat_extent(D, S, ldmp) * sizeof(T), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipMemcpyAsync(info, info_d, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
      } else if (!h->async) {
        HIP_TRY(h, hipStreamSynchronize(h->stream));
      }
      return 0;
    }
  }
  const int SP = (int)((S + kPB - 1) / kPB * kPB);
  const int NP64 = (int)((N + 63) / 64);
  const int64_t ldy = (int64_t)DP + SP;
  const int ntile_rows = SP / kPB, ntiles = ntile_rows * NC;
  // split-K over N: fill the 512 workgroup slots
  const int max_split = std::max(1, std::min<int>(64, (int)((N + LC::NSC - 1) / LC::NSC)));
  const int nsplit = std::max(1, std::min(max_split, 512 / std::max(1, ntiles)));
  // eleven call-scoped temporaries, carved out of the handle's side buffer (it only grows: no hipMalloc / hipFree -- each of which
  // drains the device -- on a steady-state call; neither the update of step (1) nor the mean stream of step (2) uses that buffer)
  void *vTf, *vlp0, *vmu, *vR, *vq, *vG, *vY, *vsq, *vuu, *vzero, *vinfo2;
  const int64_t ldr = layout == BLR_LAYOUT_COLVECS ? SP : N;
  {
    const size_t sizes[11] = {(size_t)D * D * sizeof(T), sizeof(double), (size_t)N * sizeof(T), (size_t)SP * N * sizeof(T),
                              (size_t)NP64 * SP * sizeof(double), (size_t)nsplit * ntiles * kPB * kPB * sizeof(T),
                              (size_t)ldy * DP * sizeof(T), (size_t)SP * sizeof(double), (size_t)SP * sizeof(T), sizeof(T), sizeof(int32_t)};
    void** const outs[11] = {&vTf, &vlp0, &vmu, &vR, &vq, &vG, &vY, &vsq, &vuu, &vzero, &vinfo2};
    size_t total = 0;
    for (size_t b : sizes) total += (b + 255) & ~(size_t)255;
    if ((rc = ensure_aux(h, total))) return rc;
    size_t off = 0;
    for (int i = 0; i < 11; ++i) {
      *outs[i] = h->aux + off;
      off += (sizes[i] + 255) & ~(size_t)255;
    }
  }
  T* Tf = static_cast<T*>(vTf);
  double* lp0 = static_cast<double*>(vlp0);
  T* mu = static_cast<T*>(vmu);
  T* R = static_cast<T*>(vR);
  double* qpart = static_cast<double*>(vq);
  T* Gpart = static_cast<T*>(vG);
  T* Ybar = static_cast<T*>(vY);
  double* rowsq = static_cast<double*>(vsq);
  T* uu = static_cast<T*>(vuu);
  T* zero = static_cast<T*>(vzero);
  int32_t* info2 = static_cast<int32_t*>(vinfo2);
  HIP_TRY(h, hipMemsetAsync(zero, 0, sizeof(T), h->stream));
  if (layout == BLR_LAYOUT_COLVECS) HIP_TRY(h, hipMemsetAsync(R, 0, (size_t)SP * N * sizeof(T), h->stream));

  // (1) the ordinary fused update on column 0: factor T, logpdf_0, status
  {
    PosteriorArgs<T> a{};
    a.X = X_d; a.ldx = ldx; a.strideX = 0; a.y = Y_d; a.stridey = 0; a.s = s_d; a.strides = 0; a.mw = mw_d; a.stridemw = 0;
    a.Lw = Lw_d; a.ldl = ldl; a.strideLw = 0;
    a.mw_post = nullptr; a.stride_mwpost = D; a.T_post = Tf; a.ldt = D; a.strideT = D * D;
    a.Lw_post = nullptr; a.ldlp = D; a.strideLp = 0; a.logpdf = lp0; a.info = info_d;
    a.layout = layout; a.noise_kind = noise_kind; a.prior_kind = prior_kind; a.D = (int)D; a.N = (int)N; a.B = 1;
    a.vec_ok = (layout == BLR_LAYOUT_COLVECS && D % Mfma<T>::VEC == 0 && aligned16(X_d, ldx, 0)) ? 1 : 0;
    if ((rc = dispatch_posterior<T>(h, a))) return rc;
  }
  // (2) mu_n = x_n'mw: the mean-only marginal stream
  {
    const bool was_async = h->async;
    h->async = true;
    rc = marginals_batched<T>(h, BLR_MEM_DEVICE, layout, 1, D, N, X_d, ldx, 0, noise_kind, s_d, 0, BLR_PRIOR_DIAGONAL, mw_d, 0,
                              nullptr, 1, 0, mu, N, nullptr, N, info2);
    h->async = was_async;
    if (rc) return rc;
  }
  // (3) residuals R = S (Y - mu 1') in X's layout, q partials
  {
    MultiPrepArgs<T> p{};
    p.Y = Y_d; p.ldY = ldY; p.mu = mu; p.s = s_d; p.noise_kind = noise_kind; p.R = R; p.ldr = ldr; p.layout = layout;
    p.qpart = qpart; p.N = (int)N; p.S = (int)S; p.SP = SP;
    hipLaunchKernelGGL(multi_prep_kernel<T>, dim3((unsigned)NP64), dim3(kThreads), 0, h->stream, p);
  }
  // (4) B' = R'X': split-K MFMA tiles, first operand R (rows s), second operand X (rows d)
  if ((rc = set_lds<T>(h, reinterpret_cast<const void*>(gram_tile_kernel<T>), LC::LDS_BYTES))) return rc;
  {
    GramTileArgs<T> g{};
    g.X = R; g.ldx = ldr; g.layout = layout; g.D = SP;
    g.XB = X_d; g.ldxb = ldx; g.DB = (int)D;
    g.use_dma = (layout == BLR_LAYOUT_COLVECS && ((uintptr_t)X_d % 16 == 0) && ((ldx * (int64_t)sizeof(T)) % 16 == 0)) ? 1 : 0;
    g.s = nullptr; g.noise_kind = NOISE_ISOTROPIC; g.r = nullptr;
    g.n_begin = 0; g.n_end = (int)N; g.nsplit = nsplit;
    g.tile_i0 = 0; g.tile_j0 = 0; g.tri = 3; g.ntile_rows = ntile_rows; g.ntiles = ntiles; g.nblocks = NC;
    g.Gpart = Gpart; g.bpart = nullptr; g.mode_out = 0; g.xcd_swizzle = 0;
    hipLaunchKernelGGL(gram_tile_kernel<T>, dim3(ntiles * nsplit), dim3(kThreads), LC::LDS_BYTES, h->stream, g);
    hipLaunchKernelGGL(multi_reduce_kernel<T>, dim3(ntiles, 16), dim3(kThreads), 0, h->stream, (const T*)Gpart, nsplit, ntiles,
                       ntile_rows, Ybar, ldy, DP);
  }
  // (5) rows b_s' -> b_s'L^-T (|u_s|^2 rides along) -> (means only) b_s'A^-1
  {
    dim3 grid((DP + 31) / 32, (DP + 31) / 32);
    hipLaunchKernelGGL(factor_sym_fill_kernel<T>, grid, dim3(kThreads), 0, h->stream, (const T*)Tf, D, (int)D, DP, Ybar, ldy);
  }
  if ((rc = set_lds<T>(h, reinterpret_cast<const void*>(trsm_block_kernel<T>), TC::LDS_BYTES))) return rc;
  if ((rc = set_lds<T>(h, reinterpret_cast<const void*>(trsm_back_block_kernel<T>), TC::LDS_BYTES))) return rc;
  const int R_rows = DP + SP;
  const int nblk = (SP + TC::RB - 1) / TC::RB;
  auto trailing = [&](int p, int j0, int ncolblocks) {
    GramTileArgs<T> g{};
    g.X = Ybar + (int64_t)p * kPB * ldy; g.ldx = ldy; g.layout = LAYOUT_COLVECS; g.use_dma = 1;
    g.s = nullptr; g.noise_kind = NOISE_ISOTROPIC; g.r = nullptr;
    g.D = R_rows; g.n_begin = 0; g.n_end = kPB; g.nsplit = 1;
    g.tile_i0 = NC; g.tile_j0 = j0; g.tri = 3; g.ntile_rows = ntile_rows; g.ntiles = ntile_rows * ncolblocks; g.nblocks = NC + ntile_rows;
    g.C = Ybar; g.ldc = ldy; g.mode_out = 1;
    hipLaunchKernelGGL(gram_tile_kernel<T>, dim3(g.ntiles), dim3(kThreads), LC::LDS_BYTES, h->stream, g);
  };
  for (int p = 0; p < NC; ++p) {
    RowSqArgs<T> rs{};
    rs.acc = rowsq; rs.var = uu; rs.s = zero; rs.noise_kind = BLR_NOISE_ISOTROPIC; rs.N = (int)S; rs.first = p == 0; rs.last = p == NC - 1;
    hipLaunchKernelGGL(trsm_block_kernel<T>, dim3(nblk), dim3(kThreads), TC::LDS_BYTES, h->stream, Ybar, ldy, p, DP, R_rows,
                       (const int32_t*)info_d, rs);
    if (p + 1 < NC) trailing(p, p + 1, NC - 1 - p);
  }
  hipLaunchKernelGGL(multi_finish_kernel<T>, dim3((unsigned)S), dim3(kThreads), 0, h->stream,
                     (const double*)lp0, (const double*)qpart, NP64, SP, (const T*)uu, (int)S, lp_d);
  if (mp_d) {
    for (int p = NC - 1; p >= 0; --p) {
      hipLaunchKernelGGL(trsm_back_block_kernel<T>, dim3(nblk), dim3(kThreads), TC::LDS_BYTES, h->stream, Ybar, ldy, p, DP, R_rows,
                         (const int32_t*)info_d);
      if (p > 0) trailing(p, 0, p);
    }
    hipLaunchKernelGGL(multi_means_kernel<T>, dim3(1024), dim3(kThreads), 0, h->stream, (const T*)Ybar, ldy, DP, mw_d, (int)D, (int)S,
                       mp_d, ldmp);
  }
  HIP_TRY(h, hipGetLastError());
  if (memspace == BLR_MEM_HOST) {
    HIP_TRY(h, hipMemcpyAsync(logpdf, lp_d, (size_t)S * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (mw_post) HIP_TRY(h, hipMemcpyAsync(mw_post, mp_d, mat_extent(D, S, ldmp) * sizeof(T), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(info, info_d, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  } else if (!h->async) {
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  }
  return 0;
}

// D > 128: W[:, s] = mw + U^-1 Z[:, s] with U = chol(Lw).U -- the wavefront back substitution of the posterior path with
// one grid column per draw (reference :46-52).  All pointers are device pointers.
template <typename T>
int sample_weights_large(blr_handle* h, int64_t D, int64_t S, int prior_kind, const T* mw, const T* Lw, int64_t ldl,
                         const T* Z, int64_t ldz, T* W, int64_t ldw) {
  if (prior_kind == BLR_PRIOR_DIAGONAL) {
    hipLaunchKernelGGL(diag_sample_kernel<T>, dim3(1024), dim3(kThreads), 0, h->stream, mw, Lw, Z, ldz, W, ldw, (int)D, S);
    HIP_TRY(h, hipGetLastError());
    return 0;
  }
  const int DP = (int)((D + kPB - 1) / kPB * kPB), NC = DP / kPB;
  const int64_t chunk = std::min<int64_t>(S, 1024);
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
  const size_t o_tf = carve((size_t)DP * DP * sizeof(T));
  const size_t o_wk = carve(prior_kind == BLR_PRIOR_DENSE ? (size_t)DP * DP * sizeof(T) : 0);
  const size_t o_info = carve(64);
  int rc = ensure_ws(h, off);
  if (rc) return rc;
  char* ws = h->ws;
  T* Tf = reinterpret_cast<T*>(ws + o_tf);
  int32_t* info_dev = reinterpret_cast<int32_t*>(ws + o_info);
  if (prior_kind == BLR_PRIOR_UPPER_FACTOR) {
    hipLaunchKernelGGL(upper_pad_kernel<T>, dim3(1024), dim3(kThreads), 0, h->stream, Lw, ldl, (int)D, DP, Tf, (int64_t)DP);
  } else {
    T* Wk = reinterpret_cast<T*>(ws + o_wk);
    HIP_TRY(h, hipMemsetAsync(info_dev, 0, 64, h->stream));
    hipLaunchKernelGGL(prior_copy_kernel<T>, dim3(1024), dim3(kThreads), 0, h->stream, Lw, ldl, (int)D, DP, Wk, (int64_t)DP);
    if ((rc = chol_large<T>(h, Wk, DP, DP, DP, info_dev))) return rc;
    dim3 grid((DP + 31) / 32, (DP + 31) / 32);
    hipLaunchKernelGGL(transpose_out_kernel<T>, grid, dim3(kThreads), 0, h->stream, (const T*)Wk, (int64_t)DP, DP, Tf,
                       (int64_t)DP, (T*)nullptr, (int64_t)0, 0);
    int32_t hinfo = 0;
    HIP_TRY(h, hipMemcpyAsync(&hinfo, info_dev, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (hinfo != 0) return hinfo;  // the prior precision is not positive definite
  }
  for (int64_t s0 = 0; s0 < S; s0 += chunk) {
    const int64_t ns = std::min(chunk, S - s0);
    WaveSolveArgs<T> b{};
    b.Tf = Tf; b.ldtf = DP; b.D = (int)D; b.DP = DP;
    b.rhs = Z + s0 * ldz; b.ldrhs = ldz; b.rhs_inc = 1;
    b.add = mw; b.out = W + s0 * ldw; b.ldout = ldw;
    if ((rc = launch_wave_solve<T>(h, b, NC, ns))) return rc;
  }
  HIP_TRY(h, hipGetLastError());
  return 0;
}

template <typename T>
int sample_weights_impl(blr_handle* h, int memspace, int64_t D, int64_t S, int prior_kind, const T* mw, const T* Lw,
                        int64_t ldl, const T* Z, int64_t ldz, T* W, int64_t ldw, bool sync_and_copy, T** W_dev_out,
                        Staging* outer) {
  (void)outer;
  const T *mw_d = mw, *Lw_d = Lw, *Z_d = Z;
  T* W_d = W;
  int rc;
  if (memspace == BLR_MEM_HOST) {
    const size_t lw_one = prior_kind == BLR_PRIOR_DIAGONAL ? (size_t)D : mat_extent(D, D, ldl);
    if ((rc = stage_in(h, mw, (size_t)D, &mw_d))) return rc;
    if ((rc = stage_in(h, Lw, lw_one, &Lw_d))) return rc;
    if ((rc = stage_in(h, Z, mat_extent(D, S, ldz), &Z_d))) return rc;
    if (W) {
      if ((rc = stage_out_alloc(h, W, mat_extent(D, S, ldw), &W_d))) return rc;
    }
  }
  if (!W_d) {  // internal temporary (rand): dense D x S, in the handle's grow-only side buffer -- a hipMalloc / hipFree pair and
               // the drain in front of the free were most of a small call (64 draws at D = 128: 68 us, the kernels 20)
    if ((rc = ensure_aux(h, (size_t)D * S * sizeof(T)))) return rc;
    W_d = reinterpret_cast<T*>(h->aux);
    ldw = D;
  }
  if (D > kMaxSmallD) {
    if ((rc = sample_weights_large<T>(h, D, S, prior_kind, mw_d, Lw_d, ldl, Z_d, ldz, W_d, ldw))) return rc;
    if (W_dev_out) *W_dev_out = W_d;
    if (sync_and_copy) {
      if (memspace == BLR_MEM_HOST) {
        HIP_TRY(h, hipMemcpyAsync(W, W_d, mat_extent(D, S, ldw) * sizeof(T), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
      } else if (!h->async) {
        HIP_TRY(h, hipStreamSynchronize(h->stream));
      }
    }
    return 0;
  }
  const T* U;
  int64_t ldu, strideU;
  int kind;
  int32_t* chol_info;
  if ((rc = prior_factor<T>(h, 1, D, prior_kind, Lw_d, ldl, 0, &U, &ldu, &strideU, &kind, &chol_info))) return rc;
  int32_t hinfo = 0;
  if (chol_info) {
    HIP_TRY(h, hipMemcpyAsync(&hinfo, chol_info, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (hinfo != 0) return hinfo;
  }
  if (kind == BLR_PRIOR_DIAGONAL) {
    hipLaunchKernelGGL(diag_sample_kernel<T>, dim3(1024), dim3(kThreads), 0, h->stream, mw_d, U, Z_d, ldz, W_d, ldw, (int)D, S);
  } else {  // draws as rows of an LDS block, one backward MFMA sweep per tile of draws
    using TC = TrsmCfg<T>;
    const int lds = TC::LDS_BYTES + kPB * (int)sizeof(T);
    auto kern = sample_weights_mfma_kernel<T>;
    { const int rc_lds = set_lds_once(h, reinterpret_cast<const void*>(kern), (size_t)(lds)); if (rc_lds) return rc_lds; }
    const int64_t ntiles = (S + TC::RB - 1) / TC::RB;
    hipLaunchKernelGGL(kern, dim3((unsigned)std::min<int64_t>(ntiles, 512)), dim3(kThreads), lds, h->stream, mw_d, U, ldu, Z_d, ldz,
                       W_d, ldw, (int)D, S);
  }
  HIP_TRY(h, hipGetLastError());
  if (W_dev_out) *W_dev_out = W_d;
  if (sync_and_copy) {
    if (memspace == BLR_MEM_HOST) {
      HIP_TRY(h, hipMemcpyAsync(W, W_d, mat_extent(D, S, ldw) * sizeof(T), hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(h, hipStreamSynchronize(h->stream));
    } else if (!h->async) {
      HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
  }
  return 0;
}

template <typename T>
int sample_weights(blr_handle* h, int memspace, int64_t D, int64_t S, int prior_kind, const T* mw, const T* Lw,
                   int64_t ldl, const T* Z, int64_t ldz, T* W, int64_t ldw) {
  if (!h) return -1;
  h->err.clear();
  if (memspace != BLR_MEM_HOST && memspace != BLR_MEM_DEVICE) return bad_arg(h, 2, "memspace");
  if (D < 1 || D > kMaxLargeD) return bad_arg(h, 3, "D out of range for this build (1..8192)");
  if (S < 0) return bad_arg(h, 4, "S < 0");
  if (S == 0) return 0;
  if (prior_kind != BLR_PRIOR_DENSE && prior_kind != BLR_PRIOR_UPPER_FACTOR && prior_kind != BLR_PRIOR_DIAGONAL)
    return bad_arg(h, 5, "prior_kind");
  if (!mw) return bad_arg(h, 6, "mw is NULL");
  if (!Lw) return bad_arg(h, 7, "Lw is NULL");
  if (prior_kind != BLR_PRIOR_DIAGONAL && ldl < D) return bad_arg(h, 8, "ldl < D");
  if (!Z) return bad_arg(h, 9, "Z is NULL");
  if (ldz < D) return bad_arg(h, 10, "ldz < D");
  if (!W) return bad_arg(h, 11, "W is NULL");
  if (ldw < D) return bad_arg(h, 12, "ldw < D");
  HIP_TRY(h, hipSetDevice(h->device));
  Staging guard(h);
  return sample_weights_impl<T>(h, memspace, D, S, prior_kind, mw, Lw, ldl, Z, ldz, W, ldw, true, nullptr, &guard);
}

// Y (N x S) = X'W (+ sqrt.(s) .* Z2 when Z2 != NULL): device pointers
template <typename T>
void launch_project(blr_handle* h, int layout, int64_t D, int64_t N, int64_t S, const T* X_d, int64_t ldx, const T* W_d, int64_t ldw,
                    const T* s_d, int noise_kind, const T* Z2_d, int64_t ldz2, T* Y_d, int64_t ldy) {
  constexpr int kVec = Mfma<T>::VEC;
  const bool no_mfma_proj = h->opt.no_mfma_project;  // A/B experiments only
  if (!no_mfma_proj && layout == BLR_LAYOUT_COLVECS && D % kVec == 0 && aligned16(X_d, ldx, 0) && aligned16(W_d, ldw, 0)) {
    // tall-skinny GEMM on the matrix cores
    using PC = ProjCfg<T>;
    dim3 grid((unsigned)((N + PC::TN - 1) / PC::TN), (unsigned)((S + PC::TS - 1) / PC::TS));
    hipLaunchKernelGGL(rand_project_mfma_kernel<T>, grid, dim3(kThreads), PC::LDS_BYTES, h->stream, X_d, ldx, W_d, ldw, s_d, noise_kind,
                       Z2_d, ldz2, Y_d, ldy, (int)D, (int)N, S);
  } else {
    dim3 grid((unsigned)((N + 63) / 64), (unsigned)((S + 63) / 64));
    hipLaunchKernelGGL(rand_project_kernel<T>, grid, dim3(kThreads), 0, h->stream, X_d, ldx, layout, W_d, ldw, s_d, noise_kind, Z2_d,
                       ldz2, Y_d, ldy, (int)D, (int)N, S);
  }
}

// ---- Y = X'W for S given weight vectors: evaluation of function samples (reference sampling_functions.jl:16-18) ------------
template <typename T>
int apply_weights(blr_handle* h, int memspace, int layout, int64_t D, int64_t N, int64_t S, const T* X, int64_t ldx, const T* W,
                  int64_t ldw, T* Y, int64_t ldy) {
  if (!h) return -1;
  h->err.clear();
  if (memspace != BLR_MEM_HOST && memspace != BLR_MEM_DEVICE) return bad_arg(h, 2, "memspace");
  if (layout != BLR_LAYOUT_COLVECS && layout != BLR_LAYOUT_ROWVECS) return bad_arg(h, 3, "unknown layout (reference :26-31)");
  if (D < 1 || D > (1 << 30)) return bad_arg(h, 4, "D out of range");
  if (N < 0 || N > (1 << 30)) return bad_arg(h, 5, "N out of range");
  if (S < 0 || S > (1 << 30)) return bad_arg(h, 6, "S out of range");
  if (N == 0 || S == 0) return 0;
  if (!X) return bad_arg(h, 7, "X is NULL");
  if (layout == BLR_LAYOUT_COLVECS ? ldx < D : ldx < N) return bad_arg(h, 8, "ldx too small");
  if (!W) return bad_arg(h, 9, "W is NULL");
  if (ldw < D) return bad_arg(h, 10, "ldw < D");
  if (!Y) return bad_arg(h, 11, "Y is NULL");
  if (ldy < N) return bad_arg(h, 12, "ldy < N");
  HIP_TRY(h, hipSetDevice(h->device));
  Staging guard(h);
  const T *X_d = X, *W_d = W;
  T* Y_d = Y;
  int rc;
  if (memspace == BLR_MEM_HOST) {
    const size_t x_one = layout == BLR_LAYOUT_COLVECS ? mat_extent(D, N, ldx) : mat_extent(N, D, ldx);
    if ((rc = stage_in(h, X, x_one, &X_d))) return rc;
    if ((rc = stage_in(h, W, mat_extent(D, S, ldw), &W_d))) return rc;
    if ((rc = stage_out_alloc(h, Y, mat_extent(N, S, ldy), &Y_d))) return rc;
  }
  launch_project<T>(h, layout, D, N, S, X_d, ldx, W_d, ldw, (const T*)nullptr, BLR_NOISE_ISOTROPIC, (const T*)nullptr, 0, Y_d, ldy);
  HIP_TRY(h, hipGetLastError());
  if (memspace == BLR_MEM_HOST) {
    HIP_TRY(h, hipMemcpyAsync(Y, Y_d, mat_extent(N, S, ldy) * sizeof(T), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  } else if (!h->async) {
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  }
  return 0;
}

template <typename T>
int rand_impl(blr_handle* h, int memspace, int layout, int64_t D, int64_t N, int64_t S, const T* X, int64_t ldx,
              int noise_kind, const T* s, int prior_kind, const T* mw, const T* Lw, int64_t ldl, const T* Z1,
              int64_t ldz1, const T* Z2, int64_t ldz2, T* Y, int64_t ldy) {
  if (!h) return -1;
  h->err.clear();
  if (memspace != BLR_MEM_HOST && memspace != BLR_MEM_DEVICE) return bad_arg(h, 2, "memspace");
  if (layout != BLR_LAYOUT_COLVECS && layout != BLR_LAYOUT_ROWVECS) return bad_arg(h, 3, "unknown layout (reference :26-31)");
  if (D < 1 || D > kMaxLargeD) return bad_arg(h, 4, "D out of range for this build (1..8192)");
  if (N < 0 || N > (1 << 30)) return bad_arg(h, 5, "N out of range");
  if (S < 0) return bad_arg(h, 6, "S < 0");
  if (N == 0 || S == 0) return 0;
  if (!X) return bad_arg(h, 7, "X is NULL");
  if (layout == BLR_LAYOUT_COLVECS ? ldx < D : ldx < N) return bad_arg(h, 8, "ldx too small");
  if (noise_kind != BLR_NOISE_ISOTROPIC && noise_kind != BLR_NOISE_DIAGONAL) return bad_arg(h, 9, "noise_kind");
  if (!s) return bad_arg(h, 10, "s is NULL");
  if (prior_kind != BLR_PRIOR_DENSE && prior_kind != BLR_PRIOR_UPPER_FACTOR && prior_kind != BLR_PRIOR_DIAGONAL)
    return bad_arg(h, 11, "prior_kind");
  if (!mw) return bad_arg(h, 12, "mw is NULL");
  if (!Lw) return bad_arg(h, 13, "Lw is NULL");
  if (prior_kind != BLR_PRIOR_DIAGONAL && ldl < D) return bad_arg(h, 14, "ldl < D");
  if (!Z1) return bad_arg(h, 15, "Z1 is NULL");
  if (ldz1 < D) return bad_arg(h, 16, "ldz1 < D");
  if (!Z2) return bad_arg(h, 17, "Z2 is NULL");
  if (ldz2 < N) return bad_arg(h, 18, "ldz2 < N");
  if (!Y) return bad_arg(h, 19, "Y is NULL");
  if (ldy < N) return bad_arg(h, 20, "ldy < N");
  HIP_TRY(h, hipSetDevice(h->device));
  Staging guard(h);
  T* W_dev = nullptr;
  // weights first (Z1 is the FIRST randn draw of reference :51), host staging handled inside
  int rc = sample_weights_impl<T>(h, memspace, D, S, prior_kind, mw, Lw, ldl, Z1, ldz1, (T*)nullptr, D, false, &W_dev, &guard);
  if (rc) return rc;
  const T *X_d = X, *s_d = s, *Z2_d = Z2;
  T* Y_d = Y;
  if (memspace == BLR_MEM_HOST) {
    const size_t x_one = layout == BLR_LAYOUT_COLVECS ? mat_extent(D, N, ldx) : mat_extent(N, D, ldx);
    if ((rc = stage_in(h, X, x_one, &X_d))) return rc;
    if ((rc = stage_in(h, s, noise_kind == BLR_NOISE_DIAGONAL ? (size_t)N : 1, &s_d))) return rc;
    if ((rc = stage_in(h, Z2, mat_extent(N, S, ldz2), &Z2_d))) return rc;
    if ((rc = stage_out_alloc(h, Y, mat_extent(N, S, ldy), &Y_d))) return rc;
  }
  launch_project<T>(h, layout, D, N, S, X_d, ldx, (const T*)W_dev, (int64_t)D, s_d, noise_kind, Z2_d, ldz2, Y_d, ldy);
  HIP_TRY(h, hipGetLastError());
  if (memspace == BLR_MEM_HOST) {
    HIP_TRY(h, hipMemcpyAsync(Y, Y_d, mat_extent(N, S, ldy) * sizeof(T), hipMemcpyDeviceToHost, h->stream));
  }
  // (W_dev lives in the handle's side buffer, which is only ever replaced behind a stream synchronisation)
  if (memspace == BLR_MEM_HOST || !h->async) HIP_TRY(h, hipStreamSynchronize(h->stream));
  return 0;
}


template <typename T>
int rff_features(blr_handle* h, int memspace, int64_t Din, int64_t D, int64_t N, const T* Xin, int64_t ldxin,
                 const T* Omega, int64_t ldo, const T* phase, T scale, T* Phi, int64_t ldphi) {
  if (!h) return -1;
  h->err.clear();
  if (memspace != BLR_MEM_HOST && memspace != BLR_MEM_DEVICE) return bad_arg(h, 2, "memspace");
  if (Din < 1) return bad_arg(h, 3, "Din < 1");
  if (D < 1) return bad_arg(h, 4, "D < 1");
  if (N < 0 || N > (1 << 30)) return bad_arg(h, 5, "N out of range");
  if (N == 0) return 0;
  if (!Xin) return bad_arg(h, 6, "Xin is NULL");
  if ((N + 15) / 16 > 65535) return bad_arg(h, 5, "N > 1048560 inputs per feature-map call (launch geometry)");
  if (ldxin < Din) return bad_arg(h, 7, "ldxin < Din");
  if (!Omega) return bad_arg(h, 8, "Omega is NULL");
  if (ldo < Din) return bad_arg(h, 9, "ldo < Din");
  if (!phase) return bad_arg(h, 10, "phase is NULL");
  if (!Phi) return bad_arg(h, 12, "Phi is NULL");
  if (ldphi < D) return bad_arg(h, 13, "ldphi < D");
  HIP_TRY(h, hipSetDevice(h->device));
  Staging guard(h);
  const T *Xd = Xin, *Od = Omega, *Pd = phase;
  T* Fd = Phi;
  int rc;
  if (memspace == BLR_MEM_HOST) {
    if ((rc = stage_in(h, Xin, mat_extent(Din, N, ldxin), &Xd))) return rc;
    if ((rc = stage_in(h, Omega, mat_extent(Din, D, ldo), &Od))) return rc;
    if ((rc = stage_in(h, phase, (size_t)D, &Pd))) return rc;
    if ((rc = stage_out_alloc(h, Phi, mat_extent(D, N, ldphi), &Fd))) return rc;
  }
  dim3 grid((unsigned)((D + kThreads - 1) / kThreads), (unsigned)((N + 31) / 32));  // 32 columns per workgroup (rff_features_kernel NT)
  hipLaunchKernelGGL(rff_features_kernel<T>, grid, dim3(kThreads), 0, h->stream, Xd, ldxin, Od, ldo, Pd, scale, (int)Din,
                     (int)D, (int)N, Fd, ldphi);
  HIP_TRY(h, hipGetLastError());
  if (memspace == BLR_MEM_HOST) {
    HIP_TRY(h, hipMemcpyAsync(Phi, Fd, mat_extent(D, N, ldphi) * sizeof(T), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  } else if (!h->async) {
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  }
  return 0;
}

template <typename T>
int posterior_rff(blr_handle* h, int memspace, int64_t Din, int64_t D, int64_t N, const T* Xin, int64_t ldxin,
                  const T* Omega, int64_t ldo, const T* phase, T scale, const T* y, int noise_kind, const T* s,
                  int prior_kind, const T* mw, const T* Lw, int64_t ldl, T* mw_post, T* T_post, int64_t ldt, T* Lw_post,
                  int64_t ldlp, double* logpdf, int32_t* info) {
  if (!h) return -1;
  h->err.clear();
  if (memspace != BLR_MEM_HOST && memspace != BLR_MEM_DEVICE) return bad_arg(h, 2, "memspace");
  if (Din < 1) return bad_arg(h, 3, "Din < 1");
  if (D < 1 || D > kMaxLargeD) return bad_arg(h, 4, "D out of range");
  if (N < 0 || N > (1 << 30)) return bad_arg(h, 5, "N out of range");
  if (N > 0 && !Xin) return bad_arg(h, 6, "Xin is NULL");
  if (ldxin < Din) return bad_arg(h, 7, "ldxin < Din");
  if (!Omega) return bad_arg(h, 8, "Omega is NULL");
  if (ldo < Din) return bad_arg(h, 9, "ldo < Din");
  if (!phase) return bad_arg(h, 10, "phase is NULL");
  HIP_TRY(h, hipSetDevice(h->device));
  // fp32 at D > 128: the basis is never materialised -- the planes pass of the large-D pipeline evaluates phi once per element and
  // writes the bf16 planes of the Gram operands directly (blr_planes.hpp; reference src/basis_function_regression.jl:41 builds phi(x))
  if (sizeof(T) == 4 && D > kMaxSmallD && N > 0 && !h->opt.no_planes && !h->opt.no_bf16x3) {
    Staging guard(h);
    const T *Xd = Xin, *Od = Omega, *Pd = phase;
    const T *yd = y, *sd = s, *mwd = mw, *Lwd = Lw;
    T *mwp = mw_post, *Tp = T_post, *Ap = Lw_post;
    double* lpd = logpdf;
    int32_t* infod = info;
    int rc;
    const bool host = memspace == BLR_MEM_HOST;
    if (host) {
      const size_t lw_one = prior_kind == BLR_PRIOR_DIAGONAL ? (size_t)D : mat_extent(D, D, ldl);
      if (noise_kind != BLR_NOISE_ISOTROPIC && noise_kind != BLR_NOISE_DIAGONAL) return bad_arg(h, 13, "noise_kind");
      if (!y) return bad_arg(h, 12, "y is NULL");
      if (!s) return bad_arg(h, 14, "s is NULL");
      if (!mw) return bad_arg(h, 16, "mw is NULL");
      if (!Lw) return bad_arg(h, 17, "Lw is NULL");
      if (!info) return bad_arg(h, 25, "info is NULL");
      if ((rc = stage_in(h, Xin, mat_extent(Din, N, ldxin), &Xd))) return rc;
      if ((rc = stage_in(h, Omega, mat_extent(Din, D, ldo), &Od))) return rc;
      if ((rc = stage_in(h, phase, (size_t)D, &Pd))) return rc;
      if ((rc = stage_in(h, y, (size_t)N, &yd))) return rc;
      if ((rc = stage_in(h, s, noise_kind == BLR_NOISE_DIAGONAL ? (size_t)N : 1, &sd))) return rc;
      if ((rc = stage_in(h, mw, (size_t)D, &mwd))) return rc;
      if ((rc = stage_in(h, Lw, lw_one, &Lwd))) return rc;
      if ((rc = stage_out_alloc(h, mw_post, (size_t)D, &mwp))) return rc;
      if ((rc = stage_out_alloc(h, T_post, mat_extent(D, D, ldt), &Tp))) return rc;
      if ((rc = stage_out_alloc(h, Lw_post, mat_extent(D, D, ldlp), &Ap))) return rc;
      if ((rc = stage_out_alloc(h, logpdf, 1, &lpd))) return rc;
      if ((rc = stage_out_alloc(h, info, 1, &infod))) return rc;
    }
    const blr_handle::RffSrc src{Xd, Od, Pd, ldxin, ldo, (double)scale, (int)Din};
    const bool was_async = h->async;
    if (host) h->async = true;
    h->rff_src = &src;
    // (X: any non-NULL device pointer -- with a basis source attached nothing dereferences it; ldx = D passes the argument checks)
    rc = posterior_batched<T>(h, BLR_MEM_DEVICE, BLR_LAYOUT_COLVECS, 1, D, N, Xd, D, 0, yd, 0, noise_kind, sd, 0, prior_kind, mwd, 0,
                              Lwd, ldl, 0, mwp, D, Tp, ldt, ldt * D, Ap, ldlp, ldlp * D, lpd, infod);
    h->rff_src = nullptr;
    h->async = was_async;
    if (rc) return rc;
    if (host) {
      if (mw_post) HIP_TRY(h, hipMemcpyAsync(mw_post, mwp, (size_t)D * sizeof(T), hipMemcpyDeviceToHost, h->stream));
      if (T_post) HIP_TRY(h, hipMemcpyAsync(T_post, Tp, mat_extent(D, D, ldt) * sizeof(T), hipMemcpyDeviceToHost, h->stream));
      if (Lw_post) HIP_TRY(h, hipMemcpyAsync(Lw_post, Ap, mat_extent(D, D, ldlp) * sizeof(T), hipMemcpyDeviceToHost, h->stream));
      if (logpdf) HIP_TRY(h, hipMemcpyAsync(logpdf, lpd, sizeof(double), hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(h, hipMemcpyAsync(info, infod, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    return 0;
  }
  const int64_t ldphi = (D + 3) / 4 * 4;  // keeps every feature column 16-byte aligned for the LDS-DMA loader
  const size_t need = (size_t)ldphi * (size_t)std::max<int64_t>(N, 1) * sizeof(T);
  if (need > h->feat_bytes) {
    if (h->feat) {
      HIP_TRY(h, hipStreamSynchronize(h->stream));
      HIP_TRY(h, hipFree(h->feat));
      h->feat = nullptr;
      h->feat_bytes = 0;
    }
    HIP_TRY(h, hipMalloc((void**)&h->feat, need));
    h->feat_bytes = need;
  }
  T* Phi = reinterpret_cast<T*>(h->feat);
  const bool was_async = h->async;
  int rc;
  {
    Staging guard(h);
    const T *Xd = Xin, *Od = Omega, *Pd = phase;
    if (memspace == BLR_MEM_HOST) {
      if ((rc = stage_in(h, Xin, mat_extent(Din, N, ldxin), &Xd))) return rc;
      if ((rc = stage_in(h, Omega, mat_extent(Din, D, ldo), &Od))) return rc;
      if ((rc = stage_in(h, phase, (size_t)D, &Pd))) return rc;
    }
    if (N > 0) {
      dim3 grid((unsigned)((D + kThreads - 1) / kThreads), (unsigned)((N + 31) / 32));  // 32 columns per workgroup (rff_features_kernel NT)
      hipLaunchKernelGGL(rff_features_kernel<T>, grid, dim3(kThreads), 0, h->stream, Xd, ldxin, Od, ldo, Pd, scale,
                         (int)Din, (int)D, (int)N, Phi, ldphi);
      HIP_TRY(h, hipGetLastError());
    }
    if (memspace == BLR_MEM_HOST) HIP_TRY(h, hipStreamSynchronize(h->stream));  // staged inputs die with `guard`
  }
  if (memspace == BLR_MEM_DEVICE) {
    return posterior_batched<T>(h, BLR_MEM_DEVICE, BLR_LAYOUT_COLVECS, 1, D, N, Phi, ldphi, 0, y, 0, noise_kind, s, 0,
                                prior_kind, mw, 0, Lw, ldl, 0, mw_post, D, T_post, ldt, ldt * D, Lw_post, ldlp, ldlp * D,
                                logpdf, info);
  }
  // host pointers for everything except Phi: stage the rest here, then run on device pointers
  Staging guard(h);
  const T *yd, *sd, *mwd, *Lwd;
  const size_t lw_one = prior_kind == BLR_PRIOR_DIAGONAL ? (size_t)D : mat_extent(D, D, ldl);
  if (noise_kind != BLR_NOISE_ISOTROPIC && noise_kind != BLR_NOISE_DIAGONAL) return bad_arg(h, 13, "noise_kind");
  if (!y && N > 0) return bad_arg(h, 12, "y is NULL");
  if (!s) return bad_arg(h, 14, "s is NULL");
  if (!mw) return bad_arg(h, 16, "mw is NULL");
  if (!Lw) return bad_arg(h, 17, "Lw is NULL");
  if (!info) return bad_arg(h, 25, "info is NULL");
  if ((rc = stage_in(h, y, (size_t)N, &yd))) return rc;
  if ((rc = stage_in(h, s, noise_kind == BLR_NOISE_DIAGONAL ? (size_t)N : 1, &sd))) return rc;
  if ((rc = stage_in(h, mw, (size_t)D, &mwd))) return rc;
  if ((rc = stage_in(h, Lw, lw_one, &Lwd))) return rc;
  T *mwp = nullptr, *Tp = nullptr, *Ap = nullptr;
  double* lpd = nullptr;
  int32_t* infod = nullptr;
  if ((rc = stage_out_alloc(h, mw_post, (size_t)D, &mwp))) return rc;
  if ((rc = stage_out_alloc(h, T_post, mat_extent(D, D, ldt), &Tp))) return rc;
  if ((rc = stage_out_alloc(h, Lw_post, mat_extent(D, D, ldlp), &Ap))) return rc;
  if ((rc = stage_out_alloc(h, logpdf, 1, &lpd))) return rc;
  if ((rc = stage_out_alloc(h, info, 1, &infod))) return rc;
  h->async = true;
  rc = posterior_batched<T>(h, BLR_MEM_DEVICE, BLR_LAYOUT_COLVECS, 1, D, N, Phi, ldphi, 0, yd ? yd : mwd, 0, noise_kind, sd,
                            0, prior_kind, mwd, 0, Lwd, ldl, 0, mwp, D, Tp, ldt, ldt * D, Ap, ldlp, ldlp * D, lpd, infod);
  h->async = was_async;
  if (rc) return rc;
  if (mw_post) HIP_TRY(h, hipMemcpyAsync(mw_post, mwp, (size_t)D * sizeof(T), hipMemcpyDeviceToHost, h->stream));
  if (T_post) HIP_TRY(h, hipMemcpyAsync(T_post, Tp, mat_extent(D, D, ldt) * sizeof(T), hipMemcpyDeviceToHost, h->stream));
  if (Lw_post) HIP_TRY(h, hipMemcpyAsync(Lw_post, Ap, mat_extent(D, D, ldlp) * sizeof(T), hipMemcpyDeviceToHost, h->stream));
  if (logpdf) HIP_TRY(h, hipMemcpyAsync(logpdf, lpd, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipMemcpyAsync(info, infod, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return 0;
}


// ================================================================================================================
// Dense Sigma_y and full predictive covariance: see blr_dense.hpp for the scheme
// ================================================================================================================
constexpr int64_t kMaxDenseN = 16384;  // N x N work matrices: 2 GiB in fp64 at this size

template <typename T>
int dev_alloc_tmp(blr_handle* h, size_t count, T** out) {  // freed by the enclosing Staging guard
  void* p = nullptr;
  HIP_TRY(h, hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
  h->staged.push_back(p);
  *out = static_cast<T*>(p);
  return 0;
}

// x_n' mw for D <= 128 (one thread per input; the mean of mean_and_cov -- N <= 16384, not a hot path)
template <typename T>
__global__ __launch_bounds__(kThreads) void mean_small_kernel(const T* __restrict__ X, int64_t ldx, int layout, const T* __restrict__ mw, int D,
                                                              int N, T* __restrict__ mean) {
  const int n = blockIdx.x * kThreads + threadIdx.x;
  if (n >= N) return;
  double acc = 0.0;
  for (int d = 0; d < D; ++d)
    acc += (double)((layout == LAYOUT_COLVECS) ? X[(int64_t)n * ldx + d] : X[(int64_t)d * ldx + n]) * (double)mw[d];
  mean[n] = (T)acc;
}

// Sigma_y = L L' in the top NP x NP block of M (lower, unit padding), R extra rows carried through; info_dev: status
template <typename T>
int chol_noise(blr_handle* h, const T* Sy_dev, int64_t ldsy, int N, int NP, int R, T* M, int64_t ld, int32_t* info_dev) {
  HIP_TRY(h, hipMemsetAsync(info_dev, 0, sizeof(int32_t), h->stream));
  hipLaunchKernelGGL(prior_copy_kernel<T>, dim3(2048), dim3(kThreads), 0, h->stream, Sy_dev, ldsy, N, NP, M, ld);
  return chol_large<T>(h, M, ld, NP, NP + R, info_dev);
}

template <typename T>
int posterior_dense_noise(blr_handle* h, int memspace, int layout, int64_t D64, int64_t N64, const T* X, int64_t ldx, const T* y,
                          const T* Sy, int64_t ldsy, int prior_kind, const T* mw, const T* Lw, int64_t ldl, T* mw_post, T* T_post,
                          int64_t ldt, T* Lw_post, int64_t ldlp, double* logpdf, int32_t* info) {
  if (!h) return -1;
  h->err.clear();
  if (memspace != BLR_MEM_HOST && memspace != BLR_MEM_DEVICE) return bad_arg(h, 2, "memspace");
  if (layout != BLR_LAYOUT_COLVECS && layout != BLR_LAYOUT_ROWVECS) return bad_arg(h, 3, "unknown layout (reference :26-31)");
  if (D64 < 1 || D64 > kMaxLargeD) return bad_arg(h, 4, "D out of range for this build (1..8192)");
  if (N64 < 1 || N64 > kMaxDenseN) return bad_arg(h, 5, "N out of range for a dense noise covariance (1..16384)");
  if (!X) return bad_arg(h, 6, "X is NULL");
  if (layout == BLR_LAYOUT_COLVECS ? ldx < D64 : ldx < N64) return bad_arg(h, 7, "ldx too small");
  if (!y) return bad_arg(h, 8, "y is NULL (reference :74 length check)");
  if (!Sy) return bad_arg(h, 9, "Sy is NULL");
  if (ldsy < N64) return bad_arg(h, 10, "ldsy < N");
  if (prior_kind != BLR_PRIOR_DENSE && prior_kind != BLR_PRIOR_UPPER_FACTOR && prior_kind != BLR_PRIOR_DIAGONAL)
    return bad_arg(h, 11, "prior_kind");
  if (!mw) return bad_arg(h, 12, "mw is NULL");
  if (!Lw) return bad_arg(h, 13, "Lw is NULL");
  if (prior_kind != BLR_PRIOR_DIAGONAL && ldl < D64) return bad_arg(h, 14, "ldl < D");
  if (T_post && ldt < D64) return bad_arg(h, 17, "ldt < D");
  if (Lw_post && ldlp < D64) return bad_arg(h, 19, "ldlp < D");
  if (!info) return bad_arg(h, 21, "info is NULL");
  HIP_TRY(h, hipSetDevice(h->device));
  const int D = (int)D64, N = (int)N64;
  const int NP = (N + kPB - 1) / kPB * kPB;
  const int R = (D + 1 + kPB - 1) / kPB * kPB;  // X rows + the y row, padded to whole TRSM row blocks
  const int64_t ld = (int64_t)NP + R;
  Staging guard(h);
  int rc;
  const T *X_d = X, *y_d = y, *Sy_d = Sy, *mw_d = mw, *Lw_d = Lw;
  T *mwp_d = mw_post, *Tp_d = T_post, *Lp_d = Lw_post;
  double* lp_d = logpdf;
  int32_t* info_d = info;
  if (memspace == BLR_MEM_HOST) {
    const size_t x_one = layout == BLR_LAYOUT_COLVECS ? mat_extent(D, N, ldx) : mat_extent(N, D, ldx);
    const size_t lw_one = prior_kind == BLR_PRIOR_DIAGONAL ? (size_t)D : mat_extent(D, D, ldl);
    if ((rc = stage_in(h, X, x_one, &X_d))) return rc;
    if ((rc = stage_in(h, y, (size_t)N, &y_d))) return rc;
    if ((rc = stage_in(h, Sy, mat_extent(N, N, ldsy), &Sy_d))) return rc;
    if ((rc = stage_in(h, mw, (size_t)D, &mw_d))) return rc;
    if ((rc = stage_in(h, Lw, lw_one, &Lw_d))) return rc;
    if ((rc = stage_out_alloc(h, mw_post, (size_t)D, &mwp_d))) return rc;
    if ((rc = stage_out_alloc(h, T_post, mat_extent(D, D, ldt), &Tp_d))) return rc;
    if ((rc = stage_out_alloc(h, Lw_post, mat_extent(D, D, ldlp), &Lp_d))) return rc;
    if ((rc = stage_out_alloc(h, logpdf, (size_t)1, &lp_d))) return rc;
    if ((rc = stage_out_alloc(h, info, (size_t)1, &info_d))) return rc;
  }
  T *M = nullptr, *ytil = nullptr, *one = nullptr;
  double* logdet = nullptr;
  int32_t* noise_info = nullptr;
  if ((rc = dev_alloc_tmp(h, (size_t)ld * NP, &M))) return rc;
  if ((rc = dev_alloc_tmp(h, (size_t)N, &ytil))) return rc;
  if ((rc = dev_alloc_tmp(h, 1, &one))) return rc;
  if ((rc = dev_alloc_tmp(h, 1, &logdet))) return rc;
  if ((rc = dev_alloc_tmp(h, 1, &noise_info))) return rc;
  double* lp_tmp = lp_d;
  if (!lp_tmp && (rc = dev_alloc_tmp(h, 1, &lp_tmp))) return rc;
  const T one_h = T(1);
  HIP_TRY(h, hipMemcpyAsync(one, &one_h, sizeof(T), hipMemcpyHostToDevice, h->stream));
  // [Sigma_y; X; y'] -> [L; X L^-T; (L^-1 y)']   (reference :79, :81 outer solve, :82)
  hipLaunchKernelGGL(whiten_fill_kernel<T>, dim3(2048), dim3(kThreads), 0, h->stream, X_d, ldx, layout, y_d, D, N, NP, R, M, ld, NP);
  if ((rc = chol_noise<T>(h, Sy_d, ldsy, N, NP, R, M, ld, noise_info))) return rc;
  hipLaunchKernelGGL(logdet_kernel<T>, dim3(1), dim3(kThreads), 0, h->stream, (const T*)M, ld, N, logdet);
  hipLaunchKernelGGL(row_extract_kernel<T>, dim3(256), dim3(kThreads), 0, h->stream, (const T*)M, ld, NP + D, N, ytil);
  // the whitened problem: ColVecs X~ = rows NP .. NP+D of M (leading dimension ld), y~, unit isotropic noise
  PosteriorArgs<T> a{};
  a.X = M + NP; a.ldx = ld; a.strideX = 0; a.y = ytil; a.stridey = 0; a.s = one; a.strides = 0;
  a.mw = mw_d; a.stridemw = 0; a.Lw = Lw_d; a.ldl = ldl; a.strideLw = 0;
  a.mw_post = mwp_d; a.stride_mwpost = D; a.T_post = Tp_d; a.ldt = ldt; a.strideT = 0; a.Lw_post = Lp_d; a.ldlp = ldlp; a.strideLp = 0;
  a.logpdf = lp_tmp; a.info = info_d;
  a.layout = BLR_LAYOUT_COLVECS; a.noise_kind = BLR_NOISE_ISOTROPIC; a.prior_kind = prior_kind;
  a.D = D; a.N = N; a.B = 1;
  a.vec_ok = (D % Mfma<T>::VEC == 0 && aligned16(a.X, ld, (int64_t)0)) ? 1 : 0;
  // status precedence of the reference: the prior is factored first (:78), then Sigma_y (:79), then the posterior (:86).  The
  // inner update cannot tell a prior failure from a posterior one once the whitened data are garbage, so the prior's status
  // comes from an update on ZERO observations (A = Lw: D^3 / 3 flops, nothing next to the N^3 / 3 of the whitening)
  int32_t* prior_info = nullptr;
  if ((rc = dev_alloc_tmp(h, 1, &prior_info))) return rc;
  {
    PosteriorArgs<T> p0 = a;
    p0.N = 0; p0.mw_post = nullptr; p0.T_post = nullptr; p0.Lw_post = nullptr; p0.logpdf = nullptr; p0.info = prior_info;
    if ((rc = dispatch_posterior<T>(h, p0))) return rc;
  }
  if ((rc = dispatch_posterior<T>(h, a))) return rc;
  hipLaunchKernelGGL(dense_finish_kernel, dim3(1), dim3(64), 0, h->stream, lp_tmp, info_d, (const double*)logdet, (const int32_t*)noise_info,
                     (const int32_t*)prior_info);
  HIP_TRY(h, hipGetLastError());
  if (memspace == BLR_MEM_HOST) {
    if (mw_post) HIP_TRY(h, hipMemcpyAsync(mw_post, mwp_d, (size_t)D * sizeof(T), hipMemcpyDeviceToHost, h->stream));
    if (T_post) HIP_TRY(h, hipMemcpyAsync(T_post, Tp_d, mat_extent(D, D, ldt) * sizeof(T), hipMemcpyDeviceToHost, h->stream));
    if (Lw_post) HIP_TRY(h, hipMemcpyAsync(Lw_post, Lp_d, mat_extent(D, D, ldlp) * sizeof(T), hipMemcpyDeviceToHost, h->stream));
    if (logpdf) HIP_TRY(h, hipMemcpyAsync(logpdf, lp_d, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(info, info_d, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
  }
  HIP_TRY(h, hipStreamSynchronize(h->stream));  // temporaries are freed on return
  return 0;
}

// Y = X' Lw^-T (N rows x D columns, column-major with leading dimension ldy, rows [DP, DP + NP) of Ybar) for a factored or
// dense prior: the tall-matrix panel sweep of the marginal path, keeping Y instead of folding it into row sums.
template <typename T>
int tall_solve_rows(blr_handle* h, int layout, int D, int N, const T* X, int64_t ldx, int prior_kind, const T* Lw, int64_t ldl,
                    T* Ybar, int64_t ldy, int DP, int NP, int32_t* info_dev) {
  using TC = TrsmCfg<T>;
  using LC = LargeCfg<T>;
  const int NC = DP / kPB;
  int rc;
  HIP_TRY(h, hipMemsetAsync(info_dev, 0, sizeof(int32_t), h->stream));
  {
    MeanFillArgs<T> m{};
    m.X = X; m.ldx = ldx; m.layout = layout; m.mw = nullptr; m.mean = nullptr; m.Ybar = Ybar; m.ldy = ldy; m.row0 = DP;
    m.D = D; m.DP = DP; m.N = N;
    hipLaunchKernelGGL(mean_fill_kernel<T>, dim3((unsigned)(NP / 64)), dim3(kThreads), 0, h->stream, m);
  }
  if (prior_kind == BLR_PRIOR_UPPER_FACTOR) {
    // a PDMat whose factor has a non-positive diagonal entry is not a Cholesky factor: report it (LAPACK-style index) instead
    // of returning Inf / NaN with info = 0; the panel kernels below return early on a non-zero status
    hipLaunchKernelGGL(prior_diag_kernel<T>, dim3(1), dim3(kThreads), 0, h->stream, Lw, ldl, (int)PRIOR_UPPER_FACTOR, D, (double*)nullptr,
                       info_dev);
    dim3 grid((DP + 31) / 32, (DP + 31) / 32);
    hipLaunchKernelGGL(factor_transpose_fill_kernel<T>, grid, dim3(kThreads), 0, h->stream, Lw, ldl, D, DP, Ybar, ldy);
  } else {
    hipLaunchKernelGGL(prior_copy_kernel<T>, dim3(1024), dim3(kThreads), 0, h->stream, Lw, ldl, D, DP, Ybar, ldy);
    if ((rc = chol_large<T>(h, Ybar, ldy, DP, DP, info_dev))) return rc;
  }
  if ((rc = set_lds<T>(h, reinterpret_cast<const void*>(trsm_block_kernel<T>), TC::LDS_BYTES))) return rc;
  if ((rc = set_lds<T>(h, reinterpret_cast<const void*>(gram_tile_kernel<T>), LC::LDS_BYTES))) return rc;
  const int nyb = NP / kPB;
  for (int p = 0; p < NC; ++p) {
    const int nblk = (NP + TC::RB - 1) / TC::RB;
    hipLaunchKernelGGL(trsm_block_kernel<T>, dim3(nblk), dim3(kThreads), TC::LDS_BYTES, h->stream, Ybar, ldy, p, DP, DP + NP,
                       (const int32_t*)info_dev, RowSqArgs<T>{});
    const int m = NC - 1 - p;
    if (m > 0) {
      GramTileArgs<T> g{};
      g.X = Ybar + (int64_t)p * kPB * ldy; g.ldx = ldy; g.layout = LAYOUT_COLVECS; g.use_dma = 1;
      g.s = nullptr; g.noise_kind = NOISE_ISOTROPIC; g.r = nullptr;
      g.D = DP + NP; g.n_begin = 0; g.n_end = kPB; g.nsplit = 1;
      g.tile_i0 = NC; g.tile_j0 = p + 1; g.tri = 3; g.ntile_rows = nyb; g.ntiles = nyb * m; g.nblocks = NC + nyb;
      g.C = Ybar; g.ldc = ldy; g.mode_out = 1;
      hipLaunchKernelGGL(gram_tile_kernel<T>, dim3(g.ntiles), dim3(kThreads), LC::LDS_BYTES, h->stream, g);
    }
  }
  HIP_TRY(h, hipGetLastError());
  return 0;
}

template <typename T>
int mean_and_cov(blr_handle* h, int memspace, int layout, int64_t D64, int64_t N64, const T* X, int64_t ldx, int noise_kind, const T* s,
                 int64_t lds, int prior_kind, const T* mw, const T* Lw, int64_t ldl, T* mean, T* C, int64_t ldc, int32_t* info) {
  if (!h) return -1;
  h->err.clear();
  if (memspace != BLR_MEM_HOST && memspace != BLR_MEM_DEVICE) return bad_arg(h, 2, "memspace");
  if (layout != BLR_LAYOUT_COLVECS && layout != BLR_LAYOUT_ROWVECS) return bad_arg(h, 3, "unknown layout (reference :26-31)");
  if (D64 < 1 || D64 > kMaxLargeD) return bad_arg(h, 4, "D out of range for this build (1..8192)");
  if (N64 < 1 || N64 > kMaxDenseN) return bad_arg(h, 5, "N out of range for an N x N covariance (1..16384)");
  if (!X) return bad_arg(h, 6, "X is NULL");
  if (layout == BLR_LAYOUT_COLVECS ? ldx < D64 : ldx < N64) return bad_arg(h, 7, "ldx too small");
  if (noise_kind != BLR_NOISE_ISOTROPIC && noise_kind != BLR_NOISE_DIAGONAL && noise_kind != BLR_NOISE_DENSE) return bad_arg(h, 8, "noise_kind");
  if (!s) return bad_arg(h, 9, "s is NULL");
  if (noise_kind == BLR_NOISE_DENSE && lds < N64) return bad_arg(h, 10, "lds < N");
  if (prior_kind != BLR_PRIOR_DENSE && prior_kind != BLR_PRIOR_UPPER_FACTOR && prior_kind != BLR_PRIOR_DIAGONAL)
    return bad_arg(h, 11, "prior_kind");
  if (mean && !mw) return bad_arg(h, 12, "mw is NULL");
  if (!Lw) return bad_arg(h, 13, "Lw is NULL");
  if (prior_kind != BLR_PRIOR_DIAGONAL && ldl < D64) return bad_arg(h, 14, "ldl < D");
  if (!C) return bad_arg(h, 16, "C is NULL");
  if (ldc < N64) return bad_arg(h, 17, "ldc < N");
  if (!info) return bad_arg(h, 18, "info is NULL");
  HIP_TRY(h, hipSetDevice(h->device));
  const int D = (int)D64, N = (int)N64;
  const int DP = (D + kPB - 1) / kPB * kPB, NP = (N + kPB - 1) / kPB * kPB;
  const int64_t ldy = (int64_t)DP + NP;
  Staging guard(h);
  int rc;
  const T *X_d = X, *s_d = s, *mw_d = mw, *Lw_d = Lw;
  T *mean_d = mean, *C_d = C;
  int32_t* info_d = info;
  if (memspace == BLR_MEM_HOST) {
    const size_t x_one = layout == BLR_LAYOUT_COLVECS ? mat_extent(D, N, ldx) : mat_extent(N, D, ldx);
    const size_t lw_one = prior_kind == BLR_PRIOR_DIAGONAL ? (size_t)D : mat_extent(D, D, ldl);
    const size_t s_one = noise_kind == BLR_NOISE_DENSE ? mat_extent(N, N, lds) : (noise_kind == BLR_NOISE_DIAGONAL ? (size_t)N : 1);
    if ((rc = stage_in(h, X, x_one, &X_d))) return rc;
    if ((rc = stage_in(h, s, s_one, &s_d))) return rc;
    if ((rc = stage_in(h, mw, mw ? (size_t)D : 0, &mw_d))) return rc;
    if ((rc = stage_in(h, Lw, lw_one, &Lw_d))) return rc;
    if ((rc = stage_out_alloc(h, mean, (size_t)N, &mean_d))) return rc;
    if ((rc = stage_out_alloc(h, C, mat_extent(N, N, ldc), &C_d))) return rc;
    if ((rc = stage_out_alloc(h, info, (size_t)1, &info_d))) return rc;
  }
  T *Ybar = nullptr, *Gpart = nullptr;
  double* scratch = nullptr;
  const int nb = NP / kPB, ntiles = nb * (nb + 1) / 2;
  if ((rc = dev_alloc_tmp(h, (size_t)ldy * DP, &Ybar))) return rc;
  if ((rc = dev_alloc_tmp(h, (size_t)ntiles * kPB * kPB, &Gpart))) return rc;
  if ((rc = dev_alloc_tmp(h, 1, &scratch))) return rc;
  // ---- Y = alpha' = X' Uw^-1   (reference :36 alpha = Uw' \ X)
  if (prior_kind == BLR_PRIOR_DIAGONAL) {
    hipLaunchKernelGGL(prior_diag_kernel<T>, dim3(1), dim3(kThreads), 0, h->stream, Lw_d, (int64_t)1, prior_kind, D, scratch, info_d);
    hipLaunchKernelGGL(diag_prior_rows_kernel<T>, dim3(2048), dim3(kThreads), 0, h->stream, X_d, ldx, layout, Lw_d, D, N, NP, DP, Ybar + DP, ldy);
  } else {
    if ((rc = tall_solve_rows<T>(h, layout, D, N, X_d, ldx, prior_kind, Lw_d, ldl, Ybar, ldy, DP, NP, info_d))) return rc;
  }
  // ---- Y Y' by the Gram kernel with the roles swapped: NP "rows", D "observations" (columns of Y are contiguous)
  {
    using LC = LargeCfg<T>;
    if ((rc = set_lds<T>(h, reinterpret_cast<const void*>(gram_tile_kernel<T>), LC::LDS_BYTES))) return rc;
    GramTileArgs<T> g{};
    g.X = Ybar + DP; g.ldx = ldy; g.layout = LAYOUT_COLVECS; g.use_dma = 1;
    g.s = nullptr; g.noise_kind = NOISE_ISOTROPIC; g.r = nullptr;
    g.D = NP; g.n_begin = 0; g.n_end = DP; g.nsplit = 1;
    g.tile_i0 = 0; g.tile_j0 = 0; g.tri = 1; g.ntiles = ntiles; g.nblocks = nb;
    g.Gpart = Gpart; g.bpart = nullptr; g.mode_out = 0; g.xcd_swizzle = 0;
    hipLaunchKernelGGL(gram_tile_kernel<T>, dim3(ntiles), dim3(kThreads), LC::LDS_BYTES, h->stream, g);
    hipLaunchKernelGGL(cov_assemble_kernel<T>, dim3(ntiles, 16), dim3(kThreads), 0, h->stream, (const T*)Gpart, ntiles, N, noise_kind, s_d,
                       lds, C_d, ldc);
  }
  // ---- mean = X' mw   (reference :33)
  if (mean)
    hipLaunchKernelGGL(mean_small_kernel<T>, dim3((unsigned)((N + 255) / 256)), dim3(kThreads), 0, h->stream, X_d, ldx, layout, mw_d, D, N, mean_d);
  HIP_TRY(h, hipGetLastError());
  if (memspace == BLR_MEM_HOST) {
    if (mean) HIP_TRY(h, hipMemcpyAsync(mean, mean_d, (size_t)N * sizeof(T), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(C, C_d, mat_extent(N, N, ldc) * sizeof(T), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(info, info_d, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
  }
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return 0;
}

// rand with a dense noise covariance: Y = X'(mw + Uw \ Z1) + Us' Z2   (reference :49-53).  Returns info (> 0: Sigma_y or Lw
// not positive definite).
template <typename T>
int rand_dense_noise(blr_handle* h, int memspace, int layout, int64_t D, int64_t N, int64_t S, const T* X, int64_t ldx, const T* Sy,
                     int64_t ldsy, int prior_kind, const T* mw, const T* Lw, int64_t ldl, const T* Z1, int64_t ldz1, const T* Z2,
                     int64_t ldz2, T* Y, int64_t ldy) {
  if (!h) return -1;
  h->err.clear();
  if (N < 1 || N > kMaxDenseN) return bad_arg(h, 5, "N out of range for a dense noise covariance (1..16384)");
  if (!Sy) return bad_arg(h, 9, "Sy is NULL");
  if (ldsy < N) return bad_arg(h, 10, "ldsy < N");
  if (S < 0) return bad_arg(h, 6, "S < 0");
  if (S == 0) return 0;
  // X' W with the noise term switched off (sigma^2 = 0), through the ordinary entry point (validates everything else)
  const T zero = T(0);
  Staging guard(h);
  int rc;
  const T* zero_d = &zero;
  if (memspace == BLR_MEM_DEVICE) {
    T* z = nullptr;
    HIP_TRY(h, hipSetDevice(h->device));
    if ((rc = dev_alloc_tmp(h, 1, &z))) return rc;
    HIP_TRY(h, hipMemsetAsync(z, 0, sizeof(T), h->stream));
    zero_d = z;
  }
  rc = rand_impl<T>(h, memspace, layout, D, N, S, X, ldx, BLR_NOISE_ISOTROPIC, zero_d, prior_kind, mw, Lw, ldl, Z1, ldz1, Z2, ldz2, Y, ldy);
  if (rc) return rc;
  // + L Z2 with L L' = Sigma_y
  const int NP = (int)((N + kPB - 1) / kPB * kPB);
  const T *Sy_d = Sy, *Z2_d = Z2;
  T* Y_d = Y;
  if (memspace == BLR_MEM_HOST) {
    if ((rc = stage_in(h, Sy, mat_extent(N, N, ldsy), &Sy_d))) return rc;
    if ((rc = stage_in(h, Z2, mat_extent(N, S, ldz2), &Z2_d))) return rc;
    if ((rc = stage_out_alloc(h, Y, mat_extent(N, S, ldy), &Y_d))) return rc;  // copies the X'W part back in
  }
  T* M = nullptr;
  int32_t* ninfo = nullptr;
  if ((rc = dev_alloc_tmp(h, (size_t)NP * NP, &M))) return rc;
  if ((rc = dev_alloc_tmp(h, 1, &ninfo))) return rc;
  if ((rc = chol_noise<T>(h, Sy_d, ldsy, (int)N, NP, 0, M, (int64_t)NP, ninfo))) return rc;
  dim3 grid((unsigned)((N + 63) / 64), (unsigned)((S + 15) / 16));
  hipLaunchKernelGGL(lower_mult_add_kernel<T>, grid, dim3(kThreads), 0, h->stream, (const T*)M, (int64_t)NP, (int)N, Z2_d, ldz2, Y_d, ldy, S);
  HIP_TRY(h, hipGetLastError());
  int32_t hinfo = 0;
  HIP_TRY(h, hipMemcpyAsync(&hinfo, ninfo, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
  if (memspace == BLR_MEM_HOST) HIP_TRY(h, hipMemcpyAsync(Y, Y_d, mat_extent(N, S, ldy) * sizeof(T), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return hinfo;
}

// ---- rank-k update of a resident state (blr_update.hpp) ---------------------------------------------------------------
template <typename T>
int update_factor(blr_handle* h, int memspace, int layout, int64_t B, int64_t D, int64_t k, const T* X, int64_t ldx,
                  int64_t strideX, const T* y, int64_t stridey, int noise_kind, const T* s, int64_t strides, T* mw,
                  int64_t stridemw, T* Tf, int64_t ldt, int64_t strideT, double* logpdf, int32_t* info) {
  if (!h) return -1;
  h->err.clear();
  // Route (measured, tools/update_bench.py): a sweep costs ~40 us per observation at D = 128 (a serial chain
  // of D rotations), the in-place re-factorisation ~90 us per CALL whatever k is (45 us at D = 64, where batches run on the
  // one-wave-per-regressor kernel at 27 M updates/s) -- the sweep wins for a single new observation, except in large
  // batches at D <= 64.  A blocked (Householder) sweep that eliminates up to 16 rows in one chain of D reflections was built
  // and measured in round 3: 92 / 115 / 144 / 205 us at k = 1 / 3 / 8 / 16 against 90 us -- the k + 1-term dot products and
  // three reciprocal chains per step cost more than the rotations they replace -- and was not kept.
  // BLR_MI355X_SWEEP=always / never overrides (tests exercise both routes on the same inputs).
  const int mode = h->opt.sweep;
  const bool can_sweep = D >= 1 && D <= kSweepMaxD && k >= 0 && k <= kSweepMaxK;
  bool sweep = can_sweep && k <= 1 && (D > 64 || B < 256);
  if (mode == 1) sweep = can_sweep;
  if (mode == 2) sweep = false;
  if (!sweep) {
    // k-independent cost: the SAME state re-factored in place, the old factor entering as D pseudo-observations
    // (reads of T and mw complete before the first write in both the fused and the large-D path)
    if (!mw) return bad_arg(h, 15, "mw is NULL");
    if (!Tf) return bad_arg(h, 17, "T is NULL");
    return posterior_batched<T>(h, memspace, layout, B, D, k, X, ldx, strideX, y, stridey, noise_kind, s, strides,
                                BLR_PRIOR_UPPER_FACTOR, mw, stridemw, Tf, ldt, strideT, mw, stridemw, Tf, ldt, strideT,
                                (T*)nullptr, 0, 0, logpdf, info);
  }
  if (memspace != BLR_MEM_HOST && memspace != BLR_MEM_DEVICE) return bad_arg(h, 2, "memspace");
  if (layout != BLR_LAYOUT_COLVECS && layout != BLR_LAYOUT_ROWVECS) return bad_arg(h, 3, "unknown layout (reference :26-31)");
  if (B < 0 || B > (1 << 30)) return bad_arg(h, 4, "B out of range (0..2^30)");
  if (B == 0) return 0;
  if (k > 0 && !X) return bad_arg(h, 7, "X is NULL");
  if (layout == BLR_LAYOUT_COLVECS ? ldx < D : ldx < std::max<int64_t>(k, 1)) return bad_arg(h, 8, "ldx too small");
  if (strideX < 0) return bad_arg(h, 9, "strideX < 0");
  if (k > 0 && !y) return bad_arg(h, 10, "y is NULL (reference :74 length check)");
  if (stridey < 0) return bad_arg(h, 11, "stridey < 0");
  if (noise_kind != BLR_NOISE_ISOTROPIC && noise_kind != BLR_NOISE_DIAGONAL) return bad_arg(h, 12, "noise_kind");
  if (!s) return bad_arg(h, 13, "s is NULL");
  if (strides < 0) return bad_arg(h, 14, "strides < 0");
  if (!mw) return bad_arg(h, 15, "mw is NULL");
  if (B > 1 && stridemw < D) return bad_arg(h, 16, "stridemw < D");
  if (!Tf) return bad_arg(h, 17, "T is NULL");
  if (ldt < D) return bad_arg(h, 18, "ldt < D");
  if (B > 1 && strideT < (int64_t)mat_extent(D, D, ldt)) return bad_arg(h, 19, "strideT too small");
  if (!info) return bad_arg(h, 21, "info is NULL");
  HIP_TRY(h, hipSetDevice(h->device));
  SweepArgs<T> a{};
  a.ldx = ldx; a.strideX = strideX; a.layout = layout; a.stridey = stridey; a.strides = strides; a.noise_kind = noise_kind;
  a.stridemw = stridemw; a.ldt = ldt; a.strideT = strideT; a.D = (int)D; a.k = (int)k;
  const int lds = sweep_lds_bytes<T>((int)D);
  { const int rc_lds = set_lds_once(h, reinterpret_cast<const void*>(rank1_sweep_kernel<T>), (size_t)(lds)); if (rc_lds) return rc_lds; }
  if (memspace == BLR_MEM_DEVICE) {
    a.X = X; a.y = y; a.s = s; a.mw = mw; a.Tf = Tf; a.logpdf = logpdf; a.info = info;
    hipLaunchKernelGGL(rank1_sweep_kernel<T>, dim3((unsigned)B), dim3(kThreads), lds, h->stream, a);
    HIP_TRY(h, hipGetLastError());
    if (!h->async) HIP_TRY(h, hipStreamSynchronize(h->stream));
    return 0;
  }
  Staging guard(h);
  const size_t x_one = layout == BLR_LAYOUT_COLVECS ? mat_extent(D, k, ldx) : mat_extent(k, D, ldx);
  const size_t s_one = noise_kind == BLR_NOISE_DIAGONAL ? (size_t)k : 1;
  const size_t n_mw = extent(B, stridemw, (size_t)D), n_T = extent(B, strideT, mat_extent(D, D, ldt));
  int rc;
  const T *dmw = nullptr, *dT = nullptr;
  if ((rc = stage_in(h, X, extent(B, strideX, x_one), &a.X))) return rc;
  if ((rc = stage_in(h, y, extent(B, stridey, (size_t)k), &a.y))) return rc;
  if ((rc = stage_in(h, s, extent(B, strides, s_one), &a.s))) return rc;
  if ((rc = stage_in(h, (const T*)mw, n_mw, &dmw))) return rc;
  if ((rc = stage_in(h, (const T*)Tf, n_T, &dT))) return rc;
  a.mw = const_cast<T*>(dmw); a.Tf = const_cast<T*>(dT);
  if ((rc = stage_out_alloc(h, logpdf, (size_t)B, &a.logpdf))) return rc;
  if ((rc = stage_out_alloc(h, info, (size_t)B, &a.info))) return rc;
  if (k == 0) { if (!a.X) a.X = a.mw; if (!a.y) a.y = a.mw; }
  hipLaunchKernelGGL(rank1_sweep_kernel<T>, dim3((unsigned)B), dim3(kThreads), lds, h->stream, a);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipMemcpyAsync(mw, a.mw, n_mw * sizeof(T), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipMemcpyAsync(Tf, a.Tf, n_T * sizeof(T), hipMemcpyDeviceToHost, h->stream));
  if (logpdf) HIP_TRY(h, hipMemcpyAsync(logpdf, a.logpdf, (size_t)B * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipMemcpyAsync(info, a.info, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return 0;
}

}  // namespace

// =======================================================================================================
extern "C" {

int blr_abi_version(void) { return BLR_ABI_VERSION; }

int blr_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int blr_create(int device, blr_handle** out) {
  if (!out) return -2;
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) return -(1000 + (int)(e == hipSuccess ? hipErrorNoDevice : e));
  if (device < 0 || device >= n) return -1;
  blr_handle* h = new (std::nothrow) blr_handle();
  if (!h) return -(1000 + (int)hipErrorOutOfMemory);
  h->device = device;
  if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipEventCreate(&h->ev0)) != hipSuccess || (e = hipEventCreate(&h->ev1)) != hipSuccess) {
    delete h;
    return -(1000 + (int)e);
  }
  h->stream = h->own_stream;
  h->opt.from_environment();  // the only getenv calls of the library
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) h->cus = cus;
  *out = h;
  return 0;
}

int blr_destroy(blr_handle* h) {
  if (!h) return 0;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  if (h->comm && rccl().ok) (void)rccl().CommDestroy(h->comm);
  if (h->ws) (void)hipFree(h->ws);
  if (h->feat) (void)hipFree(h->feat);
  if (h->aux) (void)hipFree(h->aux);
  if (h->i8side) (void)hipFree(h->i8side);
  if (h->stats_dev) (void)hipFree(h->stats_dev);
  if (h->xchg) (void)hipFree(h->xchg);
  if (h->ticket) (void)hipFree(h->ticket);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
  return 0;
}

const char* blr_last_error(blr_handle* h) { return h ? h->err.c_str() : "null handle"; }

int blr_release_workspace(blr_handle* h) {
  if (!h) return -1;
  h->err.clear();
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (h->ws) { HIP_TRY(h, hipFree(h->ws)); h->ws = nullptr; h->ws_bytes = 0; }
  if (h->feat) { HIP_TRY(h, hipFree(h->feat)); h->feat = nullptr; h->feat_bytes = 0; }
  if (h->aux) { HIP_TRY(h, hipFree(h->aux)); h->aux = nullptr; h->aux_bytes = 0; }
  if (h->i8side) { HIP_TRY(h, hipFree(h->i8side)); h->i8side = nullptr; h->i8side_bytes = 0; }
  return 0;
}

int blr_set_option(blr_handle* h, const char* key, const char* value) {
  if (!h) return -1;
  h->err.clear();
  const int orc = h->opt.set(key, value);
  if (orc == -2) return bad_arg(h, 2, "unknown option key");
  if (orc != 0) return bad_arg(h, 3, "malformed option value");
  h->gram_plans.clear();  // (cached launch plans were made under the old switches)
  return 0;
}

// Work of one handle is ordered by ITS stream: the arrival counters of panel_chain_kernel, the tagged exchange buffer and the
// grow-only scratch assume that a launch's predecessor on the handle has finished with them.  Switching streams therefore
// drains the old one first and re-arms both banks of arrival counters (a launch clears the bank of its successor, which on a
// second stream could already be counting in it).
static int switch_stream(blr_handle* h, hipStream_t next) {
  if (next == h->stream) return 0;
  h->err.clear();
  HIP_TRY(h, hipSetDevice(h->device));
  // The old stream must still be alive here (header).  If the caller has destroyed it all the same, the drain fails -- and must not
  // leave the handle bound to a dead stream for good: clear the error, drain the device instead, and switch anyway.
  if (hipStreamSynchronize(h->stream) != hipSuccess) {
    (void)hipGetLastError();
    HIP_TRY(h, hipDeviceSynchronize());
  }
  h->stream = next;
  if (h->ticket) HIP_TRY(h, hipMemsetAsync(h->ticket + 16, 0, 2 * kPanelArriveWords * sizeof(unsigned), h->stream));
  h->panel_launches = 0;
  return 0;
}
int blr_set_stream(blr_handle* h, void* hip_stream) {
  if (!h) return -1;
  return switch_stream(h, static_cast<hipStream_t>(hip_stream));
}
int blr_reset_stream(blr_handle* h) {
  if (!h) return -1;
  return switch_stream(h, h->own_stream);
}
const char* blr_last_route(blr_handle* h) {
  if (!h) return "null handle";
  // The int8 route decides ON THE DEVICE what it keeps: a probe slice that hands back more than a quarter of its regressors (heavy
  // tails) sends the rest of the batch to the fp64 kernel, and then that kernel is what a profile of the call shows.  Look at the
  // call's hand-back count (this drains the handle's stream) and name the kernel that did most of the work.
  if (h->route_i8_B > 0 && h->stats_dev != nullptr) {
    unsigned long long v[3] = {0, 0, 0};
    if (hipSetDevice(h->device) == hipSuccess &&
        hipMemcpyAsync(v, h->stats_dev, sizeof(v), hipMemcpyDeviceToHost, h->stream) == hipSuccess && hipStreamSynchronize(h->stream) == hipSuccess) {
      const unsigned long long handed = v[0] - v[2];
      if (2 * handed > (unsigned long long)h->route_i8_B) {
        h->route_buf = "fused_small_kernel<double, 8, 4> (int8 route handed back " + std::to_string(handed) + " of " + std::to_string(h->route_i8_B) + ")";
        return h->route_buf.c_str();
      }
    } else {
      (void)hipGetLastError();
    }
  }
  return h->route;
}

int blr_get_stat(blr_handle* h, const char* key, int64_t* value) {
  if (!h) return -1;
  h->err.clear();
  if (!key) return bad_arg(h, 2, "key is NULL");
  if (!value) return bad_arg(h, 3, "value is NULL");
  if (!strcmp(key, "i8_regressors")) { *value = (int64_t)h->i8_attempted; return 0; }
  if (!strcmp(key, "i8_handed_back")) {
    unsigned long long v = 0;
    if (h->stats_dev) {
      HIP_TRY(h, hipSetDevice(h->device));
      HIP_TRY(h, hipMemcpyAsync(&v, h->stats_dev, sizeof(v), hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    *value = (int64_t)v;
    return 0;
  }
  if (!strcmp(key, "planes_redone")) {  // large-D fp32 updates whose sampled row scales did not hold: exact row maxima + planes made again
    unsigned long long v = 0;
    if (h->stats_dev) {
      HIP_TRY(h, hipSetDevice(h->device));
      HIP_TRY(h, hipMemcpyAsync(&v, h->stats_dev + 1, sizeof(v), hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    *value = (int64_t)v;
    return 0;
  }
  if (!strcmp(key, "workspace_bytes")) { *value = (int64_t)(h->ws_bytes + h->feat_bytes + h->aux_bytes + h->i8side_bytes + h->xchg_bytes); return 0; }
  return bad_arg(h, 2, "unknown statistic");
}
int blr_reset_stats(blr_handle* h) {
  if (!h) return -1;
  h->err.clear();
  if (h->stats_dev) {  // (device counter first: a failure must not leave the two counters apart)
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMemsetAsync(h->stats_dev, 0, 3 * sizeof(unsigned long long), h->stream));  // (total, planes made twice, the last call's base)
  }
  h->i8_attempted = 0;
  return 0;
}
int blr_set_async(blr_handle* h, int async) {
  if (!h) return -1;
  h->async = async != 0;
  return 0;
}
int blr_synchronize(blr_handle* h) {
  if (!h) return -1;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return 0;
}

int blr_device_alloc(blr_handle* h, size_t bytes, void** dptr) {
  if (!h) return -1;
  if (!dptr) return -3;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipMalloc(dptr, bytes ? bytes : 1));
  return 0;
}
int blr_device_free(blr_handle* h, void* dptr) {
  if (!h) return -1;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  HIP_TRY(h, hipFree(dptr));
  return 0;
}
int blr_memcpy_h2d(blr_handle* h, void* dst, const void* src, size_t bytes) {
  if (!h) return -1;
  HIP_TRY(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return 0;
}
int blr_memcpy_d2h(blr_handle* h, void* dst, const void* src, size_t bytes) {
  if (!h) return -1;
  HIP_TRY(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return 0;
}

int blr_timer_start(blr_handle* h) {
  if (!h) return -1;
  HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
  return 0;
}
int blr_timer_stop(blr_handle* h, float* elapsed_ms) {
  if (!h) return -1;
  if (!elapsed_ms) return -2;
  HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
  HIP_TRY(h, hipEventSynchronize(h->ev1));
  HIP_TRY(h, hipEventElapsedTime(elapsed_ms, h->ev0, h->ev1));
  return 0;
}

#define BLR_DEFINE(SUF, T)                                                                                          \
  int blr_posterior_batched_##SUF(blr_handle* h, int memspace, int layout, int64_t B, int64_t D, int64_t N,        \
                                  const T* X, int64_t ldx, int64_t strideX, const T* y, int64_t stridey,            \
                                  int noise_kind, const T* s, int64_t strides, int prior_kind, const T* mw,         \
                                  int64_t stridemw, const T* Lw, int64_t ldl, int64_t strideLw, T* mw_post,         \
                                  int64_t stride_mwpost, T* T_post, int64_t ldt, int64_t strideT, T* Lw_post,      \
                                  int64_t ldlp, int64_t strideLp, double* logpdf, int32_t* info) {                  \
    return posterior_batched<T>(h, memspace, layout, B, D, N, X, ldx, strideX, y, stridey, noise_kind, s, strides,  \
                                prior_kind, mw, stridemw, Lw, ldl, strideLw, mw_post, stride_mwpost, T_post, ldt,    \
                                strideT, Lw_post, ldlp, strideLp, logpdf, info);                                    \
  }                                                                                                                 \
  int blr_update_factor_##SUF(blr_handle* h, int memspace, int layout, int64_t B, int64_t D, int64_t k, const T* X, \
                              int64_t ldx, int64_t strideX, const T* y, int64_t stridey, int noise_kind, const T* s, \
                              int64_t strides, T* mw, int64_t stridemw, T* Tf, int64_t ldt, int64_t strideT,         \
                              double* logpdf, int32_t* info) {                                                       \
    return update_factor<T>(h, memspace, layout, B, D, k, X, ldx, strideX, y, stridey, noise_kind, s, strides, mw,   \
                            stridemw, Tf, ldt, strideT, logpdf, info);                                               \
  }                                                                                                                 \
  int blr_posterior_##SUF(blr_handle* h, int layout, int64_t D, int64_t N, const T* X, int64_t ldx, const T* y,     \
                          int noise_kind, const T* s, int prior_kind, const T* mw, const T* Lw, int64_t ldl,        \
                          T* mw_post, T* T_post, int64_t ldt, T* Lw_post, int64_t ldlp, double* logpdf) {           \
    return posterior_single<T>(h, layout, D, N, X, ldx, y, noise_kind, s, prior_kind, mw, Lw, ldl, mw_post, T_post, \
                               ldt, Lw_post, ldlp, logpdf);                                                         \
  }                                                                                                                 \
  int blr_marginals_batched_##SUF(blr_handle* h, int memspace, int layout, int64_t B, int64_t D, int64_t N,        \
                                  const T* X, int64_t ldx, int64_t strideX, int noise_kind, const T* s,             \
                                  int64_t strides, int prior_kind, const T* mw, int64_t stridemw, const T* Lw,      \
                                  int64_t ldl, int64_t strideLw, T* mean, int64_t stridemean, T* var,               \
                                  int64_t stridevar, int32_t* info) {                                               \
    return marginals_batched<T>(h, memspace, layout, B, D, N, X, ldx, strideX, noise_kind, s, strides, prior_kind,  \
                                mw, stridemw, Lw, ldl, strideLw, mean, stridemean, var, stridevar, info);           \
  }                                                                                                                 \
  int blr_rand_##SUF(blr_handle* h, int memspace, int layout, int64_t D, int64_t N, int64_t S, const T* X,          \
                     int64_t ldx, int noise_kind, const T* s, int prior_kind, const T* mw, const T* Lw,             \
                     int64_t ldl, const T* Z1, int64_t ldz1, const T* Z2, int64_t ldz2, T* Y, int64_t ldy) {        \
    return rand_impl<T>(h, memspace, layout, D, N, S, X, ldx, noise_kind, s, prior_kind, mw, Lw, ldl, Z1, ldz1, Z2, \
                        ldz2, Y, ldy);                                                                              \
  }                                                                                                                 \
  int blr_sample_weights_##SUF(blr_handle* h, int memspace, int64_t D, int64_t S, int prior_kind, const T* mw,      \
                               const T* Lw, int64_t ldl, const T* Z, int64_t ldz, T* W, int64_t ldw) {              \
    return sample_weights<T>(h, memspace, D, S, prior_kind, mw, Lw, ldl, Z, ldz, W, ldw);                           \
  }                                                                                                                 \
  int blr_rff_features_##SUF(blr_handle* h, int memspace, int64_t Din, int64_t D, int64_t N, const T* Xin,          \
                             int64_t ldxin, const T* Omega, int64_t ldo, const T* phase, T scale, T* Phi,           \
                             int64_t ldphi) {                                                                       \
    return rff_features<T>(h, memspace, Din, D, N, Xin, ldxin, Omega, ldo, phase, scale, Phi, ldphi);               \
  }                                                                                                                 \
  int blr_logpdf_grad_batched_##SUF(blr_handle* h, int memspace, int layout, int64_t B, int64_t D, int64_t N,      \
                                    const T* X, int64_t ldx, int64_t strideX, const T* y, int64_t stridey,          \
                                    int noise_kind, const T* s, int64_t strides, int prior_kind, const T* mw,       \
                                    int64_t stridemw, const T* Lw, int64_t ldl, int64_t strideLw, double* logpdf,   \
                                    T* dX, int64_t lddx, int64_t stridedX, T* dy, int64_t stridedy, T* ds,          \
                                    int64_t strideds, T* dmw, int64_t stridedmw, T* mw_post, int64_t stride_mwpost, \
                                    T* Ainv, int64_t ldai, int64_t strideAi, int32_t* info) {                       \
    return logpdf_grad_batched<T>(h, memspace, layout, B, D, N, X, ldx, strideX, y, stridey, noise_kind, s, strides, \
                                  prior_kind, mw, stridemw, Lw, ldl, strideLw, logpdf, dX, lddx, stridedX, dy,       \
                                  stridedy, ds, strideds, dmw, stridedmw, mw_post, stride_mwpost, Ainv, ldai,        \
                                  strideAi, info);                                                                  \
  }                                                                                                                 \
  int blr_logpdf_multi_##SUF(blr_handle* h, int memspace, int layout, int64_t D, int64_t N, int64_t S, const T* X,  \
                             int64_t ldx, const T* Y, int64_t ldY, int noise_kind, const T* s, int prior_kind,      \
                             const T* mw, const T* Lw, int64_t ldl, double* logpdf, T* mw_post, int64_t ldmp,       \
                             int32_t* info) {                                                                       \
    return logpdf_multi<T>(h, memspace, layout, D, N, S, X, ldx, Y, ldY, noise_kind, s, prior_kind, mw, Lw, ldl,     \
                           logpdf, mw_post, ldmp, info);                                                            \
  }                                                                                                                 \
  int blr_gram_stats_##SUF(blr_handle* h, int layout, int64_t D, int64_t N, const T* X, int64_t ldx, const T* y,    \
                           int noise_kind, const T* s, const T* mw, T* stats, int64_t lds, double* scal) {          \
    return gram_stats<T>(h, layout, D, N, X, ldx, y, noise_kind, s, mw, stats, lds, scal);                          \
  }                                                                                                                 \
  int blr_posterior_from_stats_##SUF(blr_handle* h, int64_t D, int64_t N_total, T* stats, int64_t lds,              \
                                     const double* scal, int prior_kind, const T* mw, const T* Lw, int64_t ldl,     \
                                     T* mw_post, T* T_post, int64_t ldt, T* Lw_post, int64_t ldlp, double* logpdf,  \
                                     int32_t* info) {                                                               \
    return posterior_from_stats<T>(h, D, N_total, stats, lds, scal, prior_kind, mw, Lw, ldl, mw_post, T_post, ldt,   \
                                   Lw_post, ldlp, logpdf, info);                                                    \
  }                                                                                                                 \
  int blr_posterior_rff_##SUF(blr_handle* h, int memspace, int64_t Din, int64_t D, int64_t N, const T* Xin,         \
                              int64_t ldxin, const T* Omega, int64_t ldo, const T* phase, T scale, const T* y,      \
                              int noise_kind, const T* s, int prior_kind, const T* mw, const T* Lw, int64_t ldl,    \
                              T* mw_post, T* T_post, int64_t ldt, T* Lw_post, int64_t ldlp, double* logpdf,         \
                              int32_t* info) {                                                                      \
    return posterior_rff<T>(h, memspace, Din, D, N, Xin, ldxin, Omega, ldo, phase, scale, y, noise_kind, s,         \
                            prior_kind, mw, Lw, ldl, mw_post, T_post, ldt, Lw_post, ldlp, logpdf, info);            \
  }                                                                                                                 \
  int blr_posterior_dense_noise_##SUF(blr_handle* h, int memspace, int layout, int64_t D, int64_t N, const T* X,    \
                                      int64_t ldx, const T* y, const T* Sy, int64_t ldsy, int prior_kind,           \
                                      const T* mw, const T* Lw, int64_t ldl, T* mw_post, T* T_post, int64_t ldt,    \
                                      T* Lw_post, int64_t ldlp, double* logpdf, int32_t* info) {                    \
    return posterior_dense_noise<T>(h, memspace, layout, D, N, X, ldx, y, Sy, ldsy, prior_kind, mw, Lw, ldl,        \
                                    mw_post, T_post, ldt, Lw_post, ldlp, logpdf, info);                             \
  }                                                                                                                 \
  int blr_mean_and_cov_##SUF(blr_handle* h, int memspace, int layout, int64_t D, int64_t N, const T* X,             \
                             int64_t ldx, int noise_kind, const T* s, int64_t lds, int prior_kind, const T* mw,     \
                             const T* Lw, int64_t ldl, T* mean, T* C, int64_t ldc, int32_t* info) {                 \
    return mean_and_cov<T>(h, memspace, layout, D, N, X, ldx, noise_kind, s, lds, prior_kind, mw, Lw, ldl, mean,    \
                           C, ldc, info);                                                                           \
  }                                                                                                                 \
  int blr_apply_weights_##SUF(blr_handle* h, int memspace, int layout, int64_t D, int64_t N, int64_t S, const T* X, \
                              int64_t ldx, const T* W, int64_t ldw, T* Y, int64_t ldy) {                            \
    return apply_weights<T>(h, memspace, layout, D, N, S, X, ldx, W, ldw, Y, ldy);                                  \
  }                                                                                                                 \
  int blr_rand_dense_noise_##SUF(blr_handle* h, int memspace, int layout, int64_t D, int64_t N, int64_t S,          \
                                 const T* X, int64_t ldx, const T* Sy, int64_t ldsy, int prior_kind, const T* mw,   \
                                 const T* Lw, int64_t ldl, const T* Z1, int64_t ldz1, const T* Z2, int64_t ldz2,    \
                                 T* Y, int64_t ldy) {                                                               \
    return rand_dense_noise<T>(h, memspace, layout, D, N, S, X, ldx, Sy, ldsy, prior_kind, mw, Lw, ldl, Z1, ldz1,   \
                               Z2, ldz2, Y, ldy);                                                                   \
  }

BLR_DEFINE(f64, double)
BLR_DEFINE(f32, float)

// ---- multi-GPU exchange: RCCL called directly ------------------------------------------------------
int blr_comm_unique_id(void* id128) {
  if (!id128) return -1;
  if (!rccl().ok) return -2001;
  ncclUniqueId id;
  static_assert(sizeof(ncclUniqueId) == BLR_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");
  ncclResult_t r = rccl().GetUniqueId(&id);
  if (r != ncclSuccess) return -(2000 + (int)r);
  memcpy(id128, &id, sizeof id);
  return 0;
}

int blr_comm_init(blr_handle* h, int nranks, int rank, const void* id128) {
  if (!h) return -1;
  h->err.clear();
  if (nranks < 1) return bad_arg(h, 2, "nranks < 1");
  if (rank < 0 || rank >= nranks) return bad_arg(h, 3, "rank out of range");
  if (!id128) return bad_arg(h, 4, "unique id is NULL");
  if (h->comm) return bad_arg(h, 1, "the handle already has a communicator (blr_comm_destroy first)");
  if (!rccl().ok) { h->err = rccl().why; return -2001; }
  HIP_TRY(h, hipSetDevice(h->device));
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  ncclResult_t r = rccl().CommInitRank(&h->comm, nranks, id, rank);
  if (r != ncclSuccess) { h->comm = nullptr; return rccl_fail(h, r, "ncclCommInitRank"); }
  h->comm_size = nranks;
  h->comm_rank = rank;
  return 0;
}

int blr_comm_destroy(blr_handle* h) {
  if (!h) return -1;
  if (!h->comm) return 0;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  ncclResult_t r = rccl().CommDestroy(h->comm);
  h->comm = nullptr;
  h->comm_size = 0;
  h->comm_rank = 0;
  return r == ncclSuccess ? 0 : rccl_fail(h, r, "ncclCommDestroy");
}

int blr_comm_size(blr_handle* h) { return h && h->comm ? h->comm_size : (h ? 1 : -1); }
int blr_comm_rank(blr_handle* h) { return h && h->comm ? h->comm_rank : (h ? 0 : -1); }

int blr_logpdf_allgather_sum(blr_handle* h, int64_t count, const double* logpdf_local, double* logpdf_all, double* total) {
  if (!h) return -1;
  h->err.clear();
  if (count < 0 || count > ((int64_t)1 << 30)) return bad_arg(h, 2, "count out of range");
  if (count > 0 && !logpdf_local) return bad_arg(h, 3, "logpdf_local is NULL");
  if (!logpdf_all) return bad_arg(h, 4, "logpdf_all is NULL");
  if (!total) return bad_arg(h, 5, "total is NULL");
  HIP_TRY(h, hipSetDevice(h->device));
  const int64_t world = h->comm ? h->comm_size : 1;
  if (h->comm) {
    ncclResult_t r = rccl().AllGather(logpdf_local, logpdf_all, (size_t)count, ncclDouble, h->comm, h->stream);
    if (r != ncclSuccess) return rccl_fail(h, r, "ncclAllGather");
  } else if (count > 0 && logpdf_all != logpdf_local) {
    HIP_TRY(h, hipMemcpyAsync(logpdf_all, logpdf_local, (size_t)count * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  }
  // the SAME fixed-order sum over the SAME gathered vector on every rank: identical bits for any rank count
  hipLaunchKernelGGL(logpdf_sum_kernel, dim3(1), dim3(kThreads), 0, h->stream, (const double*)logpdf_all, count * world, total);
  HIP_TRY(h, hipGetLastError());
  if (!h->async) HIP_TRY(h, hipStreamSynchronize(h->stream));
  return 0;
}

int blr_allreduce_sum(blr_handle* h, int is_f64, void* buf, int64_t count) {
  if (!h) return -1;
  h->err.clear();
  if (!buf) return bad_arg(h, 3, "buf is NULL");
  if (count < 0) return bad_arg(h, 4, "count < 0");
  if (!h->comm || count == 0) return 0;  // one rank: the sum over ranks is the buffer itself
  HIP_TRY(h, hipSetDevice(h->device));
  ncclResult_t r = rccl().AllReduce(buf, buf, (size_t)count, is_f64 ? ncclDouble : ncclFloat, ncclSum, h->comm, h->stream);
  if (r != ncclSuccess) return rccl_fail(h, r, "ncclAllReduce");
  if (!h->async) HIP_TRY(h, hipStreamSynchronize(h->stream));
  return 0;
}

// One regressor, its observations split over the ranks of the handle's communicator: statistics of the local columns, ONE
// in-place all-reduce of the (DP + 128) x DP statistics matrix and one of the two scalars, redundant finish on every rank.
#define BLR_DEFINE_NSHARDED(SUF, T, IS64)                                                                               \
  int blr_posterior_nsharded_##SUF(blr_handle* h, int layout, int64_t D, int64_t N_local, int64_t N_total, const T* X,  \
                                   int64_t ldx, const T* y, int noise_kind, const T* s, int prior_kind, const T* mw,    \
                                   const T* Lw, int64_t ldl, T* stats, int64_t lds, double* scal, T* mw_post,           \
                                   T* T_post, int64_t ldt, T* Lw_post, int64_t ldlp, double* logpdf, int32_t* info) {   \
    if (!h) return -1;                                                                                                  \
    const int64_t DP = (D + 127) / 128 * 128;                                                                           \
    int rc = gram_stats<T>(h, layout, D, N_local, X, ldx, y, noise_kind, s, mw, stats, lds, scal);                     \
    if (rc) return rc;                                                                                                  \
    if ((rc = blr_allreduce_sum(h, IS64, stats, lds * DP))) return rc;                                                  \
    if ((rc = blr_allreduce_sum(h, 1, scal, 2))) return rc;                                                             \
    return posterior_from_stats<T>(h, D, N_total, stats, lds, scal, prior_kind, mw, Lw, ldl, mw_post, T_post, ldt,      \
                                   Lw_post, ldlp, logpdf, info);                                                        \
  }
BLR_DEFINE_NSHARDED(f64, double, 1)
BLR_DEFINE_NSHARDED(f32, float, 0)
#undef BLR_DEFINE_NSHARDED

int blr_logpdf_sum(blr_handle* h, int memspace, int64_t B, const double* logpdf, double* total) {
  if (!h) return -1;
  h->err.clear();
  if (memspace != BLR_MEM_HOST && memspace != BLR_MEM_DEVICE) return bad_arg(h, 2, "memspace");
  if (B < 0) return bad_arg(h, 3, "B < 0");
  if (B > 0 && !logpdf) return bad_arg(h, 4, "logpdf is NULL");
  if (!total) return bad_arg(h, 5, "total is NULL");
  HIP_TRY(h, hipSetDevice(h->device));
  Staging guard(h);
  const double* lp = logpdf;
  double* tot = total;
  int rc;
  if (memspace == BLR_MEM_HOST) {
    if ((rc = stage_in(h, logpdf, (size_t)B, &lp))) return rc;
    void* p = nullptr;
    HIP_TRY(h, hipMalloc(&p, sizeof(double)));
    h->staged.push_back(p);
    tot = static_cast<double*>(p);
  }
  hipLaunchKernelGGL(logpdf_sum_kernel, dim3(1), dim3(kThreads), 0, h->stream, lp, B, tot);
  HIP_TRY(h, hipGetLastError());
  if (memspace == BLR_MEM_HOST) {
    HIP_TRY(h, hipMemcpyAsync(total, tot, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  } else if (!h->async) {
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  }
  return 0;
}

}  // extern "C"
