"""ctypes binding of the C ABI in include/blr_mi355x.h (libblr_mi355x.so, hand-written HIP for gfx950).

There is NO CPU fallback: if the shared library has not been built, or no MI355X is visible when a
handle is requested, this module raises.  Build with ``python -c "import __graft_entry__ as g; g.build()"``
(or ``make -C bayesianlinearregressors.jl_amd/csrc``).
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BLR_MI355X_LIB") or os.path.join(_HERE, "csrc", "libblr_mi355x.so")

LAYOUT_COLVECS, LAYOUT_ROWVECS = 0, 1
NOISE_ISOTROPIC, NOISE_DIAGONAL, NOISE_DENSE = 0, 1, 2
PRIOR_DENSE, PRIOR_UPPER_FACTOR, PRIOR_DIAGONAL = 0, 1, 2
MEM_HOST, MEM_DEVICE = 0, 1

_i64, _int, _vp = C.c_int64, C.c_int, C.c_void_p
_H = C.c_void_p


class BLRError(RuntimeError):
    """HIP/runtime failure or invalid argument reported by the library (negative return code)."""

    def __init__(self, code, msg):
        super().__init__(f"libblr_mi355x: code {code}: {msg}")
        self.code = code


class PosDefException(np.linalg.LinAlgError):
    """Mirrors LinearAlgebra.PosDefException(info): Cholesky broke at leading minor ``info``."""

    def __init__(self, info):
        super().__init__(f"matrix is not positive definite; Cholesky factorization failed (info = {info})")
        self.info = info


_SIGS = {
    "blr_abi_version": ([], _int),
    "blr_device_count": ([], _int),
    "blr_create": ([_int, C.POINTER(_H)], _int),
    "blr_destroy": ([_H], _int),
    "blr_last_error": ([_H], C.c_char_p),
    "blr_set_stream": ([_H, _vp], _int),
    "blr_reset_stream": ([_H], _int),
    "blr_set_async": ([_H, _int], _int),
    "blr_synchronize": ([_H], _int),
    "blr_set_option": ([_H, C.c_char_p, C.c_char_p], _int),
    "blr_release_workspace": ([_H], _int),
    "blr_last_route": ([_H], C.c_char_p),
    "blr_get_stat": ([_H, C.c_char_p, C.POINTER(_i64)], _int),
    "blr_reset_stats": ([_H], _int),
    "blr_device_alloc": ([_H, C.c_size_t, C.POINTER(_vp)], _int),
    "blr_device_free": ([_H, _vp], _int),
    "blr_memcpy_h2d": ([_H, _vp, _vp, C.c_size_t], _int),
    "blr_memcpy_d2h": ([_H, _vp, _vp, C.c_size_t], _int),
    "blr_timer_start": ([_H], _int),
    "blr_timer_stop": ([_H, C.POINTER(C.c_float)], _int),
    "blr_logpdf_sum": ([_H, _int, _i64, _vp, _vp], _int),
    "blr_comm_unique_id": ([_vp], _int),
    "blr_comm_init": ([_H, _int, _int, _vp], _int),
    "blr_comm_destroy": ([_H], _int),
    "blr_comm_size": ([_H], _int),
    "blr_comm_rank": ([_H], _int),
    "blr_logpdf_allgather_sum": ([_H, _i64, _vp, _vp, _vp], _int),
    "blr_allreduce_sum": ([_H, _int, _vp, _i64], _int),
}
for _suf in ("f64", "f32"):
    _SIGS[f"blr_posterior_batched_{_suf}"] = (
        [_H, _int, _int, _i64, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _int, _vp, _i64, _int, _vp, _i64, _vp, _i64, _i64,
         _vp, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _vp], _int)
    _SIGS[f"blr_posterior_nsharded_{_suf}"] = (
        [_H, _int, _i64, _i64, _i64, _vp, _i64, _vp, _int, _vp, _int, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp], _int)
    _SIGS[f"blr_update_factor_{_suf}"] = (
        [_H, _int, _int, _i64, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _int, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _vp, _vp], _int)
    _SIGS[f"blr_posterior_{_suf}"] = (
        [_H, _int, _i64, _i64, _vp, _i64, _vp, _int, _vp, _int, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp], _int)
    _SIGS[f"blr_marginals_batched_{_suf}"] = (
        [_H, _int, _int, _i64, _i64, _i64, _vp, _i64, _i64, _int, _vp, _i64, _int, _vp, _i64, _vp, _i64, _i64,
         _vp, _i64, _vp, _i64, _vp], _int)
    _SIGS[f"blr_rand_{_suf}"] = (
        [_H, _int, _int, _i64, _i64, _i64, _vp, _i64, _int, _vp, _int, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64], _int)
    _SIGS[f"blr_apply_weights_{_suf}"] = ([_H, _int, _int, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64], _int)
    _SIGS[f"blr_sample_weights_{_suf}"] = (
        [_H, _int, _i64, _i64, _int, _vp, _vp, _i64, _vp, _i64, _vp, _i64], _int)
    _SIGS[f"blr_logpdf_grad_batched_{_suf}"] = (
        [_H, _int, _int, _i64, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _int, _vp, _i64, _int, _vp, _i64, _vp, _i64, _i64,
         _vp, _vp, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _vp], _int)
    _SIGS[f"blr_logpdf_multi_{_suf}"] = (
        [_H, _int, _int, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _int, _vp, _int, _vp, _vp, _i64, _vp, _vp, _i64, _vp], _int)
    _SIGS[f"blr_gram_stats_{_suf}"] = ([_H, _int, _i64, _i64, _vp, _i64, _vp, _int, _vp, _vp, _vp, _i64, _vp], _int)
    _SIGS[f"blr_posterior_from_stats_{_suf}"] = (
        [_H, _i64, _i64, _vp, _i64, _vp, _int, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _vp], _int)
    _SIGS[f"blr_posterior_dense_noise_{_suf}"] = (
        [_H, _int, _int, _i64, _i64, _vp, _i64, _vp, _vp, _i64, _int, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _vp], _int)
    _SIGS[f"blr_mean_and_cov_{_suf}"] = (
        [_H, _int, _int, _i64, _i64, _vp, _i64, _int, _vp, _i64, _int, _vp, _vp, _i64, _vp, _vp, _i64, _vp], _int)
    _SIGS[f"blr_rand_dense_noise_{_suf}"] = (
        [_H, _int, _int, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _int, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64], _int)
    _fp = C.c_double if _suf == "f64" else C.c_float
    _SIGS[f"blr_rff_features_{_suf}"] = (
        [_H, _int, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _fp, _vp, _i64], _int)
    _SIGS[f"blr_posterior_rff_{_suf}"] = (
        [_H, _int, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _fp, _vp, _int, _vp, _int, _vp, _vp, _i64,
         _vp, _vp, _i64, _vp, _i64, _vp, _vp], _int)

EXPORTED_SYMBOLS = tuple(_SIGS)

_lib = None
_lib_lock = threading.Lock()


def load_library():
    """dlopen the HIP library (works without a GPU: only handle creation needs one)."""
    global _lib
    with _lib_lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise ImportError(
                    f"{LIB_PATH} is missing: the HIP extension has not been built. "
                    "Run `python -c 'import __graft_entry__ as g; g.build()'`. There is no CPU fallback."
                )
            if os.environ.get("BLR_MI355X_NO_TORCH_PRELOAD") != "1":
                # PyTorch wheels bundle their own libamdhip64/libhsa-runtime64.  Two HIP runtimes in one
                # process do not share the device ("No HIP GPUs are available" from whichever initialises
                # second), so when torch is installed let it load its runtime FIRST; our library then binds
                # to the already-loaded libamdhip64.so.  Without torch the system ROCm runtime is used.
                try:
                    import torch  # noqa: F401
                except Exception:
                    pass
            lib = C.CDLL(LIB_PATH)
            for name, (argtypes, restype) in _SIGS.items():
                fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
                fn.argtypes = argtypes
                fn.restype = restype
            _lib = lib
    return _lib


def _ptr(a):
    """void* of a numpy array, an int device pointer, or None."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(_vp)
    return _vp(int(a))


def suffix(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return "f64"
    if dtype == np.float32:
        return "f32"
    raise TypeError(f"unsupported element type {dtype}; the library is built for Float64 and Float32")


class Handle:
    """One library handle = one device + one stream + scratch.  Not thread-safe (one per thread)."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = _H()
        rc = self.lib.blr_create(int(device), C.byref(h))
        if rc != 0:
            raise BLRError(rc, "blr_create failed: no usable MI355X/HIP device (the product path has no CPU fallback)")
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self.lib.blr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- error translation ---------------------------------------------------------------------
    def check(self, rc):
        if rc < 0:
            msg = self.lib.blr_last_error(self._h)
            raise BLRError(rc, msg.decode() if msg else "")
        return rc

    def set_stream(self, stream_ptr):
        """Run on the caller's hipStream_t (0 / None = the HIP null stream, torch's default stream)."""
        self.check(self.lib.blr_set_stream(self._h, _vp(int(stream_ptr)) if stream_ptr else None))
        self._caller_stream = (int(stream_ptr) if stream_ptr else 0,)

    def reset_stream(self):
        self.check(self.lib.blr_reset_stream(self._h))
        self._caller_stream = None

    def current_stream_setting(self):
        """None = the handle's own stream, else a 1-tuple holding the caller stream pointer (0 = the null stream)."""
        return getattr(self, "_caller_stream", None)

    def set_async(self, flag):
        self.check(self.lib.blr_set_async(self._h, int(bool(flag))))

    def synchronize(self):
        self.check(self.lib.blr_synchronize(self._h))

    def release_workspace(self):
        """Free the handle's grow-only device scratch (include/blr_mi355x.h blr_release_workspace)."""
        self.check(self.lib.blr_release_workspace(self._h))

    def set_option(self, key, value=None):
        """Run-time switch of this handle (include/blr_mi355x.h blr_set_option); value None = the built-in default."""
        self.check(self.lib.blr_set_option(self._h, str(key).encode(), None if value is None else str(value).encode()))

    def last_route(self):
        """Kernel family the most recent posterior / logpdf dispatch launched (include/blr_mi355x.h blr_last_route)."""
        return self.lib.blr_last_route(self._h).decode()

    def get_stat(self, key):
        """Counter of this handle: "i8_regressors", "i8_handed_back", "planes_redone", "workspace_bytes" (blr_get_stat)."""
        v = _i64()
        self.check(self.lib.blr_get_stat(self._h, str(key).encode(), C.byref(v)))
        return int(v.value)

    def reset_stats(self):
        self.check(self.lib.blr_reset_stats(self._h))

    def timer_start(self):
        self.check(self.lib.blr_timer_start(self._h))

    def timer_stop(self):
        ms = C.c_float()
        self.check(self.lib.blr_timer_stop(self._h, C.byref(ms)))
        return float(ms.value)

    # -- device memory (blr_device_alloc / blr_device_free / blr_memcpy_*): for hosts without a GPU array library ----------
    def device_alloc(self, nbytes):
        """-> integer device pointer (0 bytes -> 0)."""
        if nbytes <= 0:
            return 0
        p = _vp()
        self.check(self.lib.blr_device_alloc(self._h, C.c_size_t(int(nbytes)), C.byref(p)))
        return int(p.value or 0)

    def device_free(self, dptr):
        if dptr and getattr(self, "_h", None):
            self.check(self.lib.blr_device_free(self._h, _vp(int(dptr))))

    def memcpy_h2d(self, dptr, host_array):
        a = np.ascontiguousarray(host_array) if not (host_array.flags.c_contiguous or host_array.flags.f_contiguous) else host_array
        if a.nbytes:
            self.check(self.lib.blr_memcpy_h2d(self._h, _vp(int(dptr)), a.ctypes.data_as(_vp), C.c_size_t(a.nbytes)))

    def memcpy_d2h(self, host_array, dptr):
        if not (host_array.flags.c_contiguous or host_array.flags.f_contiguous):
            raise ValueError("memcpy_d2h needs a contiguous destination")
        if host_array.nbytes:
            self.check(self.lib.blr_memcpy_d2h(self._h, host_array.ctypes.data_as(_vp), _vp(int(dptr)), C.c_size_t(host_array.nbytes)))

    # -- raw entry points (see include/blr_mi355x.h for argument meaning) ---------------------------
    def posterior_batched(self, dtype, memspace, layout, B, D, N, X, ldx, strideX, y, stridey, noise_kind, s, strides,
                          prior_kind, mw, stridemw, Lw, ldl, strideLw, mw_post, stride_mwpost, T_post, ldt, strideT,
                          Lw_post, ldlp, strideLp, logpdf, info):
        fn = getattr(self.lib, f"blr_posterior_batched_{suffix(dtype)}")
        return self.check(fn(self._h, memspace, layout, B, D, N, _ptr(X), ldx, strideX, _ptr(y), stridey, noise_kind,
                             _ptr(s), strides, prior_kind, _ptr(mw), stridemw, _ptr(Lw), ldl, strideLw,
                             _ptr(mw_post), stride_mwpost, _ptr(T_post), ldt, strideT, _ptr(Lw_post), ldlp, strideLp,
                             _ptr(logpdf), _ptr(info)))

    def posterior_nsharded(self, dtype, layout, D, N_local, N_total, X, ldx, y, noise_kind, s, prior_kind, mw, Lw, ldl, stats, lds,
                           scal, mw_post, T_post, ldt, Lw_post, ldlp, logpdf, info):
        """One regressor with its observations split over the ranks of comm_init (device pointers); blr_posterior_nsharded_*."""
        fn = getattr(self.lib, f"blr_posterior_nsharded_{suffix(dtype)}")
        return self.check(fn(self._h, layout, D, N_local, N_total, _ptr(X), ldx, _ptr(y), noise_kind, _ptr(s), prior_kind, _ptr(mw),
                             _ptr(Lw), ldl, _ptr(stats), lds, _ptr(scal), _ptr(mw_post), _ptr(T_post), ldt, _ptr(Lw_post), ldlp,
                             _ptr(logpdf), _ptr(info)))

    def update_factor(self, dtype, memspace, layout, B, D, k, X, ldx, strideX, y, stridey, noise_kind, s, strides, mw, stridemw,
                      T, ldt, strideT, logpdf, info):
        """In-place rank-k update of the resident state (mw, T); include/blr_mi355x.h blr_update_factor_*."""
        fn = getattr(self.lib, f"blr_update_factor_{suffix(dtype)}")
        return self.check(fn(self._h, memspace, layout, B, D, k, _ptr(X), ldx, strideX, _ptr(y), stridey, noise_kind, _ptr(s),
                             strides, _ptr(mw), stridemw, _ptr(T), ldt, strideT, _ptr(logpdf), _ptr(info)))

    def posterior(self, dtype, layout, D, N, X, ldx, y, noise_kind, s, prior_kind, mw, Lw, ldl, mw_post, T_post, ldt,
                  Lw_post, ldlp, logpdf):
        fn = getattr(self.lib, f"blr_posterior_{suffix(dtype)}")
        rc = self.check(fn(self._h, layout, D, N, _ptr(X), ldx, _ptr(y), noise_kind, _ptr(s), prior_kind, _ptr(mw),
                           _ptr(Lw), ldl, _ptr(mw_post), _ptr(T_post), ldt, _ptr(Lw_post), ldlp, _ptr(logpdf)))
        if rc > 0:
            raise PosDefException(rc)
        return rc

    def logpdf_grad_batched(self, dtype, memspace, layout, B, D, N, X, ldx, strideX, y, stridey, noise_kind, s, strides,
                            prior_kind, mw, stridemw, Lw, ldl, strideLw, logpdf, dX, lddx, stridedX, dy, stridedy, ds,
                            strideds, dmw, stridedmw, mw_post, stride_mwpost, Ainv, ldai, strideAi, info):
        fn = getattr(self.lib, f"blr_logpdf_grad_batched_{suffix(dtype)}")
        return self.check(fn(self._h, memspace, layout, B, D, N, _ptr(X), ldx, strideX, _ptr(y), stridey, noise_kind,
                             _ptr(s), strides, prior_kind, _ptr(mw), stridemw, _ptr(Lw), ldl, strideLw, _ptr(logpdf),
                             _ptr(dX), lddx, stridedX, _ptr(dy), stridedy, _ptr(ds), strideds, _ptr(dmw), stridedmw,
                             _ptr(mw_post), stride_mwpost, _ptr(Ainv), ldai, strideAi, _ptr(info)))

    def logpdf_multi(self, dtype, memspace, layout, D, N, S, X, ldx, Y, ldY, noise_kind, s, prior_kind, mw, Lw, ldl, logpdf,
                     mw_post, ldmp, info):
        fn = getattr(self.lib, f"blr_logpdf_multi_{suffix(dtype)}")
        return self.check(fn(self._h, memspace, layout, D, N, S, _ptr(X), ldx, _ptr(Y), ldY, noise_kind, _ptr(s), prior_kind,
                             _ptr(mw), _ptr(Lw), ldl, _ptr(logpdf), _ptr(mw_post), ldmp, _ptr(info)))

    def gram_stats(self, dtype, layout, D, N, X, ldx, y, noise_kind, s, mw, stats, lds, scal):
        fn = getattr(self.lib, f"blr_gram_stats_{suffix(dtype)}")
        return self.check(fn(self._h, layout, D, N, _ptr(X), ldx, _ptr(y), noise_kind, _ptr(s), _ptr(mw), _ptr(stats), lds, _ptr(scal)))

    def posterior_from_stats(self, dtype, D, N_total, stats, lds, scal, prior_kind, mw, Lw, ldl, mw_post, T_post, ldt, Lw_post,
                             ldlp, logpdf, info):
        fn = getattr(self.lib, f"blr_posterior_from_stats_{suffix(dtype)}")
        return self.check(fn(self._h, D, N_total, _ptr(stats), lds, _ptr(scal), prior_kind, _ptr(mw), _ptr(Lw), ldl,
                             _ptr(mw_post), _ptr(T_post), ldt, _ptr(Lw_post), ldlp, _ptr(logpdf), _ptr(info)))

    def marginals_batched(self, dtype, memspace, layout, B, D, N, X, ldx, strideX, noise_kind, s, strides, prior_kind,
                          mw, stridemw, Lw, ldl, strideLw, mean, stridemean, var, stridevar, info):
        fn = getattr(self.lib, f"blr_marginals_batched_{suffix(dtype)}")
        return self.check(fn(self._h, memspace, layout, B, D, N, _ptr(X), ldx, strideX, noise_kind, _ptr(s), strides,
                             prior_kind, _ptr(mw), stridemw, _ptr(Lw), ldl, strideLw, _ptr(mean), stridemean,
                             _ptr(var), stridevar, _ptr(info)))

    def rand(self, dtype, memspace, layout, D, N, S, X, ldx, noise_kind, s, prior_kind, mw, Lw, ldl, Z1, ldz1, Z2, ldz2,
             Y, ldy):
        fn = getattr(self.lib, f"blr_rand_{suffix(dtype)}")
        rc = self.check(fn(self._h, memspace, layout, D, N, S, _ptr(X), ldx, noise_kind, _ptr(s), prior_kind, _ptr(mw),
                           _ptr(Lw), ldl, _ptr(Z1), ldz1, _ptr(Z2), ldz2, _ptr(Y), ldy))
        if rc > 0:
            raise PosDefException(rc)
        return rc

    def apply_weights(self, dtype, memspace, layout, D, N, S, X, ldx, W, ldw, Y, ldy):
        """Y (N x S) = X'W for S given weight vectors; include/blr_mi355x.h blr_apply_weights_*."""
        fn = getattr(self.lib, f"blr_apply_weights_{suffix(dtype)}")
        return self.check(fn(self._h, memspace, layout, D, N, S, _ptr(X), ldx, _ptr(W), ldw, _ptr(Y), ldy))

    def sample_weights(self, dtype, memspace, D, S, prior_kind, mw, Lw, ldl, Z, ldz, W, ldw):
        fn = getattr(self.lib, f"blr_sample_weights_{suffix(dtype)}")
        rc = self.check(fn(self._h, memspace, D, S, prior_kind, _ptr(mw), _ptr(Lw), ldl, _ptr(Z), ldz, _ptr(W), ldw))
        if rc > 0:
            raise PosDefException(rc)
        return rc

    def rff_features(self, dtype, memspace, Din, D, N, Xin, ldxin, Omega, ldo, phase, scale, Phi, ldphi):
        fn = getattr(self.lib, f"blr_rff_features_{suffix(dtype)}")
        return self.check(fn(self._h, memspace, Din, D, N, _ptr(Xin), ldxin, _ptr(Omega), ldo, _ptr(phase), float(scale),
                             _ptr(Phi), ldphi))

    def posterior_rff(self, dtype, memspace, Din, D, N, Xin, ldxin, Omega, ldo, phase, scale, y, noise_kind, s, prior_kind,
                      mw, Lw, ldl, mw_post, T_post, ldt, Lw_post, ldlp, logpdf, info):
        fn = getattr(self.lib, f"blr_posterior_rff_{suffix(dtype)}")
        return self.check(fn(self._h, memspace, Din, D, N, _ptr(Xin), ldxin, _ptr(Omega), ldo, _ptr(phase), float(scale),
                             _ptr(y), noise_kind, _ptr(s), prior_kind, _ptr(mw), _ptr(Lw), ldl, _ptr(mw_post), _ptr(T_post),
                             ldt, _ptr(Lw_post), ldlp, _ptr(logpdf), _ptr(info)))

    def posterior_dense_noise(self, dtype, memspace, layout, D, N, X, ldx, y, Sy, ldsy, prior_kind, mw, Lw, ldl, mw_post, T_post, ldt,
                              Lw_post, ldlp, logpdf, info):
        fn = getattr(self.lib, f"blr_posterior_dense_noise_{suffix(dtype)}")
        return self.check(fn(self._h, memspace, layout, D, N, _ptr(X), ldx, _ptr(y), _ptr(Sy), ldsy, prior_kind, _ptr(mw), _ptr(Lw), ldl,
                             _ptr(mw_post), _ptr(T_post), ldt, _ptr(Lw_post), ldlp, _ptr(logpdf), _ptr(info)))

    def mean_and_cov(self, dtype, memspace, layout, D, N, X, ldx, noise_kind, s, lds, prior_kind, mw, Lw, ldl, mean, Cov, ldc, info):
        fn = getattr(self.lib, f"blr_mean_and_cov_{suffix(dtype)}")
        return self.check(fn(self._h, memspace, layout, D, N, _ptr(X), ldx, noise_kind, _ptr(s), lds, prior_kind, _ptr(mw), _ptr(Lw), ldl,
                             _ptr(mean), _ptr(Cov), ldc, _ptr(info)))

    def rand_dense_noise(self, dtype, memspace, layout, D, N, S, X, ldx, Sy, ldsy, prior_kind, mw, Lw, ldl, Z1, ldz1, Z2, ldz2, Y, ldy):
        fn = getattr(self.lib, f"blr_rand_dense_noise_{suffix(dtype)}")
        rc = self.check(fn(self._h, memspace, layout, D, N, S, _ptr(X), ldx, _ptr(Sy), ldsy, prior_kind, _ptr(mw), _ptr(Lw), ldl,
                           _ptr(Z1), ldz1, _ptr(Z2), ldz2, _ptr(Y), ldy))
        if rc > 0:
            raise PosDefException(rc)
        return rc

    # -- RCCL called directly (one rank per handle) ------------------------------------------------------------------
    @staticmethod
    def comm_unique_id():
        """128 opaque bytes (rank 0 calls this and ships them to the other ranks)."""
        buf = C.create_string_buffer(128)
        rc = load_library().blr_comm_unique_id(C.cast(buf, _vp))
        if rc != 0:
            raise BLRError(rc, "blr_comm_unique_id failed (librccl missing?)")
        return buf.raw

    def comm_init(self, nranks, rank, unique_id):
        buf = C.create_string_buffer(bytes(unique_id), 128)
        return self.check(self.lib.blr_comm_init(self._h, int(nranks), int(rank), C.cast(buf, _vp)))

    def comm_destroy(self):
        return self.check(self.lib.blr_comm_destroy(self._h))

    def comm_size(self):
        return self.lib.blr_comm_size(self._h)

    def comm_rank(self):
        return self.lib.blr_comm_rank(self._h)

    def logpdf_allgather_sum(self, count, logpdf_local, logpdf_all, total):
        return self.check(self.lib.blr_logpdf_allgather_sum(self._h, count, _ptr(logpdf_local), _ptr(logpdf_all), _ptr(total)))

    def allreduce_sum(self, is_f64, buf, count):
        return self.check(self.lib.blr_allreduce_sum(self._h, int(bool(is_f64)), _ptr(buf), count))

    def logpdf_sum(self, memspace, B, logpdf, total):
        return self.check(self.lib.blr_logpdf_sum(self._h, memspace, B, _ptr(logpdf), _ptr(total)))


_default = {}
_default_lock = threading.Lock()


def default_handle(device=0):
    """Process-wide handle per device (created on first use; raises without a GPU)."""
    with _default_lock:
        key = (os.getpid(), device)
        if key not in _default:
            _default[key] = Handle(device)
        return _default[key]
