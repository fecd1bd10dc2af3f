"""Host-side AddressSanitizer + UBSan pass over the C-ABI layer: blr_abi.hip is ~2.6 k lines of pointer / stride
arithmetic, validation and staging.  The sanitizer build (csrc/Makefile target `asan`; device code compiled as usual) is loaded
into a child interpreter with the ASan runtime preloaded and driven through everything that runs WITHOUT a GPU: symbol table,
argument validation of every entry point, handle-less calls, the RCCL binding's checks (tests/test_abi_cpu.py).  GPU-side
sanitizers are not available on the pool; the device code paths are covered by the -m gpu parity tests."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bayesianlinearregressors.jl_amd", "csrc")


def _asan_runtime():
    hits = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    return hits[-1] if hits else None


@pytest.mark.timeout(900)
def test_abi_layer_under_address_and_ub_sanitizer():
    rt = _asan_runtime()
    if rt is None or not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc / ASan runtime in this image")
    r = subprocess.run(["make", "-C", CSRC, "asan", "ARCH=gfx950"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lib = os.path.join(CSRC, "san", "libblr_mi355x_asan.so")
    env = dict(os.environ)
    env.update({"LD_PRELOAD": rt, "ASAN_OPTIONS": "detect_leaks=0:halt_on_error=1", "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1",
                "BLR_MI355X_LIB": lib})
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_abi_cpu.py"), "-q", "-x", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, env=env, cwd=ROOT)
    out = r.stdout + r.stderr
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert r.returncode == 0, out[-4000:]
    assert " passed" in out
