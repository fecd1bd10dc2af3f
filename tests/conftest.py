import json
import os
import socket
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def repo_root():
    return ROOT


# ---- bench.py with two ranks on a one-GPU box (tests/test_gpu_parity.py::test_bench_two_ranks_...) -------------------------------
# The launcher and its workers must be STARTED before this process touches the GPU (a process that has initialised HIP must
# not fork + exec on this pool), so they are spawned here, right after collection and before the first test runs; the test
# itself only waits for them and reads their output.  torch.cuda.device_count() does not initialise the GPU.
_BENCH_RUNS = {}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def pytest_collection_finish(session):
    if not any("bench_two_rank_runs" in getattr(item, "fixturenames", ()) for item in session.items):
        return
    try:
        import torch

        if torch.cuda.device_count() < 1:
            return
    except Exception:
        return
    tmp = tempfile.mkdtemp(prefix="blr_bench2_")
    common = ["bench.py", "--config", "c4", "--steps", "3", "--warmup", "1", "--cpu-seconds", "0", "--secondary", "0"]
    env = dict(os.environ, BLR_BENCH_BACKEND="gloo", BLR_BENCH_SAME_DEVICE="1", MASTER_ADDR="127.0.0.1")
    runs = {
        "two": [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port())] + common + ["--gpus", "2"],
        "one": [sys.executable] + common + ["--gpus", "1"],
        # run BARE, the way the driver calls `python bench.py --gpus 1`: bench.py itself must spawn the two ranks
        "bare2": [sys.executable] + common + ["--gpus", "2"],
    }
    # one rank with the exchange on the LIBRARY's RCCL binding (a one-rank communicator on a side stream), and the same with
    # the binding's set-up forced to fail: the job must still end with its line, on the torch.distributed / local-sum path
    runs["comm1"] = [sys.executable] + common + ["--gpus", "1"]
    runs["commfail"] = [sys.executable] + common + ["--gpus", "1"]
    extra = {"comm1": {"BLR_BENCH_FORCE_COMM": "1"}, "commfail": {"BLR_BENCH_FORCE_COMM": "1", "BLR_BENCH_FAIL_LIB_COMM": "1"}}
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    for key, cmd in runs.items():
        out = open(os.path.join(tmp, key + ".out"), "w")
        err = open(os.path.join(tmp, key + ".err"), "w")
        _BENCH_RUNS[key] = (subprocess.Popen(cmd, cwd=ROOT, env=dict(env, **extra.get(key, {})), stdout=out, stderr=err), out.name, err.name)


@pytest.fixture(scope="session")
def bench_two_rank_runs():
    """-> {"one": json line of the 1-rank run, "two": json line of the 2-rank run} (both at --config c4), or skips."""
    if not _BENCH_RUNS:
        pytest.skip("the bench runs were not started (no GPU visible at collection time)")
    res = {}
    for key, (proc, out, err) in _BENCH_RUNS.items():
        rc = proc.wait(timeout=900)
        text = open(out).read()
        assert rc == 0, f"bench ({key}) exited {rc}: {open(err).read()[-2000:]}"
        lines = [ln for ln in text.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, f"bench ({key}) must print ONE JSON line, got {len(lines)}"
        res[key] = json.loads(lines[0])
        res[key]["_stderr"] = open(err).read()[-4000:]
    return res


def pytest_sessionfinish(session, exitstatus):
    for proc, _, _ in _BENCH_RUNS.values():
        if proc.poll() is None:
            proc.kill()
