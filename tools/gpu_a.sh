#!/bin/bash
# GPU batch A: parity suite, then the driver's bench command exactly as the driver runs it; prints the headline line and the
# per-entry lines that precede it
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout ${PYTEST_TIMEOUT:-1500} python -m pytest tests -m gpu -x -q ${PYTEST_K:+-k "$PYTEST_K"} ${PYTEST_ARGS:-} > gpurun_out/pytest_a.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_a.log
[ -n "${SKIP_BENCH:-}" ] && exit 0
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_a.out 2> gpurun_out/bench_a.err; echo "bench rc=$?"
tail -c 600 gpurun_out/bench_a.err
echo "stdout bytes: $(wc -c < gpurun_out/bench_a.out), lines: $(wc -l < gpurun_out/bench_a.out), last line bytes: $(tail -1 gpurun_out/bench_a.out | wc -c)"
python3 - <<'PY'
import json
lines = open("gpurun_out/bench_a.out").read().strip().splitlines()
for l in lines[:-1]:
    d = json.loads(l)
    if "error" in d: print(d["secondary"], "ERROR", d["error"]); continue
    print(f"{d['secondary']:32s} {d['ms']:9.4f} ms  {d['per_s']:12.4g} {d['unit']:14s} {d['bound']:5s} frac {d['frac']:.3f}  traffic_x {d['traffic_x']}  {d['kernel']}")
h = json.loads(lines[-1])
print(json.dumps(h, indent=None)[:3000])
PY
