#!/bin/bash
# round-4 GPU batch A: parity suite, sustained clock probe, the driver's bench line with every secondary entry
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_a.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_a.log
bash tools/sustained_probe.sh > /dev/null 2>&1; echo "probe rc=$?"
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err; echo "bench rc=$?"
tail -c 600 gpurun_out/bench_a.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_a.json"))
print("value", d["value"], "frac", d["roofline"]["frac"])
for k, v in d.get("secondary", {}).items():
    if "error" in v: print(k, "ERROR", v["error"][:300])
    else: print(f"{k:32s} {v['ms']:9.4f} ms  {v['per_s']:14.4g} {v['unit']:14s} {v['roofline']['bound']:5s} frac {v['roofline']['frac']:.3f}")
PY
cat gpurun_out/microbench/ring_probe_sustained.txt
