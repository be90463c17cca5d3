#!/bin/bash
# round-2 evidence session: the driver's three commands (pytest -m gpu, smoke, bench) + rocprofv3 kernel trace of the bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 --tb=short -p no:cacheprovider -s > gpurun_out/pytest_gpu.log 2>&1
echo "[session] pytest -m gpu rc=$?"; grep "^\[bench" gpurun_out/pytest_gpu.log | tail -6; tail -n 3 gpurun_out/pytest_gpu.log
timeout 600 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1
echo "[session] smoke rc=$?"; tail -n 3 gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err
echo "[session] bench rc=$?"; tail -n 3 gpurun_out/bench.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench.log").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "fwd", d["ms_per_forward"], "b1", d["ms_per_forward_batch1"])
print({k:(v["avg_us"], v["in_step_event_us"], v["frac_hbm"], v["frac_mfma"], v["bound"]) for k,v in d["kernels"].items()})
print("roofline", {k:v for k,v in d["roofline"].items() if k not in ("note","byte_formula","floor_formula","traffic_source")})
print("cpu", {k:v for k,v in d["cpu_baseline"].items() if k != "sample"})
print("secondary", json.dumps(d["secondary"]))
PY
rm -rf gpurun_out/prof
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o ffno -- python "$OLDPWD/bench.py" --steps 20 --warmup 5 --cpu-steps 0 --no-secondary > "$OLDPWD/gpurun_out/prof.log" 2>&1)
echo "[session] rocprof rc=$?"
db=$(find gpurun_out/prof -name "*.db" | head -1); python tools/rocpd_stats.py "$db" 7 > gpurun_out/kernel_stats.md 2>&1; head -n 14 gpurun_out/kernel_stats.md | cut -c1-200
find gpurun_out/prof -size +20M -delete
