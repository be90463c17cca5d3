#!/bin/bash
# round-2 session C: role-split feed-forward kernels vs the in-phase ones
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_ffx.py tests/test_bench_geometry.py -m gpu -q --maxfail=10 --tb=short -p no:cacheprovider -s > gpurun_out/pytest_gpu_c.log 2>&1
echo "[session] pytest subset rc=$?"; grep "^\[" gpurun_out/pytest_gpu_c.log | tail; tail -n 3 gpurun_out/pytest_gpu_c.log
for v in 1 0 1; do
  timeout 600 python bench.py --steps 20 --warmup 5 --cpu-steps 0 --ffx-schedule $v > gpurun_out/bench_sched$v.log 2>&1
  echo "[session] bench ffx-schedule $v rc=$?"; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/bench_sched$v.log").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], {k:(v["avg_us"]) for k,v in d["kernels"].items()}, "fwd", d["ms_per_forward"], "b1", d["ms_per_forward_batch1"])
except Exception as e:
    print("parse failed", e); print(open("gpurun_out/bench_sched$v.log").read()[-1500:])
PY
done
rm -rf gpurun_out/prof
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o ffno -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --cpu-steps 0 > "$OLDPWD/gpurun_out/prof.log" 2>&1)
echo "[session] rocprof rc=$?"
db=$(find gpurun_out/prof -name "*.db" | head -1); python tools/rocpd_stats.py "$db" 7 > gpurun_out/kernel_stats.md 2>&1; head -n 12 gpurun_out/kernel_stats.md | cut -c1-200
find gpurun_out/prof -size +20M -delete
