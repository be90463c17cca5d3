#!/bin/bash
# Round-4 GPU sessions (stages by name; logs under gpurun_out/, merged back by gpurun).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for st in "$@"; do
  case $st in
    ubench)
      timeout 120 tools/ubench/bin/stream_patterns > gpurun_out/ubench_stream.log 2>&1
      echo "[r4] ubench rc=$?"; cat gpurun_out/ubench_stream.log ;;
    newtests)
      timeout 1500 python -m pytest tests/test_bench_geometry.py tests/test_trainer.py tests/test_rollout.py tests/test_capi_exports.py -m gpu -q -s --tb=short -p no:cacheprovider > gpurun_out/pytest_new.log 2>&1
      echo "[r4] new tests rc=$?"; grep -E "^\[|passed|failed|Error|error" gpurun_out/pytest_new.log | tail -40 ;;
    benchfast)
      timeout 600 python bench.py --steps 20 --warmup 5 --cpu-steps 0 > gpurun_out/bench_fast.log 2> gpurun_out/bench_fast.err
      echo "[r4] benchfast rc=$?"; tail -n 12 gpurun_out/bench_fast.err; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/bench_fast.log").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "ms_per_step", "ms_per_step_median", "ms_per_step_min", "ms_per_forward", "ms_per_forward_batch1", "git_head", "lib_source_stamp")})
    print("roofline", {k: d["roofline"][k] for k in ("kernel", "achieved", "frac", "avg_launch_us", "traffic")})
    print({n: k["avg_us"] for n, k in d["kernels"].items()})
    for s in d["secondary"] or []:
        print(s.get("workload", "?")[:50], s.get("value"), s.get("spectral") or s.get("roofline"), s.get("kernel_us_replay"))
except Exception as e:
    print("parse error", e)
PY
      ;;
  esac
done
