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
    spectests)
      timeout 900 python -m pytest tests/test_kernels_spectral.py tests/test_storage_bf16.py -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/pytest_spec.log 2>&1
      echo "[r4] spectral tests rc=$?"; tail -n 4 gpurun_out/pytest_spec.log
      timeout 900 python -m pytest tests/test_bench_geometry.py -m gpu -q -s --tb=short -p no:cacheprovider -k "B32 or 12L or mesh3d or bf16" > gpurun_out/pytest_geo.log 2>&1
      echo "[r4] geometry tests rc=$?"; grep -E "^\[|passed|failed" gpurun_out/pytest_geo.log | tail -12 ;;
    sq256)
      # SQ counters of the 256 x 256 / 64-mode step (its own PMC pass: kernel-trace + pmc only)
      rm -rf gpurun_out/pmc_SQ256
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d "$OLDPWD/gpurun_out/pmc_SQ256" -o ffno -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --cpu-steps 0 --no-secondary --grid 256 --layers 24 --modes 64 --batch 2 > "$OLDPWD/gpurun_out/pmc_SQ256.log" 2>&1)
      echo "[r4] pmc SQ256 rc=$?"
      db=$(find gpurun_out/pmc_SQ256 -name "*.db" | head -1); python tools/rocpd_pmc_multi.py "$db" ffno > gpurun_out/pmc_SQ256.md 2>&1; head -n 14 gpurun_out/pmc_SQ256.md | cut -c1-260
      rm -rf gpurun_out/pmc_SQ2
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d "$OLDPWD/gpurun_out/pmc_SQ2" -o ffno -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --cpu-steps 0 --no-secondary --grid 256 --layers 24 --modes 64 --batch 2 > "$OLDPWD/gpurun_out/pmc_SQ2.log" 2>&1)
      echo "[r4] pmc SQ2 rc=$?"; tail -n 3 gpurun_out/pmc_SQ2.log | cut -c1-200
      db=$(find gpurun_out/pmc_SQ2 -name "*.db" | head -1); python tools/rocpd_pmc_multi.py "$db" ffno > gpurun_out/pmc_SQ2.md 2>&1; head -n 14 gpurun_out/pmc_SQ2.md | cut -c1-260
      find gpurun_out/pmc_SQ256 gpurun_out/pmc_SQ2 -name "*.db" -size +20M -delete ;;
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
