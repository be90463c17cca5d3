#!/bin/bash
# Round-5 GPU sessions (stages by name; logs under gpurun_out/, merged back by gpurun).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for st in "$@"; do
  case $st in
    ubench)
      timeout 120 tools/ubench/bin/stream_patterns > gpurun_out/ubench_stream.log 2>&1
      echo "[r5] ubench rc=$?"; cat gpurun_out/ubench_stream.log ;;
    newtests)
      timeout 1500 python -m pytest tests/test_bench_geometry.py tests/test_trainer.py tests/test_rollout.py tests/test_capi_exports.py -m gpu -q -s --tb=short -p no:cacheprovider > gpurun_out/pytest_new.log 2>&1
      echo "[r5] new tests rc=$?"; grep -E "^\[|passed|failed|Error|error" gpurun_out/pytest_new.log | tail -40 ;;
    spectests)
      timeout 900 python -m pytest tests/test_kernels_spectral.py tests/test_storage_bf16.py -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/pytest_spec.log 2>&1
      echo "[r5] spectral tests rc=$?"; tail -n 4 gpurun_out/pytest_spec.log
      timeout 900 python -m pytest tests/test_bench_geometry.py -m gpu -q -s --tb=short -p no:cacheprovider -k "B32 or 12L or mesh3d or bf16" > gpurun_out/pytest_geo.log 2>&1
      echo "[r5] geometry tests rc=$?"; grep -E "^\[|passed|failed" gpurun_out/pytest_geo.log | tail -12 ;;
    sq256)
      # SQ counters of the 256 x 256 / 64-mode step (its own PMC pass: kernel-trace + pmc only)
      rm -rf gpurun_out/pmc_SQ256
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d "$OLDPWD/gpurun_out/pmc_SQ256" -o ffno -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --cpu-steps 0 --no-secondary --grid 256 --layers 24 --modes 64 --batch 2 > "$OLDPWD/gpurun_out/pmc_SQ256.log" 2>&1)
      echo "[r5] pmc SQ256 rc=$?"
      db=$(find gpurun_out/pmc_SQ256 -name "*.db" | head -1); python tools/rocpd_pmc_multi.py "$db" ffno > gpurun_out/pmc_SQ256.md 2>&1; head -n 14 gpurun_out/pmc_SQ256.md | cut -c1-260
      rm -rf gpurun_out/pmc_SQ2
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d "$OLDPWD/gpurun_out/pmc_SQ2" -o ffno -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --cpu-steps 0 --no-secondary --grid 256 --layers 24 --modes 64 --batch 2 > "$OLDPWD/gpurun_out/pmc_SQ2.log" 2>&1)
      echo "[r5] pmc SQ2 rc=$?"; tail -n 3 gpurun_out/pmc_SQ2.log | cut -c1-200
      db=$(find gpurun_out/pmc_SQ2 -name "*.db" | head -1); python tools/rocpd_pmc_multi.py "$db" ffno > gpurun_out/pmc_SQ2.md 2>&1; head -n 14 gpurun_out/pmc_SQ2.md | cut -c1-260
      find gpurun_out/pmc_SQ256 gpurun_out/pmc_SQ2 -name "*.db" -size +20M -delete ;;
    profall)
      # rocprofv3 kernel stats, one file per benchmarked configuration
      prof() {  # tag, steps-in-run, command...
        tag=$1; shift; n=$1; shift
        rm -rf gpurun_out/prof_$tag
        (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_$tag" -o p -- "$@" > "$OLDPWD/gpurun_out/prof_$tag.log" 2>&1)
        echo "[r5] rocprof $tag rc=$?"
        db=$(find gpurun_out/prof_$tag -name "*.db" | head -1); python tools/rocpd_stats.py "$db" $n > gpurun_out/r05_${tag}_kernel_stats.md 2>&1
        head -n 12 gpurun_out/r05_${tag}_kernel_stats.md | cut -c1-150
        [ "$tag" = markov24 ] && python tools/rocpd_idle.py "$db" > gpurun_out/r05_markov24_step_idle.md 2>&1
        find gpurun_out/prof_$tag -type f -size +1M -delete
      }
      R=$PWD
      prof markov24 7 python $R/bench.py --steps 5 --warmup 2 --cpu-steps 0 --no-secondary
      prof kochkov256_k32 7 python $R/bench.py --steps 5 --warmup 2 --cpu-steps 0 --no-secondary --grid 256 --layers 12 --modes 32 --batch 2
      prof kochkov256_k64 7 python $R/bench.py --steps 5 --warmup 2 --cpu-steps 0 --no-secondary --grid 256 --layers 24 --modes 64 --batch 2
      prof cube64 7 python $R/tools/bench_mesh.py --preset cube64 --steps 5 --warmup 2
      ;;
    pmcall)
      # HBM traffic counters, one PMC pass each, per configuration (kernel-trace + pmc only)
      pmc() {  # tag, label, command...
        tag=$1; shift; label=$1; shift
        for c in FETCH_SIZE WRITE_SIZE; do
          rm -rf gpurun_out/pmc_${tag}_$c
          (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d "$OLDPWD/gpurun_out/pmc_${tag}_$c" -o p -- "$@" > "$OLDPWD/gpurun_out/pmc_${tag}_$c.log" 2>&1)
          echo "[r5] pmc $tag $c rc=$?"
        done
        f=$(find gpurun_out/pmc_${tag}_FETCH_SIZE -name "*.db" | head -1); w=$(find gpurun_out/pmc_${tag}_WRITE_SIZE -name "*.db" | head -1)
        (cd tools && python make_pmc_traffic.py "../$f" "../$w" "${FFNO_GIT_HEAD:-unknown}" "$label") > gpurun_out/pmc_traffic_$tag.json
        head -c 700 gpurun_out/pmc_traffic_$tag.json; echo
        find gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE -type f -size +1M -delete
      }
      R=$PWD
      pmc markov24 "markov/24 B=32 64x64 fp32 (bench.py defaults)" python $R/bench.py --steps 3 --warmup 1 --cpu-steps 0 --no-secondary
      pmc kochkov256_k32 "256x256 12L K=32 B=2 fp32" python $R/bench.py --steps 3 --warmup 1 --cpu-steps 0 --no-secondary --grid 256 --layers 12 --modes 32 --batch 2
      pmc kochkov256_k64 "256x256 24L K=64 B=2 fp32" python $R/bench.py --steps 3 --warmup 1 --cpu-steps 0 --no-secondary --grid 256 --layers 24 --modes 64 --batch 2
      pmc cube64 "64^3 (72^3 padded) width 32 12L B=1 fp32" python $R/tools/bench_mesh.py --preset cube64 --steps 3 --warmup 1
      ;;
    final)
      timeout 600 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1
      echo "[r5] smoke rc=$?"; tail -n 2 gpurun_out/smoke.log
      timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2> gpurun_out/bench.err
      echo "[r5] bench rc=$?"; tail -n 4 gpurun_out/bench.err
      timeout 1700 python -m pytest tests -m gpu -q --maxfail=25 --tb=short -p no:cacheprovider -s --durations=25 > gpurun_out/pytest_gpu.log 2>&1
      echo "[r5] pytest -m gpu rc=$?"; tail -n 6 gpurun_out/pytest_gpu.log ;;
    benchfast)
      timeout 600 python bench.py --steps 20 --warmup 5 --cpu-steps 0 > gpurun_out/bench_fast.log 2> gpurun_out/bench_fast.err
      echo "[r5] benchfast rc=$?"; tail -n 12 gpurun_out/bench_fast.err; python - <<'PY'
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
