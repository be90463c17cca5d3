#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprof.  Logs go to gpurun_out/ (merged back).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
STAGES="${1:-test smoke bench prof}"
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu_info.log
nproc >> gpurun_out/gpu_info.log
for st in $STAGES; do
  case $st in
    test)
      timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
      echo "[session] pytest -m gpu rc=$?"; tail -n 25 gpurun_out/pytest_gpu.log ;;
    smoke)
      timeout 600 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1
      echo "[session] smoke rc=$?"; tail -n 5 gpurun_out/smoke.log ;;
    bench)
      timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2>&1
      echo "[session] bench rc=$?"; tail -n 3 gpurun_out/bench.log ;;
    benchfast)
      timeout 600 python bench.py --steps 20 --warmup 5 --cpu-steps 0 > gpurun_out/bench_fast.log 2>&1
      echo "[session] benchfast rc=$?"; tail -n 3 gpurun_out/bench_fast.log ;;
    prof)
      rm -rf gpurun_out/prof
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o ffno -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --cpu-steps 0 > "$OLDPWD/gpurun_out/prof.log" 2>&1)
      echo "[session] rocprof rc=$?"; tail -n 2 gpurun_out/prof.log
      db=$(find gpurun_out/prof -name "*.db" | head -1); python tools/rocpd_stats.py "$db" 7 > gpurun_out/kernel_stats.md 2>&1; head -n 24 gpurun_out/kernel_stats.md | cut -c1-170
      find gpurun_out/prof -name "*kernel_stats*" | head -3
      f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
      [ -n "$f" ] && head -n 25 "$f" | cut -c1-200
      # keep the merge-back small: drop the per-dispatch trace, keep stats
      find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete ;;
    bench256)
      timeout 900 python bench.py --steps 10 --warmup 3 --grid 256 --layers 12 --modes 32 --batch 2 --cpu-steps 1 > gpurun_out/bench_256.log 2>&1
      echo "[session] bench256 rc=$?"; tail -n 1 gpurun_out/bench_256.log | cut -c1-1200 ;;
    mesh3d)
      timeout 600 python tools/bench_mesh.py --preset plasticity > gpurun_out/bench_mesh3d_plasticity.log 2>&1
      echo "[session] mesh3d plasticity rc=$?"; tail -n 1 gpurun_out/bench_mesh3d_plasticity.log | cut -c1-400
      timeout 600 python tools/bench_mesh.py --preset cube64 > gpurun_out/bench_mesh3d_cube64.log 2>&1
      echo "[session] mesh3d cube64 rc=$?"; tail -n 1 gpurun_out/bench_mesh3d_cube64.log | cut -c1-400
      timeout 600 python tools/bench_mesh.py --preset airfoil > gpurun_out/bench_mesh2d_airfoil.log 2>&1
      echo "[session] mesh2d airfoil rc=$?"; tail -n 1 gpurun_out/bench_mesh2d_airfoil.log | cut -c1-400
      rm -rf gpurun_out/prof_mesh3d
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_mesh3d" -o m3 -- python "$OLDPWD/tools/bench_mesh.py" --preset plasticity --steps 5 --warmup 2 > "$OLDPWD/gpurun_out/prof_mesh3d.log" 2>&1)
      db=$(find gpurun_out/prof_mesh3d -name "*.db" | head -1); python tools/rocpd_stats.py "$db" > gpurun_out/mesh3d_kernel_stats.md 2>&1; head -n 30 gpurun_out/mesh3d_kernel_stats.md | cut -c1-180
      find gpurun_out/prof_mesh3d -size +20M -delete ;;
    zongyi)
      timeout 600 python tools/bench_zongyi.py > gpurun_out/bench_zongyi.log 2>&1
      echo "[session] zongyi rc=$?"; tail -n 2 gpurun_out/bench_zongyi.log | cut -c1-700
      rm -rf gpurun_out/prof_zongyi
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_zongyi" -o z -- python "$OLDPWD/tools/bench_zongyi.py" --steps 3 --warmup 1 --cpu 0 > "$OLDPWD/gpurun_out/prof_zongyi.log" 2>&1)
      db=$(find gpurun_out/prof_zongyi -name "*.db" | head -1); python tools/rocpd_stats.py "$db" > gpurun_out/zongyi_kernel_stats.md 2>&1; head -n 22 gpurun_out/zongyi_kernel_stats.md | cut -c1-180
      find gpurun_out/prof_zongyi -size +20M -delete ;;
    latency)
      timeout 300 python tools/bench_latency.py > gpurun_out/bench_latency.log 2>&1
      echo "[session] latency rc=$?"; tail -n 14 gpurun_out/bench_latency.log ;;
    dry2)
      # the N > 1 control flow of bench.py on ONE device (both ranks on cuda:0 over gloo: a dry run, not a measurement)
      FFNO_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --cpu-steps 0 > gpurun_out/bench_dry2.log 2>&1
      echo "[session] dry2 rc=$?"; tail -n 1 gpurun_out/bench_dry2.log | cut -c1-300 ;;
    bench19)
      timeout 600 python bench.py --steps 20 --warmup 5 --batch 19 --cpu-steps 0 > gpurun_out/bench_b19.log 2>&1
      echo "[session] bench19 rc=$?"; grep "timed region" gpurun_out/bench_b19.log ;;
    traincli)
      python - <<'PY' > gpurun_out/example_config.yaml
print("""routine:
  _target_: fourierflow.routines.Grid2DMarkovExperiment
  conv:
    _target_: fourierflow.modules.FNOFactorized2DBlock
    modes: 16
    width: 64
    n_layers: 24
    input_dim: 3
    share_weight: true
    factor: 4
    ff_weight_norm: true
    gain: 0.1
  n_steps: 10
  max_accumulations: 1000
  noise_std: 0.01
  optimizer:
    _target_: functools.partial
    _args_: ["${get_method: torch.optim.AdamW}"]
    lr: 0.0025
    weight_decay: 0.0001
  scheduler:
    scheduler:
      _target_: functools.partial
      _args_: ["${get_method: fourierflow.schedulers.CosineWithWarmupScheduler}"]
      num_warmup_steps: 500
      num_training_steps: 100000
      num_cycles: 0.5
builder:
  batch_size: 19
""")
PY
      timeout 600 python -m fourierflow_amd.train gpurun_out/example_config.yaml --steps 40 > gpurun_out/train_cli.log 2>&1
      echo "[session] traincli rc=$?"; tail -n 3 gpurun_out/train_cli.log ;;
    sq)
      rm -rf gpurun_out/pmc_SQ
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d "$OLDPWD/gpurun_out/pmc_SQ" -o ffno -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --cpu-steps 0 --no-secondary > "$OLDPWD/gpurun_out/pmc_SQ.log" 2>&1)
      echo "[session] pmc SQ rc=$?"; tail -n 2 gpurun_out/pmc_SQ.log | cut -c1-200
      db=$(find gpurun_out/pmc_SQ -name "*.db" | head -1); python tools/rocpd_pmc_multi.py "$db" ffno > gpurun_out/pmc_SQ.md 2>&1; head -n 12 gpurun_out/pmc_SQ.md | cut -c1-230
      find gpurun_out/pmc_SQ -name "*.db" -size +20M -delete ;;
    pmc)
      # HBM traffic counters, one PMC pass each (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2), kernel-trace only
      for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf gpurun_out/pmc_$c
        (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $c -d "$OLDPWD/gpurun_out/pmc_$c" -o ffno -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --cpu-steps 0 --no-secondary > "$OLDPWD/gpurun_out/pmc_$c.log" 2>&1)
        echo "[session] pmc $c rc=$?"
      done
      f=$(find gpurun_out/pmc_FETCH_SIZE -name "*.db" | head -1); w=$(find gpurun_out/pmc_WRITE_SIZE -name "*.db" | head -1)
      python tools/rocpd_pmc.py "$f" > gpurun_out/pmc_FETCH_SIZE.md; python tools/rocpd_pmc.py "$w" > gpurun_out/pmc_WRITE_SIZE.md
      (cd tools && python make_pmc_traffic.py "../$f" "../$w" "${FFNO_GIT_HEAD:-unknown}") > gpurun_out/pmc_traffic.json; head -c 600 gpurun_out/pmc_traffic.json
      find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name "*.db" -size +20M -delete ;;
  esac
done
