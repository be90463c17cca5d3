#!/bin/bash
# Round-6 GPU sessions (stages by name; logs under gpurun_out/, merged back by gpurun).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for st in "$@"; do
  case $st in
    powerprobe)
      # which power / clock sources can an ordinary user read on this box?
      {
        echo "== sysfs hwmon"; for d in /sys/class/drm/card*/device/hwmon/hwmon*; do echo "$d"; ls "$d" 2>/dev/null | tr '\n' ' '; echo; for f in power1_average power1_input power1_cap freq1_input freq2_input temp1_input; do [ -r "$d/$f" ] && echo "$f=$(cat $d/$f 2>&1)"; done; done
        echo "== pp_dpm"; for c in /sys/class/drm/card*/device; do for f in pp_dpm_sclk pp_dpm_mclk gpu_busy_percent; do [ -r "$c/$f" ] && { echo "$c/$f"; cat "$c/$f" 2>&1 | head -12; }; done; done
        echo "== gpu_metrics"; ls -la /sys/class/drm/card*/device/gpu_metrics 2>&1
        echo "== amd-smi"; which amd-smi; timeout 60 amd-smi metric -p -c --json 2>&1 | head -60
        echo "== rocm-smi"; which rocm-smi; timeout 60 rocm-smi --showpower --showclocks --json 2>&1 | head -40
      } > gpurun_out/powerprobe.log 2>&1
      echo "[r6] powerprobe rc=$?"; head -c 6000 gpurun_out/powerprobe.log ;;
    benchfast)
      timeout 900 python bench.py --steps 20 --warmup 5 --cpu-steps 0 > gpurun_out/bench_fast.log 2> gpurun_out/bench_fast.err
      echo "[r6] benchfast rc=$?"; tail -n 12 gpurun_out/bench_fast.err; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/bench_fast.log").read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "ms_per_step", "ms_per_step_median", "ms_per_step_min", "ms_per_forward", "ms_per_forward_batch1", "git_head", "lib_source_stamp")})
    print("roofline", {k: d["roofline"].get(k) for k in ("kernel", "achieved", "frac", "avg_launch_us", "traffic")})
    print("roofline_forward", d.get("roofline_forward"))
    print({n: k["avg_us"] for n, k in d["kernels"].items()})
    for s in d["secondary"] or []:
        print(s.get("workload", "?")[:50], s.get("value"), s.get("spectral") or s.get("roofline"), s.get("kernel_us_replay"))
except Exception as e:
    print("parse error", e)
PY
      ;;
    infer)
      timeout 600 python -m pytest tests/test_infer_layer.py -m gpu -q -x -s --tb=short -p no:cacheprovider > gpurun_out/pytest_infer.log 2>&1
      echo "[r6] infer tests rc=$?"; grep -E "^\[|passed|failed|Error|error" gpurun_out/pytest_infer.log | tail -20
      timeout 300 python tools/time_infer.py > gpurun_out/time_infer.log 2>&1; echo "[r6] time_infer rc=$?"; tail -n 12 gpurun_out/time_infer.log
      timeout 300 python tools/time_infer.py 1 64 64 16 > gpurun_out/time_infer_b1.log 2>&1; echo "[r6] time_infer b1 rc=$?"; tail -n 8 gpurun_out/time_infer_b1.log ;;
    inferring)
      for d in ${INFER_RINGS:-1 2 3 4 1 2 3 4}; do echo "== FFNO_INFER_RING=$d"; FFNO_INFER_RING=$d timeout 300 python tools/time_infer.py 2>&1 | grep -E "K2|K1 |rel-L2"; done ;;
    fullsize)
      timeout 1500 python -m pytest tests/test_infer_layer.py tests/test_bench_geometry.py tests/test_real_mesh_shapes.py -m gpu -q -s --tb=short -p no:cacheprovider --durations=15 > gpurun_out/pytest_fullsize.log 2>&1
      echo "[r6] fullsize tests rc=$?"; grep -E "^\[|passed|failed|Error|error|^[0-9.]+s " gpurun_out/pytest_fullsize.log | tail -60 ;;
    pmcinfer)
      R=$PWD
      for pass in "A SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "B SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
        set -- $pass; n=$1; shift
        rm -rf gpurun_out/pmc_infer_$n
        (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d "$R/gpurun_out/pmc_infer_$n" -o ffno -- python "$R/tools/time_infer.py" > "$R/gpurun_out/pmc_infer_$n.log" 2>&1)
        echo "[r6] pmc infer $n rc=$?"
        db=$(find gpurun_out/pmc_infer_$n -name "*.db" | head -1); python tools/rocpd_pmc_multi.py "$db" ffno > gpurun_out/pmc_infer_$n.md 2>&1
        find gpurun_out/pmc_infer_$n -type f -size +1M -delete
        cut -c1-400 gpurun_out/pmc_infer_$n.md
      done ;;
    power)
      timeout 900 python tools/power_trace.py --seconds ${POWER_SECONDS:-5} > gpurun_out/power_trace.log 2>&1; echo "[r6] power rc=$?"; tail -n 45 gpurun_out/power_trace.log | cut -c1-330 ;;
    lvl1)
      timeout 600 python -m pytest tests/test_kernels_spectral.py tests/test_block.py -m gpu -q -x --tb=short -p no:cacheprovider -k "spectral2d or standalone" > gpurun_out/pytest_lvl1.log 2>&1
      echo "[r6] level-1 tests rc=$?"; tail -n 3 gpurun_out/pytest_lvl1.log
      timeout 300 python tools/time_spectral2d.py > gpurun_out/time_spectral2d.log 2>&1; echo "[r6] time_spectral2d rc=$?"; tail -n 5 gpurun_out/time_spectral2d.log ;;
    profall)
      # rocprofv3 kernel stats, one file per benchmarked configuration
      prof() {  # tag, steps-in-run, command...
        tag=$1; shift; n=$1; shift
        rm -rf gpurun_out/prof_$tag
        (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_$tag" -o p -- "$@" > "$OLDPWD/gpurun_out/prof_$tag.log" 2>&1)
        echo "[r6] rocprof $tag rc=$?"
        db=$(find gpurun_out/prof_$tag -name "*.db" | head -1); python tools/rocpd_stats.py "$db" $n > gpurun_out/r06_${tag}_kernel_stats.md 2>&1
        head -n 14 gpurun_out/r06_${tag}_kernel_stats.md | cut -c1-150
        [ "$tag" = markov24 ] && python tools/rocpd_idle.py "$db" > gpurun_out/r06_markov24_step_idle.md 2>&1
        find gpurun_out/prof_$tag -type f -size +1M -delete
      }
      R=$PWD
      prof markov24 7 python $R/bench.py --steps 5 --warmup 2 --cpu-steps 0 --no-secondary
      if [ -z "$PROF_ONLY_HEADLINE" ]; then
        prof kochkov256_k32 7 python $R/bench.py --steps 5 --warmup 2 --cpu-steps 0 --no-secondary --grid 256 --layers 12 --modes 32 --batch 2
        prof kochkov256_k64 7 python $R/bench.py --steps 5 --warmup 2 --cpu-steps 0 --no-secondary --grid 256 --layers 24 --modes 64 --batch 2
        prof cube64 7 python $R/tools/bench_mesh.py --preset cube64 --steps 5 --warmup 2
      fi ;;
    pmctraffic)
      # HBM traffic counters of the headline configuration (training step + forward-only pass), one PMC pass each
      R=$PWD
      for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf gpurun_out/pmc_markov24_$c
        (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $c -d "$R/gpurun_out/pmc_markov24_$c" -o p -- python $R/bench.py --steps 3 --warmup 1 --cpu-steps 0 --no-secondary > "$R/gpurun_out/pmc_markov24_$c.log" 2>&1)
        echo "[r6] pmc markov24 $c rc=$?"
      done
      f=$(find gpurun_out/pmc_markov24_FETCH_SIZE -name "*.db" | head -1); w=$(find gpurun_out/pmc_markov24_WRITE_SIZE -name "*.db" | head -1)
      (cd tools && python make_pmc_traffic.py "../$f" "../$w" "$(cat ../fourierflow_amd/lib/git_head.stamp 2>/dev/null || echo unknown)" "markov/24 B=32 64x64 fp32 (bench.py defaults: training step + forward-only pass)") > gpurun_out/pmc_traffic_markov24.json
      head -c 1500 gpurun_out/pmc_traffic_markov24.json; echo
      find gpurun_out/pmc_markov24_FETCH_SIZE gpurun_out/pmc_markov24_WRITE_SIZE -type f -size +1M -delete ;;
    final)
      timeout 600 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1
      echo "[r6] smoke rc=$?"; tail -n 2 gpurun_out/smoke.log
      timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2> gpurun_out/bench.err
      echo "[r6] bench rc=$?"; tail -n 4 gpurun_out/bench.err
      timeout 1700 python -m pytest tests -m gpu -q --maxfail=25 --tb=short -p no:cacheprovider -s --durations=25 > gpurun_out/pytest_gpu.log 2>&1
      echo "[r6] pytest -m gpu rc=$?"; tail -n 32 gpurun_out/pytest_gpu.log | cut -c1-200 ;;
    steptests)
      timeout 900 python -m pytest tests/test_trainer.py tests/test_kernels_ffh.py tests/test_kernels_ffx.py tests/test_block.py -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/pytest_step.log 2>&1
      echo "[r6] step tests rc=$?"; tail -n 3 gpurun_out/pytest_step.log ;;
    fwgrad)
      timeout 900 python -m pytest tests/test_kernels_spectral.py tests/test_range.py tests/test_trainer.py tests/test_bench_geometry.py -m gpu -q -x --tb=short -p no:cacheprovider -k "fw_grad or growth or golden or B32 or B19 or kochkov or mesh3d" -s > gpurun_out/pytest_fwgrad.log 2>&1
      echo "[r6] fwgrad tests rc=$?"; grep -oE "(passed|failed).*|\[bench-geometry B=(32|19)[^]]*\] (worst|vs fp64).{0,200}" gpurun_out/pytest_fwgrad.log | tail -8
      for v in bf16x3 fp16x2 bf16x3 fp16x2; do
        FFNO_FW_GRAD_SPLIT=$v timeout 300 python bench.py --steps 30 --warmup 5 --cpu-steps 0 --no-secondary > gpurun_out/bench_fwgrad_$v.log 2> gpurun_out/bench_fwgrad_$v.err
        python - <<PY
import json
d = json.loads(open("gpurun_out/bench_fwgrad_$v.log").read().strip().splitlines()[-1])
print("FFNO_FW_GRAD_SPLIT=$v", d["value"], "steps/s", d["ms_per_step_median"], "ms median; fw_grad_partial", d["kernels"]["fw_grad_partial"]["avg_us"], "us (zero operands", d["kernels"]["fw_grad_partial"].get("zero_operand_us"), ")")
PY
      done ;;
    small)
      # the non-hot launches one by one, the forward A/B of the self-ranged inference layers, the same-address atomic micro-benchmark
      timeout 300 python tools/time_small.py > gpurun_out/time_small.log 2>&1; echo "[r6] time_small rc=$?"; tail -n 19 gpurun_out/time_small.log
      timeout 300 python tools/ab_forward.py 5 > gpurun_out/ab_forward.log 2>&1; echo "[r6] ab_forward rc=$?"; tail -n 5 gpurun_out/ab_forward.log
      [ -x tools/ubench/bin/atomic_fold ] && { timeout 300 tools/ubench/bin/atomic_fold > gpurun_out/atomic_fold.log 2>&1; echo "[r6] atomic_fold rc=$?"; } ;;
    stack)
      # the persistent inference stack: phase trace of one image group, stack vs per-layer launches over batch sizes, soak
      for b in 8 32; do timeout 200 python tools/trace_stack.py $b 24 2>&1 | grep -v amdgpu.ids; done > gpurun_out/trace_stack.log 2>&1; echo "[r6] trace_stack rc=$?"; grep -v "^phase" gpurun_out/trace_stack.log | tail -n 16
      timeout 400 python tools/ab_stack_batch.py 3 > gpurun_out/ab_stack_batch.log 2>&1; echo "[r6] ab_stack_batch rc=$?"; grep "^B" gpurun_out/ab_stack_batch.log
      timeout 300 python tools/soak_stack.py > gpurun_out/soak_stack.log 2>&1; echo "[r6] soak rc=$?"; tail -n 2 gpurun_out/soak_stack.log ;;
    tables)
      # same-box A/B of the DFT-fragment tables in the <= 16-mode kernels
      for r in 1 2; do for t in 1 0; do FFNO_X3_DFT_TABLES=$t timeout 300 python bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-secondary 2> gpurun_out/ab_tab_$t.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['kernels']
print('$t', d['value'], d['ms_per_step_median'], d['ms_per_forward'], d['ms_per_forward_batch1'], {n: k[n]['avg_us'] for n in k})"; done; done ;;
    *) echo "[r6] unknown stage $st" ;;
  esac
done
