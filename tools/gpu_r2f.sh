#!/bin/bash
# round-2 session F: 8-line x3 tiles (batch-1 latency), parity subset
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_spectral.py tests/test_block.py tests/test_layer_entry.py tests/test_rollout.py tests/test_routine.py -m gpu -q --maxfail=10 --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu_f.log 2>&1
echo "[session] pytest subset rc=$?"; tail -n 3 gpurun_out/pytest_gpu_f.log
for b in 32 8 1; do
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-secondary --batch $b 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $b:', d['value'], 'ms/step', d['ms_per_step'], 'fwd', d['ms_per_forward'], 'b1', d['ms_per_forward_batch1'], {k:v['avg_us'] for k,v in d['kernels'].items()})"
done
timeout 300 python tools/bench_rollout.py 2>&1 | tail -n 3 | cut -c1-600
rm -rf gpurun_out/prof_b1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_b1" -o k -- python "$OLDPWD/tools/bench_rollout.py" > "$OLDPWD/gpurun_out/rollout_prof.log" 2>&1)
db=$(find gpurun_out/prof_b1 -name "*.db" | head -1); python tools/rocpd_stats.py "$db" > gpurun_out/kernel_stats_rollout_b1.md 2>&1; head -n 12 gpurun_out/kernel_stats_rollout_b1.md | cut -c1-170
find gpurun_out/prof_b1 -size +20M -delete
