#!/bin/bash
# round-2 session E: 256x256 stage kernels (split-bf16 vs fp32) with kernel traces
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_spectral.py tests/test_block.py -m gpu -q --maxfail=10 --tb=short -p no:cacheprovider -k "x3 or large" > gpurun_out/pytest_gpu_e.log 2>&1
echo "[session] pytest subset rc=$?"; tail -n 3 gpurun_out/pytest_gpu_e.log
for v in "" "--no-x3"; do
  tag=x3$(echo $v | tr -d ' -')
  rm -rf gpurun_out/prof256_$tag
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof256_$tag" -o k -- python "$OLDPWD/bench.py" --grid 256 --layers 12 --modes 32 --batch 2 --steps 10 --warmup 3 --cpu-steps 0 $v > "$OLDPWD/gpurun_out/bench256_$tag.log" 2>&1)
  python -c "
import json
d=json.loads(open('gpurun_out/bench256_$tag.log').read().strip().splitlines()[-1]); print('256 $tag', d['value'], {k:v['avg_us'] for k,v in d['kernels'].items()})"
  db=$(find gpurun_out/prof256_$tag -name "*.db" | head -1); python tools/rocpd_stats.py "$db" > gpurun_out/kernel_stats_256_$tag.md 2>&1; head -n 16 gpurun_out/kernel_stats_256_$tag.md | cut -c1-170
  find gpurun_out/prof256_$tag -size +20M -delete
done
