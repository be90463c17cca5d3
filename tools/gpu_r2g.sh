#!/bin/bash
# round-2 session G: ring-8 x3 mix (batch 32 / 1), LayerNorm + layer-entry parity, PMC traffic passes of the bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_spectral.py tests/test_block.py tests/test_layer_entry.py tests/test_velocity.py tests/test_kernels_ffx.py -m gpu -q --maxfail=10 --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu_g.log 2>&1
echo "[session] pytest subset rc=$?"; tail -n 3 gpurun_out/pytest_gpu_g.log
for b in 32 1; do
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-secondary --batch $b 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $b:', d['value'], 'ms/step', d['ms_per_step'], 'fwd', d['ms_per_forward'], 'b1', d['ms_per_forward_batch1'], {k:v['avg_us'] for k,v in d['kernels'].items()})"
done
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d "$OLDPWD/gpurun_out/pmc_$c" -o ffno -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --cpu-steps 0 --no-secondary > "$OLDPWD/gpurun_out/pmc_$c.log" 2>&1)
  echo "[session] pmc $c rc=$?"
done
f=$(find gpurun_out/pmc_FETCH_SIZE -name "*.db" | head -1); w=$(find gpurun_out/pmc_WRITE_SIZE -name "*.db" | head -1)
python tools/rocpd_pmc.py "$f" > gpurun_out/pmc_FETCH_SIZE.md; python tools/rocpd_pmc.py "$w" > gpurun_out/pmc_WRITE_SIZE.md
(cd tools && python make_pmc_traffic.py "../$f" "../$w") > gpurun_out/pmc_traffic.json; head -c 1500 gpurun_out/pmc_traffic.json
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name "*.db" -size +20M -delete
