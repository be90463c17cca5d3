#!/bin/bash
# other configurations at the end of round 5 (one gpurun call): batch 19, plasticity / airfoil meshes, latency by batch
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 200 python bench.py --steps 20 --warmup 5 --batch 19 --cpu-steps 0 --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[others] markov/24 batch 19:', d['value'], 'steps/s', d['ms_per_step'], 'ms/step')"
timeout 200 python tools/bench_mesh.py --preset plasticity 2>/dev/null | tail -n 1 | cut -c1-300
timeout 200 python tools/bench_mesh.py --preset airfoil 2>/dev/null | tail -n 1 | cut -c1-300
timeout 200 python tools/bench_latency.py 2>/dev/null | tail -n 12 | cut -c1-200
