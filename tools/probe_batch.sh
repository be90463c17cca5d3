for B in 8 16 32 64 128; do
  timeout 300 python bench.py --steps 10 --warmup 3 --cpu-steps 0 --no-secondary --batch $B > gpurun_out/pb_$B.log 2> gpurun_out/pb_$B.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/pb_$B.log").read().strip().splitlines()[-1])
print($B, d["value"], {n: k["avg_us"] for n, k in d["kernels"].items()})
PY
done
