for v in "FFNO_FF_SCHED=3" "FFNO_FF_SCHED=3 FFNO_EXP_NOSUM=1" "FFNO_FF_SCHED=2" "FFNO_FF_SCHED=2 FFNO_EXP_NOSUM=1"; do
  env $v timeout 300 python bench.py --steps 10 --warmup 3 --cpu-steps 0 --no-secondary > gpurun_out/exp.log 2> gpurun_out/exp.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/exp.log").read().strip().splitlines()[-1])
print("$v", d["value"], {n: k["avg_us"] for n, k in d["kernels"].items()})
PY
done
