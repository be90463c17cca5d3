#!/bin/bash
# SQ counters of one bench configuration (two PMC passes, kernel-trace + pmc only).  usage: tools/pmc_sq.sh TAG [bench args...]
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
tag=$1; shift
R=$PWD
run() {  # name, counters...
  n=$1; shift
  rm -rf gpurun_out/pmc_${tag}_$n
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d "$R/gpurun_out/pmc_${tag}_$n" -o ffno -- python "$R/bench.py" --steps 3 --warmup 1 --cpu-steps 0 --no-secondary $BENCH_ARGS > "$R/gpurun_out/pmc_${tag}_$n.log" 2>&1)
  echo "[pmc] $tag $n rc=$?"
  db=$(find gpurun_out/pmc_${tag}_$n -name "*.db" | head -1); python tools/rocpd_pmc_multi.py "$db" ffno > gpurun_out/pmc_${tag}_$n.md 2>&1
  find gpurun_out/pmc_${tag}_$n -type f -size +1M -delete
}
BENCH_ARGS="$*"
run A SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
run B SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
