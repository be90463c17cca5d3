#!/bin/bash
# SQ counters of the stand-alone weight-gradient launch for a list of library builds.  usage: tools/pmc_wgrad.sh TAG lib.so ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
tag=$1; shift
R=$PWD
run() {  # name, counters...
  n=$1; shift
  for lib in $LIBS; do
    b=$(basename $lib .so)
    rm -rf gpurun_out/pmcw_${tag}_${b}_$n
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d "$R/gpurun_out/pmcw_${tag}_${b}_$n" -o ffno -- python "$R/tools/time_wgrad.py" "$R/$lib" > "$R/gpurun_out/pmcw_${tag}_${b}_$n.log" 2>&1)
    db=$(find gpurun_out/pmcw_${tag}_${b}_$n -name "*.db" | head -1); python tools/rocpd_pmc_multi.py "$db" wgrad > gpurun_out/pmcw_${tag}_${b}_$n.md 2>&1
    find gpurun_out/pmcw_${tag}_${b}_$n -type f -size +1M -delete
    echo "== $b $n"; cat gpurun_out/pmcw_${tag}_${b}_$n.md
  done
}
LIBS="$*"
run A SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run B SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
