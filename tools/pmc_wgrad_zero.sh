cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
for z in 0 1; do
  rm -rf gpurun_out/pmcz_$z
  (cd /tmp && WG_ZERO=$z timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA -d "$R/gpurun_out/pmcz_$z" -o ffno -- python "$R/tools/time_wgrad.py" > "$R/gpurun_out/pmcz_$z.log" 2>&1)
  db=$(find gpurun_out/pmcz_$z -name "*.db" | head -1); python tools/rocpd_pmc_multi.py "$db" wgrad > gpurun_out/pmcz_$z.md 2>&1
  find gpurun_out/pmcz_$z -type f -size +1M -delete
  echo "== WG_ZERO=$z"; cat gpurun_out/pmcz_$z.md
done
