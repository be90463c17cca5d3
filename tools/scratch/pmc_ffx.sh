cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD
pass() { n=$1; shift; rm -rf gpurun_out/pmc_ffx_$n; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d "$R/gpurun_out/pmc_ffx_$n" -o x -- python "$R/tools/time_ffx.py" > "$R/gpurun_out/pmc_ffx_$n.log" 2>&1); db=$(find gpurun_out/pmc_ffx_$n -name "*.db" | head -1); python tools/rocpd_pmc_multi.py "$db" ff > gpurun_out/pmc_ffx_$n.md 2>&1; cat gpurun_out/pmc_ffx_$n.md | cut -c1-260; find gpurun_out/pmc_ffx_$n -name "*.db" -size +10M -delete; }
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
pass b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
pass c SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_INSTS_WAVE32_LDS
