#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the two rocprofv3 PMC passes of the bench (tools/gpu_session.sh pmc):
usage: tools/make_pmc_traffic.py FETCH.db WRITE.db > profiles/pmc_traffic.json
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (FETCH_SIZE x2 = the gfx950 correction of
MI355X_MICROARCH.md, calibrated on adamw_kernel: it reads p, g, m, v = 4 x n floats and writes 3 x n)."""
import json
import sys

from rocpd_pmc import per_kernel

# first match wins, so the instantiations of the default arithmetic (fp16x2) come before the generic patterns: bench.py also runs
# the all-bf16x3 variant, whose kernels are in the same trace
NAMES = [("spectral_x3_pair_kernel<16, true, ffno::StF32, true", "spectral_mix"), ("infer_ff_kernel", "infer_ff"),
         ("infer_stack_kernel", "infer_stack"),
         ("ffw_chain_kernel<64, 256, false, ffno::StF32", "ff_fwd"), ("ffw_chain_kernel<64, 256, true, ffno::StF32", "ff_bwd_data"),
         ("ffh_wgrad_m_multi_kernel<64, 256, 8, ffno::StF32", "ff_bwd_weights_partial"),
         ("ffh_wgrad_m_multi_kernel<64, 256", "ff_bwd_weights_partial"), ("ffh_wgrad_m_multi_kernel<32, 128", "ff_bwd_weights_partial"),
         ("ffh_wgrad_m_kernel<64, 256", "ff_bwd_weights_partial"), ("spectral_x3_pair_kernel<16, true", "spectral_fused"),
         ("spectral_x3k_pair_kernel<64, true", "spectral_fused"), ("spectral_x3k_pair_kernel<128, true", "spectral_fused"),
         ("spectral_x3c32_pair_kernel<true", "spectral_fused_pair"), ("spectral_x3c32_kernel<true", "spectral_fused_single"),
         ("ffh_wgrad_m_kernel<32, 128", "ff_bwd_weights_partial"), ("ffx_chain_rs_kernel<32, 128, false, ffno::SplitHf2", "ff_fwd"),
         ("ffx_chain_kernel<32, 128, true, ffno::SplitHf2", "ff_bwd_data"), ("ffx_chain_rs_kernel<64, 256, false, ffno::SplitHf2", "ff_fwd"),
         ("ffx_chain_kernel<64, 256, true, ffno::SplitHf2", "ff_bwd_data"), ("ffx_wgrad_kernel<64, 256, ffno::SplitHf2", "ff_bwd_weights_partial"),
         ("spectral_x3_pair_kernel<16", "spectral_fused"), ("ffx_chain_rs_kernel<64, 256, false", "ff_fwd"),
         ("ffx_chain_rs_kernel<64, 256, true", "ff_bwd_data"), ("ffx_wgrad_rs_kernel", "ff_bwd_weights_partial"),
         ("fw_grad_x3_kernel", "fw_grad_partial"),
         ("spectral_fused_pair_kernel", "spectral_fused"), ("spectral_fused_kernel", "spectral_fused"), ("ffx_chain_kernel<64, 256, false", "ff_fwd"),
         ("ffx_chain_kernel<64, 256, true", "ff_bwd_data"), ("ffx_wgrad_kernel", "ff_bwd_weights_partial"),
         ("ffx_wgrad_reduce_kernel", "ff_bwd_weights_reduce"), ("ff_chain_kernel<64, 256, 8, false>", "ff_fwd"),
         ("ff_chain_kernel<64, 256, 8, true>", "ff_bwd_data"), ("ff_bwd_weights_partial_kernel", "ff_bwd_weights_partial"),
         ("fw_grad_partial_kernel", "fw_grad_partial"), ("fw_grad_reduce_kernel", "fw_grad_reduce"),
         ("adamw_kernel", "adamw"), ("ffx_pack_kernel", "ffx_pack")]


def short(kernel):
    for pat, name in NAMES:
        if pat in kernel:
            return name
    return None


fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
GIT_HEAD = sys.argv[3] if len(sys.argv) > 3 else None
out = {}
order = sorted(fetch.items(), key=lambda kv: next((i for i, (pat, _) in enumerate(NAMES) if pat in kv[0]), len(NAMES)))
for k, (_, n, avg, _, _) in order:
    name = short(k)
    if name is None or k not in write or name in out:
        continue
    w = write[k][2]
    out[name] = dict(fetch_kib_raw=round(avg, 1), write_kib=round(w, 1), hbm_bytes_per_launch=int((2 * avg + w) * 1024),
                     launches_sampled=n, symbol=k)
WORKLOAD = sys.argv[4] if len(sys.argv) > 4 else "markov/24 B=32 64x64 fp32 (bench.py defaults)"
print(json.dumps(dict(workload=WORKLOAD, git_head=GIT_HEAD,
                      method="rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); bytes = "
                             "(2*FETCH_SIZE + WRITE_SIZE)*1024; the x2 on FETCH_SIZE is the gfx950 correction of "
                             "MI355X_MICROARCH.md, confirmed on adamw_kernel; WRITE_SIZE reads exact",
                      kernels=out), indent=1))
