#!/bin/bash
# Stand-alone timing harnesses of the split-bf16 feed-forward kernels (no torch, ~1 s on the GPU box):
#   build.sh bench            -> tools/ffx_profile/bin/ffx_bench        launch times of fwd2 / bwd_data2 / wgrad per schedule
#   build.sh stamp <abl>      -> tools/ffx_profile/bin/ffx_stamp_<abl>  ffx_chain_sp_kernel with s_memtime stamps around every
#                                iteration (body / barrier wait / rest, per wave) and the ablation mask FFX_ABL of ffx.hip
# The harness is the product source (fourierflow_amd/csrc/ffx.hip) with a main() appended -- nothing is duplicated.
set -e
cd "$(dirname "$0")"
mkdir -p bin
SRC=../../fourierflow_amd/csrc
EXTRA='extern "C" size_t ffno_ff_wgrad_partial_floats(int C, int H, int nsplit) { return (size_t)nsplit * (size_t)(2 * H * C + H + C); }
extern "C" size_t ffno_ff_mask_words(int P, int H) { return ((size_t)((P + 31) / 32) * 32 * H + 31) / 32; }'
if [ "$1" = bench ]; then
    (cat $SRC/ffx.hip; echo "$EXTRA"; cat bench_main.inc) > bin/ffx_bench.hip
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -I $SRC -I ../../include bin/ffx_bench.hip -o bin/ffx_bench
else
    ABL=${2:-0}
    (echo '#include <hip/hip_runtime.h>'; echo '__device__ long long* g_dbg;'; echo "#define FFX_ABL $ABL"
     echo '#define FFX_STAMP(i) { long long tn_ = clock64(); if ((threadIdx.x & 63) == 0 && g_dbg) { long long* T_ = g_dbg + (blockIdx.x * 8 + (threadIdx.x >> 6)) * 8; if (i == 0) { if (tl_) T_[2] += tn_ - tl_; else T_[4] += tn_ - tk_; } else if (i == 1) T_[0] += tn_ - tl_; else if (i == 2) T_[1] += tn_ - tl_; else if (i == 11) T_[3] += tn_ - tk_; else if (i == 12) { T_[5] += tn_ - tk_; T_[6] += wall_clock64() - tw_; } } if (i == 10) { tk_ = tn_; tl_ = 0; tw_ = wall_clock64(); } else tl_ = tn_; }'
     sed 's/    FFX_STAMP(10);/    long long tl_ = 0, tk_ = 0, tw_ = 0; FFX_STAMP(10);/' $SRC/ffx.hip
     echo "$EXTRA"; cat stamp_main.inc) > bin/ffx_stamp_$ABL.hip
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -I $SRC -I ../../include bin/ffx_stamp_$ABL.hip -o bin/ffx_stamp_$ABL
fi
