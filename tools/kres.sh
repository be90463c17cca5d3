#!/bin/bash
# compile one kernel source for gfx950 with -save-temps and print the register / scratch use of its kernels: tools/kres.sh infer.hip
cd /root/repo/fourierflow_amd && rm -rf /tmp/kres && mkdir -p /tmp/kres && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -c csrc/$1 -I csrc -I ../include -o /tmp/kres/out.o -Wno-unused-result -save-temps=obj 2>&1 | grep -v warning | head -30
grep -E "^\s*\.(vgpr_count|agpr_count|vgpr_spill_count|private_segment_fixed_size)|\.name:" /tmp/kres/*gfx950.s | sed 's/^.*\.s://' | paste - - - - - 2>/dev/null | sed 's/  */ /g' | grep "${2:-.}"
