#!/bin/bash
# shader clock while the feed-forward kernels run back to back (rocm-smi sampled beside the stand-alone harness)
cd "$(dirname "$0")/../.."
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
(for i in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 17 18 19 20; do tools/ffx_profile/bin/ffx_bench > /dev/null; done) &
LOAD=$!
sleep 1.0
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" | tr '\n' ' '; echo; sleep 0.4; done
wait $LOAD
