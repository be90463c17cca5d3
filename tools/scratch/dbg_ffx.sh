for d in 0 1 2; do echo "== FFX_XY=$d"; FFX_XY=$d timeout 120 python tools/time_ffx.py 2>&1 | grep "ffx_fwd  (mask)\|ffx_bwd_data"; done
