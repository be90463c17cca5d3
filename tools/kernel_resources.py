#!/usr/bin/env python3
"""Print one line per gfx950 kernel: VGPR/AGPR/SGPR, scratch (spill) bytes, LDS bytes, occupancy."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fourierflow_amd", "csrc")
filt = sys.argv[1] if len(sys.argv) > 1 else ""
for f in sorted(os.listdir(CSRC)):
    if not f.endswith(".hip"):
        continue
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "-c",
                          os.path.join(CSRC, f), "-I", CSRC, "-I", os.path.join(ROOT, "include"), "-o", "/dev/null",
                          "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
    cur = {}
    for line in out.splitlines():
        m = re.search(r"remark: [^ ]+ +(Function Name|VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (.*?) \[-R", line)
        if not m:
            m = re.search(r"(Function Name|VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
        cur[k] = v
        if k.startswith("LDS"):
            name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip().split("(")[0]
            if filt in name:
                print(f"{name[:60]:60s} vgpr={cur.get('VGPRs'):>4} agpr={cur.get('AGPRs'):>4} sgpr={cur.get('SGPRs'):>4} "
                      f"scratch={cur.get('ScratchSize [bytes/lane]'):>5} occ={cur.get('Occupancy [waves/SIMD]'):>2} lds={cur.get('LDS Size [bytes/block]')}")
