#!/usr/bin/env python3
"""Compact per-kernel register/LDS/occupancy table from hipcc's resource-usage remarks.
usage: python tools/kernel_resources.py scmomer_amd/csrc/favor.hip [...]"""
import re
import subprocess
import sys

for src in sys.argv[1:]:
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-c", src,
                          "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                         capture_output=True, text=True).stderr
    cur = {}
    for line in out.splitlines():
        m = re.search(r"remark: [^:]+:\d+:\d+: (.*?) \[-Rpass", line) or re.search(r"remark: (.*?) \[-Rpass", line)
        if not m:
            continue
        body = m.group(1).strip()
        if body.startswith("Function Name:"):
            cur = {"name": body.split(":", 1)[1].strip()}
        elif ":" in body:
            k, v = body.split(":", 1)
            cur[k.strip()] = v.strip()
            if k.strip().startswith("LDS Size"):
                name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
                name = re.sub(r"\(.*", "", name)
                print(f"{name[:70]:70s} vgpr {cur.get('VGPRs','?'):>4} agpr {cur.get('AGPRs','?'):>4} "
                      f"spill {cur.get('VGPR Spill','?'):>3} scratch {cur.get('ScratchSize [bytes/lane]','?'):>4} "
                      f"occ {cur.get('Occupancy [waves/SIMD]','?')} lds {cur.get('LDS Size [bytes/block]','?')}")
