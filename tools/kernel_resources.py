#!/usr/bin/env python3
"""Print VGPR / SGPR / scratch / LDS / occupancy per kernel of one .hip file (hipcc remarks)."""
import re, subprocess, sys
src = sys.argv[1]
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src,
                      "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r"remark: +([\w][\w /\[\]]*?): +(\S+) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k in ("Function Name", "Name"):
        cur = {"name": v}
        rows.append(cur)
    else:
        cur[k] = v
dem = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, d in zip(rows, dem):
    d = re.sub(r"\(.*", "", d).replace("cmtfpls::", "").replace("void ", "")
    print(f"{d:58s} vgpr={r.get('VGPRs','?'):>4} agpr={r.get('AGPRs','?'):>3} sgpr={r.get('TotalSGPRs', r.get('SGPRs','?')):>4} "
          f"scratch={r.get('ScratchSize [bytes/lane]','?'):>4} lds={r.get('LDS Size [bytes/block]','?'):>6} occ={r.get('Occupancy [waves/SIMD]','?')}")
