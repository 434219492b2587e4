"""Prints VGPR/SGPR/LDS/scratch/occupancy per kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re
import subprocess
import sys

src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
r = subprocess.run(cmd, capture_output=True, text=True)
cur = None
rows = []
for line in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["/usr/bin/c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": name}
        rows.append(cur)
        continue
    for key in ("VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"):
        m = re.search(re.escape(key) + r": (\d+)", line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
for row in rows:
    nm = re.sub(r"\(anonymous namespace\)::", "", row["name"])
    nm = re.sub(r"\(.*", "", nm)
    print(f"{nm[:110]:110s} vgpr={row.get('VGPRs')} agpr={row.get('AGPRs')} sgpr={row.get('TotalSGPRs')} "
          f"scratch={row.get('ScratchSize [bytes/lane]')} occ={row.get('Occupancy [waves/SIMD]')} lds={row.get('LDS Size [bytes/block]')}")
