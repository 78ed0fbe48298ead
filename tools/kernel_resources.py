"""Print VGPR / spill / LDS / occupancy per kernel of one HIP source (compile-time remarks, no GPU needed)."""
import re, subprocess, sys

src = sys.argv[1]
out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-fast-math", "-ffp-contract=off", "-c", src,
                      "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:], capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: (?:.*?:\d+:\d+: )?\s*(Function Name|VGPRs|AGPRs|VGPR Spill|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]|ScratchSize \[bytes/lane\]): (\S+)", line)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(anonymous namespace\)::|\(.*", "", cur)
        rows[cur] = {}
    elif cur:
        rows[cur][k.split(" [")[0]] = v
for k, r in rows.items():
    print(f"{k:60s} vgpr {r.get('VGPRs','?'):>4} agpr {r.get('AGPRs','?'):>3} spill {r.get('VGPR Spill','?'):>3} scratch {r.get('ScratchSize','?'):>4} lds {r.get('LDS Size','?'):>6} occ {r.get('Occupancy','?')}")
