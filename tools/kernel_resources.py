"""Summarise hipcc -Rpass-analysis=kernel-resource-usage remarks (VGPR/SGPR/scratch/occupancy/LDS per kernel)."""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
seen = set()
for b in blocks:
    name = b.split("\n")[0].strip()
    if name in seen:
        continue
    seen.add(name)

    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1

    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(ss::ChainParams.*", "", dem).replace("void ss::", "")
    scratch = g(r"ScratchSize \[bytes/lane\]")
    occ = g(r"Occupancy \[waves/SIMD\]")
    lds = g(r"LDS Size \[bytes/block\]")
    print(f"{dem[:70]:70s} vgpr={g('VGPRs'):4d} sgpr={g('SGPRs'):4d} scratch={scratch:4d} occ={occ} lds={lds}")
