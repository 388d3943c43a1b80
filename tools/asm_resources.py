#!/usr/bin/env python3
"""Per-kernel VGPR / SGPR / scratch / LDS from the metadata of a `hipcc -S --cuda-device-only` listing."""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
meta = txt[txt.index("amdhsa.kernels:"):]
for blk in meta.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*", "", dem).replace("void ss::", "")
    if pat and pat not in dem:
        continue
    g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))
    print(f"{dem:48s} vgpr {g('vgpr_count'):3d} sgpr {g('sgpr_count'):3d} scratch {g('private_segment_fixed_size'):5d} "
          f"lds {g('group_segment_fixed_size'):6d} sgpr_spill {g('sgpr_spill_count'):3d} vgpr_spill {g('vgpr_spill_count'):3d}")
