"""Per-call kernel times of the masking path from a rocprofv3 results database.

usage: python tools/mask_profile.py gpurun_out/prof_m/m_results.db
(made with: rocprofv3 --kernel-trace --stats -d DIR -o m -- python3 tools/bench_kernels.py --configs m --reps 5)
"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
q = (f"select s.kernel_name, d.start, (d.end-d.start)/1000.0 from {kd} d join rocpd_info_kernel_symbol s "
     "on d.kernel_id=s.id where s.kernel_name like '%mask_%' order by d.start")
calls = []
for name, _, us in c.execute(q):
    k = re.search(r"mask_(\w+?)_kernel", name).group(1)
    if k == "runs":
        calls.append({})
    calls[-1].setdefault(k, []).append(us)
for d in calls:
    print("  ".join(f"{k} {sum(v):7.0f}" + (" " + str([round(x) for x in v if x > 20]) if k == "flatten" else "")
                    for k, v in d.items()), f"  total {sum(sum(v) for v in d.values()):7.0f} us")
