#!/bin/bash
# Headline kernel only, steady clocks: tools/bench_short.sh TAG [extra bench.py arguments]  (run on the GPU box)
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-short}; shift
timeout -k 10 300 python3 bench.py --no-others --no-cpu-baseline --steps 100 "$@" > "gpurun_out/b_$T.json" 2> "gpurun_out/b_$T.err" || { tail -5 "gpurun_out/b_$T.err"; exit 1; }
python3 - "gpurun_out/b_$T.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("MP/s", d["value"], "kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"])
PY
