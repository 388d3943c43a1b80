#!/bin/bash
# Dynamic instruction mix of the library's kernels in a bench_kernels.py config: tools/pmc_mix.sh OUTDIR CONFIGS  (GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 \
  -d "gpurun_out/$1" -o a --output-format csv -- python3 tools/bench_kernels.py --configs ${2:-2} --reps 2 > "gpurun_out/$1.log" 2>&1
