#!/bin/bash
# One SQ counter pass over the unsharp-mask kernel: tools/pmc_valu.sh OUTDIR  (run on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -s KILL 90 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32 \
  -d "gpurun_out/$1" -o a --output-format csv -- python3 tools/bench_kernels.py --configs ${2:-2} --reps 2 > "gpurun_out/$1.log" 2>&1
