#!/bin/bash
# Counter passes over tools/bench_kernels.py for any of its configurations (run on the GPU box):
#   tools/pmc_kernels.sh TAG CONFIGS [BATCH]      e.g.  tools/pmc_kernels.sh r03_lab l
# four rocprofv3 --pmc passes, one counter set each (issue / waits, LDS and memory instructions, FETCH_SIZE, WRITE_SIZE)
#   -> gpurun_out/TAG/{sq,lds,fetch,write}/a_counter_collection.csv; summarise with tools/pmc_table.py
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
T=$1; CFG=$2; B=${3:-8}
O=gpurun_out/$T
mkdir -p "$O"
CMD="python3 tools/bench_kernels.py --configs $CFG --reps 3 --batch $B"
timeout -s KILL 150 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32 \
  -d "$O/sq" -o a --output-format csv -- $CMD > "$O/sq.log" 2>&1 &&
timeout -s KILL 150 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS \
  -d "$O/lds" -o a --output-format csv -- $CMD > "$O/lds.log" 2>&1 &&
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE -d "$O/fetch" -o a --output-format csv -- $CMD > "$O/fetch.log" 2>&1 &&
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE -d "$O/write" -o a --output-format csv -- $CMD > "$O/write.log" 2>&1 &&
echo "passes done: $O"
