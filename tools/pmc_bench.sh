#!/bin/bash
# Everything the bench line quotes, collected over bench.py itself (run on the GPU box): tools/pmc_bench.sh TAG
#   four counter passes (SQ set, FETCH_SIZE, WRITE_SIZE, instruction mix; a pass each) -> profiles/TAG_usm_pmc.json
#   rocprofv3 --kernel-trace --stats over the headline workload alone    -> gpurun_out/TAG_bench/trace_headline
#     (bench.py --no-others --no-cpu-baseline: every usm_kernel launch in it is the headline launch)
#   rocprofv3 --kernel-trace --stats over the default bench command     -> gpurun_out/TAG_bench/trace
#   the default bench command on its own                                 -> gpurun_out/TAG_bench/bench.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-r03}
O=gpurun_out/${T}_bench
mkdir -p "$O"
SHORT="--steps 3 --warmup 2 --no-others --no-cpu-baseline"
timeout -s KILL 150 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32 \
  -d "$O/sq" -o a --output-format csv -- python3 bench.py $SHORT > "$O/sq.log" 2>&1 &&
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE -d "$O/fetch" -o a --output-format csv -- python3 bench.py $SHORT > "$O/fetch.log" 2>&1 &&
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE -d "$O/write" -o a --output-format csv -- python3 bench.py $SHORT > "$O/write.log" 2>&1 &&
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FLOPS_FP32 \
  -d "$O/mix" -o a --output-format csv -- python3 bench.py $SHORT > "$O/mix.log" 2>&1 &&
python3 tools/pmc_summary.py "profiles/${T}_usm_pmc.json" 1536e6 usm_kernel \
  "$(ls $O/sq/*/a_counter_collection.csv $O/sq/a_counter_collection.csv 2>/dev/null | head -1)" \
  "$(ls $O/fetch/*/a_counter_collection.csv $O/fetch/a_counter_collection.csv 2>/dev/null | head -1)" \
  "$(ls $O/write/*/a_counter_collection.csv $O/write/a_counter_collection.csv 2>/dev/null | head -1)" \
  "$(ls $O/mix/*/a_counter_collection.csv $O/mix/a_counter_collection.csv 2>/dev/null | head -1)" > "$O/summary.log" 2>&1 &&
cp profiles/${T}_usm_pmc*.{json,csv} "$O/" &&
echo "counters done" &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$O/trace_headline" -o t --output-format csv -- python3 bench.py --no-others --no-cpu-baseline > "$O/bench_traced_headline.json" 2> "$O/bench_traced_headline.err" &&
timeout -k 10 420 rocprofv3 --kernel-trace --stats -d "$O/trace" -o t --output-format csv -- python3 bench.py > "$O/bench_traced.json" 2> "$O/bench_traced.err" &&
echo "trace done" &&
timeout -k 10 420 python3 bench.py > "$O/bench.json" 2> "$O/bench.err" &&
cat "$O/bench.json"
