#!/bin/bash
# VALU counters and timing of library variants: tools/pmc_variants.sh CONFIG NAME [NAME ...]   ("base" = the in-tree library)
cd "$GRAFT_REPO_ROOT" || exit 1
cfg=$1; shift
for v in "$@"; do
  if [ "$v" = base ]; then unset SCANSPACE_LIB; else export SCANSPACE_LIB=$GRAFT_REPO_ROOT/scanspaceimageprocessor_amd/variants/libscanspace_hip.$v.so; fi
  bash tools/pmc_valu.sh var_$v $cfg || exit 1
  echo "== $v: $(python3 tools/pmc_read.py $(find gpurun_out/var_$v -name '*counter_collection.csv' | head -1) | grep -E 'usm_kernel|chain_kernel' | sed 's/.*|//')"
  timeout -k 10 100 python3 tools/bench_kernels.py --configs $cfg --reps 30 2>&1 | tail -1
done
