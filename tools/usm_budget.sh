#!/bin/bash
# per-phase VALU budget of the unsharp-mask kernel: counters of the full kernel and of the three ablation builds
set -e
cd "$GRAFT_REPO_ROOT"
for v in base abl_NO_P1 abl_NO_P2 abl_NO_P3; do
  if [ $v = base ]; then unset SCANSPACE_LIB; else export SCANSPACE_LIB=$GRAFT_REPO_ROOT/scanspaceimageprocessor_amd/variants/libscanspace_hip.$v.so; fi
  bash tools/pmc_valu.sh budget_$v 2
  python3 tools/pmc_read.py $(find gpurun_out/budget_$v -name '*counter_collection.csv' | head -1) | grep usm_kernel > gpurun_out/budget_$v.txt
  cat gpurun_out/budget_$v.txt
done
