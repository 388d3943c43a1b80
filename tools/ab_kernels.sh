#!/bin/bash
# Same-box A/B over tools/bench_kernels.py: tools/ab_kernels.sh VARIANT CONFIGS [reps]
cd "$GRAFT_REPO_ROOT" || exit 1
V=$1; C=$2; R=${3:-40}
for i in 1 2; do
  echo "--- shipped"; timeout -k 10 300 python3 tools/bench_kernels.py --reps $R --configs $C 2>/dev/null | cut -c1-110 || exit 1
  echo "--- $V"; SCANSPACE_LIB=$GRAFT_REPO_ROOT/scanspaceimageprocessor_amd/variants/libscanspace_hip.$V.so timeout -k 10 300 python3 tools/bench_kernels.py --reps $R --configs $C 2>/dev/null | cut -c1-110 || exit 1
done
