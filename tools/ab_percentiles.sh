#!/bin/bash
# Same-box comparison of ss_negative_params across library variants: tools/ab_percentiles.sh VARIANT...
cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2; do
  echo "--- shipped"; timeout -k 10 120 python3 tools/bench_percentiles.py 8 64 2>/dev/null || exit 1
  for V in "$@"; do
    echo "--- $V"; SCANSPACE_LIB=$GRAFT_REPO_ROOT/scanspaceimageprocessor_amd/variants/libscanspace_hip.$V.so timeout -k 10 120 python3 tools/bench_percentiles.py 8 64 2>/dev/null || exit 1
  done
done
