#!/bin/bash
# Same-box comparison of the headline kernel across several variants: tools/ab_many.sh ROUNDS VARIANT...
cd "$GRAFT_REPO_ROOT" || exit 1
N=$1; shift
for i in $(seq 1 $N); do
  echo -n "shipped : "; bash tools/bench_short.sh abm_main_$i || exit 1
  for V in "$@"; do
    echo -n "$V : "; SCANSPACE_LIB=$GRAFT_REPO_ROOT/scanspaceimageprocessor_amd/variants/libscanspace_hip.$V.so bash tools/bench_short.sh abm_${V}_$i || exit 1
  done
done
