#!/bin/bash
# Same-box A/B of the headline kernel: tools/ab.sh VARIANT [rounds]   (shipped library against variants/libscanspace_hip.VARIANT.so)
cd "$GRAFT_REPO_ROOT" || exit 1
V=$1; N=${2:-2}
for i in $(seq 1 $N); do
  echo -n "shipped  : "; bash tools/bench_short.sh ab_main_$i || exit 1
  echo -n "$V : "; SCANSPACE_LIB=$GRAFT_REPO_ROOT/scanspaceimageprocessor_amd/variants/libscanspace_hip.$V.so bash tools/bench_short.sh ab_${V}_$i || exit 1
done
