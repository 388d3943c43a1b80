#!/bin/bash
# Build a tuning variant of libscanspace_hip.so: [FILES="usm mask"] tools/build_variant.sh NAME [extra hipcc flags...]
# (FILES: the sources recompiled with the extra flags, default usm; the other objects come from the main build)
# -> scanspaceimageprocessor_amd/variants/libscanspace_hip.NAME.so  (select it with SCANSPACE_LIB=...)
set -e
cd "$(dirname "$0")/../scanspaceimageprocessor_amd/csrc"
name=$1; shift
mkdir -p ../variants build/$name
for f in chain usm stats resize mask lab api; do
  if [[ " ${FILES:-usm} " == *" $f "* ]] || [ ! -f build/$f.o ]; then
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -fno-slp-vectorize "$@" -c $f.hip -o build/$name/$f.o &
  else
    cp build/$f.o build/$name/$f.o
  fi
done
wait
hipcc -shared -fPIC --offload-arch=gfx950 -pthread -o ../variants/libscanspace_hip.$name.so build/$name/*.o
echo built ../variants/libscanspace_hip.$name.so
