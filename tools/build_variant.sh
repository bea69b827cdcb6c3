#!/bin/bash
# usage: tools/build_variant.sh NAME "-DFLAG ..." file.hip [file.hip ...]   -> tools/bin/libaim_NAME.so
# A diagnostic build of the library: the listed sources are recompiled with the extra flags, the rest come from csrc/build.
# Select it with AIM_HIP_LIB=tools/bin/libaim_NAME.so (tools only).
set -e
NAME=$1; FLAGS=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/adapt-image-models_amd/csrc
make -s -C $C
mkdir -p $R/tools/bin/obj_$NAME
OBJS=""
for f in $C/*.hip; do
  b=$(basename $f .hip)
  if [[ " $* " == *" $b.hip "* ]]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-gpu-rdc $FLAGS -c $f -o $R/tools/bin/obj_$NAME/$b.o
    OBJS="$OBJS $R/tools/bin/obj_$NAME/$b.o"
  else
    OBJS="$OBJS $C/build/$b.o"
  fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/tools/bin/libaim_$NAME.so $OBJS
echo $R/tools/bin/libaim_$NAME.so
