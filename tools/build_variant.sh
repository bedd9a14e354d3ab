#!/bin/bash
# usage: tools/build_variant.sh <name> "<extra hipcc flags>" [groups...]  -- ab/libwrhip_<name>.so: the product library with the given
# instantiation groups (wrhip_inst.h; "host" = wrhip.hip) recompiled under extra flags, the other objects taken from csrc/build/
# (run `make -j8 libwrhip.so` first).  For A/B measurements on one GPU box (tools/ab.sh).
set -e
name=$1; flags=$2; shift 2
cd "$(dirname "$0")/../webrender_amd/csrc"
mkdir -p build/$name ../../ab
HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result -mllvm -structurizecfg-skip-uniform-regions"
objs=""
for g in host 1 2 3 4 5 6 7 8; do
  if [[ " $* " == *" $g "* ]]; then
    if [ $g = host ]; then /opt/rocm/bin/hipcc $HIPFLAGS $flags -c wrhip.hip -o build/$name/wrhip.o & objs="$objs build/$name/wrhip.o"
    else /opt/rocm/bin/hipcc $HIPFLAGS $flags -DWR_INST_GROUP=$g -c wrhip_inst.hip -o build/$name/wrhip_inst_$g.o & objs="$objs build/$name/wrhip_inst_$g.o"; fi
  else
    if [ $g = host ]; then objs="$objs build/wrhip.o"; else objs="$objs build/wrhip_inst_$g.o"; fi
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic $objs -o ../../ab/libwrhip_$name.so
ls -la ../../ab/libwrhip_$name.so
