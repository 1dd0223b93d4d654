#!/bin/bash
# tools/build_all_variant.sh NAME [-DFLAG ...]: lib/lab/libmemgym_NAME.so = the lab build with EVERY source compiled with the extra flags
# (for switches that live in shared headers, e.g. -DMG_LAB_OBS_STRIDE=21248 -DMG_LAB_XCD_GROUP in csrc/mg_stream_out.hpp)
set -e
cd "$(dirname "$0")/../endless-memory-gym_amd"
name=$1; shift
mkdir -p build/variants/$name lib/lab
for f in csrc/mg_*.hip; do
  b=$(basename $f)
  extra=""; [ "$b" = mg_spot.hip ] && extra="-mllvm -disable-machine-licm"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -DMG_LAB $extra "$@" -c $f -o build/variants/$name/$b.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/lab/libmemgym_$name.so build/variants/$name/*.o
echo lib/lab/libmemgym_$name.so
