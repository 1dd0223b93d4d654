#!/bin/bash
# tools/build_variant.sh NAME FILE.hip [-DFLAG ...]: lib/lab/libmemgym_NAME.so = the lab build with FILE.hip compiled with the extra flags
# (an experiment's variant next to the product; run `python __graft_entry__.py` first so that build/lab/*.o exist)
set -e
cd "$(dirname "$0")/../endless-memory-gym_amd"
name=$1; f=$2; shift 2
extra=""; [ "$f" = mg_spot.hip ] && extra="-mllvm -disable-machine-licm"
mkdir -p build/variants lib/lab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -DMG_LAB $extra "$@" -c csrc/$f -o build/variants/$name.o
objs=$(ls build/lab/*.o | grep -v "/$f.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/lab/libmemgym_$name.so $objs build/variants/$name.o
echo lib/lab/libmemgym_$name.so
