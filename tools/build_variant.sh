#!/bin/bash
# builds advanced-soft-actor-critic_amd/lib/libasac_hip_<tag>.so with extra -D flags on ONE translation unit (the other objects
# are reused):   tools/build_variant.sh <tag> <file.hip> -DFOO -DBAR=1       (A/B through ASAC_HIP_LIB)
set -e
R=$(cd $(dirname $0)/.. && pwd)
tag=$1; src=$2; shift 2
L=$R/advanced-soft-actor-critic_amd/lib
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function -I$R/include -I$R/advanced-soft-actor-critic_amd/csrc "$@" \
  -c $R/advanced-soft-actor-critic_amd/csrc/$src -o /tmp/variant_$tag.o
base=${src%.hip}; base=${base%_old}       # <x>_old.hip stands in for x.hip
objs=$(ls $L/obj/*.o | grep -v "/${base}.o")
hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/variant_$tag.o -o $L/libasac_hip_$tag.so
echo $L/libasac_hip_$tag.so
