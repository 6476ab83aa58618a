#!/bin/bash
# a side library that differs from the product in ONE translation unit:  bash scripts/build_variant.sh <name> <file.hip> -DSWITCH ...
# (objects of the other units come from quake_amd/build; select with QUAKE_HIP_LIB=quake_amd/lib/libquake_hip_<name>.so)
set -e
cd "$(dirname "$0")/.."
NAME=$1; SRC=$2; shift 2
OBJ=/tmp/variant_${NAME}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wno-unused-value "$@" -c quake_amd/csrc/$SRC -o $OBJ
OTHERS=$(ls quake_amd/build/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o quake_amd/lib/libquake_hip_${NAME}.so $OBJ $OTHERS
echo quake_amd/lib/libquake_hip_${NAME}.so
