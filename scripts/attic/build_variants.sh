#!/bin/bash
# builds A/B variants of libquake_hip.so: scripts/build_variants.sh name "-DQK_OPT_X=0 ..."
set -e
cd /root/repo
NAME=$1; shift
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wno-unused-value $@"
mkdir -p quake_amd/build/$NAME
/opt/rocm/bin/hipcc $FLAGS -c quake_amd/csrc/qk_scan.hip -o quake_amd/build/$NAME/qk_scan.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o quake_amd/lib/libquake_hip_$NAME.so quake_amd/build/$NAME/qk_scan.o quake_amd/build/qk_ctx.o quake_amd/build/qk_store.o quake_amd/build/qk_dense.o quake_amd/build/qk_kmeans.o quake_amd/build/qk_aps.o quake_amd/build/qk_api.o
echo built $NAME
