#!/bin/bash
# developer tool: build libpgx with extra flags for k_match.hip only -> build/variants/libpgx_<name>.so
# usage: tools/build_variant.sh <name> [extra hipcc flags for k_match.hip...]
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
cs=$root/photogrammetry_amd/csrc
mkdir -p $root/build/variants /tmp/pgx_var_$name
make -s -C $cs >/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function "$@" -c $cs/k_match.hip -o /tmp/pgx_var_$name/k_match.o
objs=""
for o in pgx_api pgx_comm k_image k_fast k_nms k_brief k_pose k_tracks pgx_hostutil pgx_tracks; do objs="$objs $cs/$o.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/build/variants/libpgx_$name.so $objs /tmp/pgx_var_$name/k_match.o -ldl
echo built $root/build/variants/libpgx_$name.so
