#!/bin/bash
# Developer experiment ("the predicated-load hazard"): rebuild the NMS round kernel in the forms that
# were compared, as separate libraries next to the shipped libpgx.so.  Run from the repository root after `make -C
# photogrammetry_amd/csrc` (the other objects are reused).  The shipped k_nms.hip is restored afterwards.
#   EXPFORM 0 = form under investigation (global neighbour reads), 1 = shipped evaluation on the same global reads
#   EXPSUB  0 = per-cell predicates + ternaries (fails)      1 = per-cell predicates + branch-free evaluation (passes)
#           2 = unconditional loads + ternaries (passes)     3 = one predicate per lane and row + ternaries (fails most)
#   EXPDELAY_EVAL=n / EXPDELAY_ACCEPT=n: n x s_sleep 127 between filter and evaluation / between evaluation and acceptance
#   run time: PGX_NMS_SPLIT=1 (frozen-state phases), PGX_NMS_COLOUR=1 (one non-interacting colour class per launch)
set -e
cd "$(dirname "$0")/../.."
C=photogrammetry_amd/csrc
BASE=348b354   # the commit whose k_nms.hip the patch was cut against (the forms under test do not depend on later changes)
cp $C/k_nms.hip /tmp/k_nms_ship.hip
trap 'cp /tmp/k_nms_ship.hip '$C'/k_nms.hip' EXIT
git show $BASE:$C/k_nms.hip > $C/k_nms.hip
patch -p1 < tests/nmsexp/experiment.patch
OBJS="pgx_api.o pgx_comm.o k_image.o k_fast.o k_brief.o k_match.o k_pose.o pgx_hostutil.o pgx_tracks.o"
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Iinclude"
bld() { n=$1; shift; /opt/rocm/bin/hipcc $F "$@" -c $C/k_nms.hip -o /tmp/nms_$n.o && (cd $C && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tests/nmsexp/libpgx_$n.so $OBJS /tmp/nms_$n.o -ldl); }
bld s0 -DEXPFORM=0 -DEXPSUB=0 & bld s1 -DEXPFORM=0 -DEXPSUB=1 & bld s2 -DEXPFORM=0 -DEXPSUB=2 & bld s3 -DEXPFORM=0 -DEXPSUB=3 &
wait
bld s3e8 -DEXPFORM=0 -DEXPSUB=3 -DEXPDELAY_EVAL=8 & bld s3a8 -DEXPFORM=0 -DEXPSUB=3 -DEXPDELAY_ACCEPT=8 & bld g1e8 -DEXPFORM=1 -DEXPDELAY_EVAL=8 &
wait
echo "on the GPU box:  for v in s0 s1 s2 s3 s3e8 s3a8 g1e8; do PGX_LIB=\$PWD/tests/nmsexp/libpgx_\$v.so python tests/nmsexp/diag.py | tail -1; done"
