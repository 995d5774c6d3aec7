#!/bin/bash
# Evidence for DESIGN.md section 4, "the predicated-load hazard": the generated code of the NMS round kernel in the form that
# produces wrong survivors (EXPFORM=0, EXPSUB=3: one load predicate per lane and row + divergent selects on the level) and
# in the two builds of the SAME source that are exact (-mllvm -disable-peephole, -mllvm -simplifycfg-hoist-common=false).
# Needs no GPU (hipcc cross-compiles); writes tests/nmsexp/disasm/:
#   round_s3_fail.s, round_s3_nopeephole.s, round_s3_nohoist.s   k_nmsm_round<2, 0>, comments and directives stripped
#   round_s3_fail_vs_nopeephole.shape.diff, ..._vs_nohoist...      diffs with register numbers and labels blanked out, so
#                                                                  that only real instruction changes show (40 lines for
#                                                                  the peephole switch: three copies not folded, moved
#                                                                  s_nop / s_waitcnt, one more s_waitcnt vmcnt(1) ahead
#                                                                  of the loop)
# The source is rebuilt from commit 348b354's k_nms.hip + tests/nmsexp/experiment.patch in a scratch directory.
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
D="$(mktemp -d)"
mkdir -p "$D/photogrammetry_amd/csrc" "$D/include"
for f in photogrammetry_amd/csrc/k_nms.hip photogrammetry_amd/csrc/pgx_internal.h include/pgx.h; do git -C "$ROOT" show 348b354:$f > "$D/$f"; done
(cd "$D" && patch -p1 < "$ROOT/tests/nmsexp/experiment.patch" > /dev/null)
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I$D/include --cuda-device-only -S -DEXPFORM=0 -DEXPSUB=3"
/opt/rocm/bin/hipcc $F "$D/photogrammetry_amd/csrc/k_nms.hip" -o "$D/s3_fail.s" 2> /dev/null &
/opt/rocm/bin/hipcc $F -mllvm -disable-peephole "$D/photogrammetry_amd/csrc/k_nms.hip" -o "$D/s3_nopeephole.s" 2> /dev/null &
/opt/rocm/bin/hipcc $F -mllvm -simplifycfg-hoist-common=false "$D/photogrammetry_amd/csrc/k_nms.hip" -o "$D/s3_nohoist.s" 2> /dev/null &
wait
(cd "$D" && python3 "$ROOT/tests/nmsexp/disasm_norm.py" s3_fail s3_nopeephole s3_nohoist && python3 "$ROOT/tests/nmsexp/disasm_shape.py" s3_fail s3_nopeephole s3_nohoist)
O="$ROOT/tests/nmsexp/disasm"
mkdir -p "$O"
for v in s3_fail s3_nopeephole s3_nohoist; do cp "$D/$v.round2.s" "$O/round_$v.s"; done
diff -U4 "$D/s3_fail.shape.s" "$D/s3_nopeephole.shape.s" > "$O/round_s3_fail_vs_nopeephole.shape.diff" || true
diff -U4 "$D/s3_fail.shape.s" "$D/s3_nohoist.shape.s" > "$O/round_s3_fail_vs_nohoist.shape.diff" || true
rm -rf "$D"
ls -la "$O"
