#!/bin/bash
# dev tool: compile ONLY the headline kernel's translation unit (obca_kernel_s5_3_6.hip) with extra backend flags, link it with the
# product's other objects into libobca_mpc_<name>.so (git-ignored) for an A/B on the GPU (tools/ab_run.sh).
#   tools/flag_variants.sh name1 "flags1" name2 "flags2" ...
set -e
cd "$(dirname "$0")/../vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd/csrc"
PROD=$(cat _build/linked_flags)
while [ $# -ge 2 ]; do
  NAME=$1; FLAGS=$2; shift 2
  OBJ=/tmp/obca_flagvar_$NAME; rm -rf $OBJ; mkdir -p $OBJ
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wno-pass-failed $FLAGS -c obca_kernel_s5_3_6.hip -o $OBJ/obca_kernel_s5_3_6.o &&
    g++ -shared -fPIC -o ../libobca_mpc_$NAME.so $(ls $PROD/*.o | grep -v obca_kernel_s5_3_6.o) $OBJ/obca_kernel_s5_3_6.o -lm && echo "$NAME ok: $FLAGS" ) &
done
wait
