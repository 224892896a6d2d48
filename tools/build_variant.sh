#!/bin/bash
# dev tool: the library from the sources in the tree with extra compile flags, as libobca_mpc_<name>.so next to the product
# library (git-ignored); objects under /tmp.  OBCA_LIB=libobca_mpc_<name>.so selects it in tools/gpu_variant_bench.py,
# gpu_cmp_builds.py, gpu_prof.py.
#   tools/build_variant.sh prof -DOBCA_PROFILE        per-phase shader-clock counters (tools/gpu_prof.py)
#   tools/build_variant.sh straight -DOBCA_LOOP_R4=false -DOBCA_LOOP_R56=false     form of the ladder's passes per kernel (csrc/obca_kernel.hip: solve_passes)
set -e
NAME=$1; shift
cd "$(dirname "$0")/../vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd/csrc"
OBJ=/tmp/obca_variant_$NAME
rm -rf $OBJ; mkdir -p $OBJ
for f in obca_kernel obca_kernel_mw obca_lpi obca_capi obca_rollout obca_astar obca_kernel_s5_2_2 obca_kernel_s5_6_18 obca_kernel_s6_2_2 obca_kernel_s5_3_6 obca_kernel_s5_4_10 obca_kernel_s5_5_14 obca_kernel_s6_3_6 obca_kernel_s6_4_10 obca_kernel_s6_5_14 obca_kernel_mw_s20_3_6 obca_kernel_mw_s20_5_14; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wno-pass-failed "$@" -c $f.hip -o $OBJ/$f.o &
done
wait
g++ -shared -fPIC -o ../libobca_mpc_$NAME.so $OBJ/*.o -lm
ls -la ../libobca_mpc_$NAME.so
