#!/bin/bash
# dev tool: the library with -DOBCA_PROFILE (per-phase shader-clock counters, read by tools/gpu_prof.py) as
# libobca_mpc_prof.so next to the product library; objects under /tmp
set -e
cd "$(dirname "$0")/../vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd/csrc"
mkdir -p /tmp/obca_prof
for f in obca_kernel obca_kernel_mw obca_lpi obca_capi obca_rollout obca_astar; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -DOBCA_PROFILE $OBCA_HIPCC_FLAGS -c $f.hip -o /tmp/obca_prof/$f.o &
done
wait
g++ -shared -fPIC -o ../libobca_mpc_prof.so /tmp/obca_prof/*.o -lm
