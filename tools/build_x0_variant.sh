#!/bin/bash
# dev tool: builds libobca_mpc_x0.so (git-ignored, next to libobca_mpc.so) -- the library with the EXPERIMENTAL x0 start of DESIGN.md
# section 9 (the all-zero cold start with every pose at x0) -- from a patched COPY of csrc/; the sources in the tree, and with them
# the source hash the profiles are keyed by, stay as they are.  tools/gpu_x0_variant.py measures it on the GPU box.
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=$ROOT/vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd
TMP=$(mktemp -d)
mkdir -p $TMP/pkg $TMP/include
cp -r $PKG/csrc $TMP/pkg/csrc
rm -rf $TMP/pkg/csrc/_build
cp $ROOT/include/obca_mpc.h $TMP/include/
python3 - "$TMP/pkg/csrc/obca_kernel.hip" <<'PY'
import sys
p = sys.argv[1]
s = open(p).read()
anchor = "    if (from_window && lane == 0) window_start_point(S.x, S.xref, &in, L.N, L.NS, L.free_T ? L.iT() : -1);\n"
assert s.count(anchor) == 1
s = s.replace(anchor, anchor + "    if (!from_window && !warm) for (int t = lane; t < 3 * (L.N + 1); t += NT) { const int k = t / 3; S.x[L.ip(k) + (t - 3 * k)] = in.x0[t - 3 * k]; }\n")
open(p, "w").write(s)
PY
cd $TMP/pkg/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wno-pass-failed -DOBCA_COLD_AT_X0"
for f in obca_kernel obca_kernel_mw obca_lpi obca_capi obca_rollout obca_astar; do hipcc $FLAGS -c $f.hip -o $f.o & done
wait
g++ -shared -fPIC -o $PKG/libobca_mpc_x0.so obca_kernel.o obca_kernel_mw.o obca_lpi.o obca_capi.o obca_rollout.o obca_astar.o -lm
rm -rf $TMP
ls -la $PKG/libobca_mpc_x0.so
