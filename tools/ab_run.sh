# dev tool: A/B of library builds on the GPU.  usage: bash tools/ab_run.sh "libA.so libB.so" "c2 c3g5" [cmp]
LIBS=${1:-"libobca_mpc_base.so libobca_mpc.so"}
WL=${2:-"c2 c2m12 c3g5 c3f c3g"}
for lib in $LIBS; do
  for w in $WL; do
    if [ "$w" = "c5" ]; then OBCA_LIB=$lib OBCA_QUEUE_MODES=2,2 python tools/gpu_c5_modes.py 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/$lib c5 /"
    else OBCA_LIB=$lib python tools/gpu_variant_bench.py $w 2>&1 | grep -v amdgpu.ids; fi
  done
  if [ -n "$3" ]; then OBCA_LIB=$lib python tools/gpu_cmp_builds.py /tmp/cmp_$lib.npz 2>&1 | grep -v amdgpu.ids; fi
done
if [ -n "$3" ]; then python - $LIBS <<'PY'
import sys, numpy as np
ref = np.load("/tmp/cmp_%s.npz" % sys.argv[1])
for lib in sys.argv[2:]:
    o = np.load("/tmp/cmp_%s.npz" % lib)
    bad = [k for k in ref.files if not np.array_equal(ref[k], o[k])]
    worst = max([float(np.max(np.abs(ref[k].astype(float) - o[k].astype(float)))) for k in bad], default=0.0)
    print("%s vs %s: %s" % (lib, sys.argv[1], "every output word equal" if not bad else "DIFFERENT in %s (max |diff| %.3e)" % (bad, worst)))
PY
fi
