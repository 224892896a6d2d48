#!/bin/bash
# rocprofv3 evidence of a round (run on the MI355X box through gpurun): kernel trace + stats of the default bench, then the
# counter passes -- each --pmc pass in its OWN run, never combined with trace domains other than --kernel-trace.
# Outputs under gpurun_out/prof_<tag>/; tools/pmc_summary.py <tag> condenses them into profiles/.
#   gpurun -- 'bash tools/profile.sh r04'   then here:   python tools/pmc_summary.py r04
set -u
TAG=${1:-r04}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
run() { (cd $REPO && "$@"); }
# 1. kernel trace + stats of the bench itself (the roofline figures of bench.py come from the same command)
(cd $REPO && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o $TAG -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err)
# 2. counter passes on the dominant kernels only
for tgt in c2 c5 c3g; do
  for ctr in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES"; do
    name=$(echo $ctr | tr ' ' '_' | cut -c1-40)
    (cd $REPO && rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_${tgt}_$name -o $TAG -- python tools/gpu_profile_targets.py $tgt 3 > $OUT/pmc_${tgt}_$name.log 2>&1)
  done
done
ls $OUT
