#!/bin/bash
# dev tool (round-5 review item 4a): the 256-register build of the compile-time-shape kernels (two waves per SIMD possible) against
# the product build on the headline workload, with the counters that say why.  Build first:
#   tools/build_variant.sh regs256 '-DOBCA_SHAPE_KERNEL_ATTR=__attribute__((amdgpu_waves_per_eu(2,2)))'
# then on the GPU box: bash tools/regs256_experiment.sh  ->  gpurun_out/regs256/
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/regs256; rm -rf $OUT; mkdir -p $OUT
for lib in libobca_mpc.so libobca_mpc_regs256.so; do
  OBCA_LIB=$lib python tools/gpu_variant_bench.py c2 2>&1 | grep -v amdgpu.ids | sed "s/^/$lib /" >> $OUT/summary.txt
  for ctr in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU"; do
    name=$(echo $ctr | tr ' ' '_' | cut -c1-24)
    OBCA_LIB=$lib rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/${lib}_$name -o q -- python tools/gpu_profile_targets.py c2 3 > $OUT/${lib}_$name.log 2>&1
    f=$(find $OUT/${lib}_$name -name '*counter_collection.csv' | head -1)
    python - "$f" "$lib" <<'PY' >> $OUT/summary.txt
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'obca_ipm' in r['Kernel_Name']]
acc = collections.OrderedDict()
for r in rows:
    acc.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
extra = {k: rows[-1].get(k) for k in ('Scratch_Size', 'VGPR_Count', 'Accum_VGPR_Count', 'SGPR_Count', 'LDS_Block_Size')} if rows else {}
print(sys.argv[2], rows[-1]['Kernel_Name'] if rows else '?', extra, {k: v[-1] for k, v in acc.items()})
PY
  done
done
cat $OUT/summary.txt
