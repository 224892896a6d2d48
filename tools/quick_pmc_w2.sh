# dev tool: the PMC passes of tools/quick_pmc.sh for the two-wavefront kernel (mode 4)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_r02/pmc_c2w2_SQ; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT -o r02 -- python tools/gpu_profile_targets.py c2w2 3 > $OUT.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do O=$PWD/gpurun_out/prof_r02/pmc_c2w2_$c; rm -rf $O; rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O -o r02 -- python tools/gpu_profile_targets.py c2w2 3 > $O.log 2>&1; done
python - <<'PY'
import csv,glob
for p in sorted(glob.glob("gpurun_out/prof_r02/pmc_c2w2_*/**/*counter_collection.csv", recursive=True)):
    acc={}
    for r in csv.DictReader(open(p)):
        if "obca" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"])); meta=(r["Kernel_Name"],r["Scratch_Size"],r["VGPR_Count"],r["Workgroup_Size"],(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
    print(meta, {k:max(v) for k,v in acc.items()})
PY
