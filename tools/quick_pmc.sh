# dev tool: FETCH_SIZE / WRITE_SIZE of the headline kernel, one --pmc pass each (kernel trace only)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/quick; rm -rf $OUT; mkdir -p $OUT
python tools/gpu_soc_check.py c2 512 2>&1 | tail -3
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -o q -- python tools/gpu_profile_targets.py c2 3 > $OUT/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w -o q -- python tools/gpu_profile_targets.py c2 3 > $OUT/w.log 2>&1
python - <<'PY'
import csv,glob
for d in ("f","w"):
    for p in glob.glob("gpurun_out/quick/%s/**/*counter_collection.csv"%d, recursive=True):
        for r in csv.DictReader(open(p)):
            if "obca" in r["Kernel_Name"]: print(r["Kernel_Name"], r["Counter_Name"], r["Counter_Value"], "scratch", r["Scratch_Size"], "ms", (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
PY
