# dev tool: instruction-cache counters of the headline kernel (one --pmc pass, kernel trace only)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/icache; rm -rf $OUT; mkdir -p $OUT
cd /tmp
(cd $OLDPWD && rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/i -o q -- python tools/gpu_profile_targets.py c2 3 > $OUT/i.log 2>&1)
cd $OLDPWD
python - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(float); n=0
for p in glob.glob("gpurun_out/icache/i/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "obca_ipm_kernel_r4" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
print(dict(acc))
if acc.get("SQC_ICACHE_REQ"): print("icache miss rate %.4f, misses per VALU instruction %.5f" % (acc["SQC_ICACHE_MISSES"]/acc["SQC_ICACHE_REQ"], acc["SQC_ICACHE_MISSES"]/max(1,acc["SQ_INSTS_VALU"])))
PY
tail -3 $OUT/i.log
