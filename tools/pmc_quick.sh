set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_quick; rm -rf $OUT; mkdir -p $OUT
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/$ctr -o q -- python tools/gpu_profile_targets.py c2 3 > $OUT/$ctr.log 2>&1
  f=$(find $OUT/$ctr -name '*counter_collection.csv' | head -1)
  python - "$f" $ctr <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
v=[float(r['Counter_Value']) for r in rows if 'obca_ipm' in r['Kernel_Name'] and r['Counter_Name']==sys.argv[2]]
print(sys.argv[2], 'per launch (KiB as reported):', [round(x) for x in v])
PY
done
