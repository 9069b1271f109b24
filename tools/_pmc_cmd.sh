cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4l; export TMPDIR=/tmp
for what in trace witness; do
  arg=""; [ $what = trace ] && arg="trace"
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY -d gpurun_out/r4l/${what}_a -o w --output-format csv -- python3 tools/witness_probe.py 20 2 $arg > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS -d gpurun_out/r4l/${what}_b -o w --output-format csv -- python3 tools/witness_probe.py 20 2 $arg > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/r4l/*/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if 'trace' in row['Kernel_Name'] or 'witness' in row['Kernel_Name']:
            acc[row['Counter_Name']].append(float(row['Counter_Value']))
    print(f.split('/')[2], {k: "%.4g" % (sum(v)/len(v)) for k, v in acc.items()})
PY
