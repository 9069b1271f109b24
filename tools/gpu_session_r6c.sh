mkdir -p gpurun_out/r6c
timeout 900 python -m pytest tests/test_gpu_f4_witness.py -m gpu -x -q > gpurun_out/r6c/pytest_f4.txt 2>&1; echo "rc f4 $?"; tail -5 gpurun_out/r6c/pytest_f4.txt
python bench.py > gpurun_out/r6c/bench_N1.json 2> gpurun_out/r6c/bench_N1.err; echo "rc bench $?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r6c/bench_N1.json') if l.startswith('{')][0])
g=d['secondary']['gadget_witness']
print('trace', g['trace']['ms'], g['trace']['roofline']['frac'], 'scaled', g['trace_scaled']['ms'], g['trace_scaled']['roofline']['frac'], g['trace_scaled']['last_round_times_mul_equals_perm'])
print('value', d['value'], d['secondary'].get('error'))
PY
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d gpurun_out/r6c/pmc_sec -- python3 tools/secondary_kernels.py > gpurun_out/r6c/secondary_kernels.json 2> gpurun_out/r6c/pmc_sec.log
python - <<'PY'
import csv, glob
acc={}
for f in glob.glob('gpurun_out/r6c/pmc_sec/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if 'trace' in k or 'witness' in k:
            acc.setdefault((k,r['Counter_Name']),[]).append(float(r['Counter_Value']))
for (k,c),v in sorted(acc.items()): print(k,c,len(v),sum(v)/len(v))
PY
