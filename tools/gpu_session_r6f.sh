export TMPDIR=/tmp
mkdir -p gpurun_out/r6f
rocprofv3 --list-avail > gpurun_out/r6f/avail.txt 2>&1
grep -o "SQ_[A-Z_0-9]*" gpurun_out/r6f/avail.txt | sort -u | tr '\n' ' ' | head -c 6000; echo
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_WAIT_ANY" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_WAVES SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-60)
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/r6f/pmc_$tag -- python3 tools/trace_pmc_probe.py > /dev/null 2> gpurun_out/r6f/log_$tag.txt
done
python3 - <<'PY'
import csv, glob
acc={}
for f in glob.glob('gpurun_out/r6f/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if 'trace' in k or 'k_perm_fast' in k:
            acc.setdefault((k,r['Counter_Name']),[]).append(float(r['Counter_Value']))
for (k,c),v in sorted(acc.items()): print("%-28s %-26s n=%d avg %.5g" % (k,c,len(v),sum(v)/len(v)))
PY
