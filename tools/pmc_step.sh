cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_WAIT_ANY"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_step/$tag -- python $R/tools/time_step.py 4096 20 base > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, os, collections
R=os.environ['GRAFT_REPO_ROOT']
for f in sorted(glob.glob(R+'/gpurun_out/pmc_step/*/*/*counter_collection.csv')):
    acc=collections.defaultdict(lambda: [0,0])
    for r in csv.DictReader(open(f)):
        if r['Kernel_Name'].startswith('wbc_step_kernel'):
            a=acc[r['Counter_Name']]; a[0]+=float(r['Counter_Value']); a[1]+=1
    for k,(v,n) in acc.items(): print(k, v/n)
PY
