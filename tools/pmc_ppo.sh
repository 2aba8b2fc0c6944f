cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_ppo/$tag -- python $R/tools/time_ppo.py > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, os, collections
R=os.environ['GRAFT_REPO_ROOT']
for kn in ("ppo_chain","ppo_fwd_bwd","ppo_wgrad_kernel"):
    print(kn)
    for f in sorted(glob.glob(R+'/gpurun_out/pmc_ppo/*/*/*counter_collection.csv')):
        acc=collections.defaultdict(lambda: [0,0])
        for r in csv.DictReader(open(f)):
            if r['Kernel_Name'].startswith(kn):
                a=acc[r['Counter_Name']]; a[0]+=float(r['Counter_Value']); a[1]+=1
        for k,(v,n) in acc.items(): print("  ",k, v/n)
PY
