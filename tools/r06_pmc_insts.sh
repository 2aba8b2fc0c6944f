# instruction counts per wave of wbc_step_kernel on tools/time_step.py (one counter group): bash tools/r06_pmc_insts.sh [envs]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=${1:-4096}
rm -rf $R/gpurun_out/pmc_insts
V=${2:-base}
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_insts -- python $R/tools/time_step.py $N 20 $V > /dev/null 2>&1
python - <<'PY'
import csv, glob, os, collections
R=os.environ['GRAFT_REPO_ROOT']
for kern in ('wbc_step_kernel', 'wbc_simulate_kernel', 'wbc_fk_kernel', 'wbc_reset_kernel'):
    acc=collections.defaultdict(lambda: [0,0])
    for f in sorted(glob.glob(R+'/gpurun_out/pmc_insts/*/*counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            if r['Kernel_Name'].startswith(kern):
                a=acc[r['Counter_Name']]; a[0]+=float(r['Counter_Value']); a[1]+=1
    w=acc['SQ_WAVES'][0]
    if w: print(kern, {k: round(v/w,1) for k,(v,n) in acc.items()}, 'launches', acc['SQ_WAVES'][1])
PY
