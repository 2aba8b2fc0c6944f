# per-phase instruction counts of wbc_step_kernel (tools/phase_counts.py): bash tools/r06_phase_counts.sh [envs]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=${1:-4096}
python $R/tools/build_variant.py phaseexit -DWBC_PHASE_EXIT 2>&1 | tail -1
rm -rf $R/gpurun_out/pmc_phase
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_phase -- python $R/tools/phase_counts.py $N > /dev/null 2>&1
python - <<'PY'
import csv, glob, os, collections, sys
R=os.environ['GRAFT_REPO_ROOT']
sys.path.insert(0, R + '/tools')
SEQ = [11, 12, 0, 25, 27, 28, 26, 1, 2, 3, 4, 5, 6, 18, 7, 8, 9, 10, 13, 14, 15, -1]
NAMES = {11: 'entry + deal', 12: 'prologue (loads)', 0: 'torque pass + substep set-up', 25: 'base / box in F', 27: 'walk: prefetch', 28: 'walk: levels 0..3', 26: 'walk: levels 4..5', 1: '-', 2: '-',
         3: 'inertias + bias forces', 4: 'pass 2', 5: 'root', 6: 'pass 3', 18: 'contact detection', 7: 'contact set-up (blocks)', 8: 'solver sweeps', 9: 'contact outputs', 10: 'integrate',
         13: 'substeps 2..4', 14: 'rigid bodies', 15: 'task logic + rewards', -1: 'reset + observe + store'}
d = collections.defaultdict(dict)
for f in sorted(glob.glob(R+'/gpurun_out/pmc_phase/*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if r['Kernel_Name'].startswith('wbc_step_kernel'):
            d[int(r['Dispatch_Id'])][r['Counter_Name']] = float(r['Counter_Value'])
ids = sorted(d)[-len(SEQ):]
prev = {'SQ_INSTS_VALU': 0, 'SQ_INSTS_SALU': 0, 'SQ_INSTS_LDS': 0}
print(f"{'phase (ends at stamp)':40s} {'VALU':>8s} {'SALU':>8s} {'LDS':>8s}   per wave, cumulative VALU")
for k, i in zip(SEQ, ids):
    w = d[i]['SQ_WAVES']
    cur = {c: d[i][c] / w for c in prev}
    print(f"{NAMES[k]:34s} ({k:3d}) {cur['SQ_INSTS_VALU'] - prev['SQ_INSTS_VALU']:8.1f} {cur['SQ_INSTS_SALU'] - prev['SQ_INSTS_SALU']:8.1f} {cur['SQ_INSTS_LDS'] - prev['SQ_INSTS_LDS']:8.1f}   {cur['SQ_INSTS_VALU']:8.1f}")
    prev = cur
PY
