# HBM traffic of the PPO minibatch kernels: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (MI355X_MICROARCH.md, HBM section)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export WBC_ITERS=10
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_traffic_ppo/$c -- python $R/tools/time_ppo.py > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, os, statistics
R=os.environ['GRAFT_REPO_ROOT']
for kn in ("ppo_chain_kernel", "ppo_wgrad_kernel", "ppo_grad_reduce_kernel", "chain_pack_kernel"):
    for c in ("FETCH_SIZE","WRITE_SIZE"):
        f=sorted(glob.glob(f"{R}/gpurun_out/pmc_traffic_ppo/{c}/*/*counter_collection.csv"))[-1]
        v=[float(r['Counter_Value']) for r in csv.DictReader(open(f)) if r['Kernel_Name'].startswith(kn) and r['Counter_Name']==c]
        if v: print(kn, c, "launches", len(v), "median_KB", statistics.median(v), "mean_KB", sum(v)/len(v))
PY
